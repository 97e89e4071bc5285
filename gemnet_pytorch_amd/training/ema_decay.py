"""Exponential moving average of the trainable parameters (counterpart of gemnet/training/ema_decay.py:18-186).

One flat shadow buffer instead of a list of per-parameter clones: update / store / copy_to / restore are
each a single fused elementwise pass over ~2 M floats."""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=False):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self._params = [p for p in parameters if p.requires_grad]
        self._sizes = [p.numel() for p in self._params]
        self.shadow = self._flatten()
        self.backup = None

    def _flatten(self):
        return torch.cat([p.detach().reshape(-1) for p in self._params]) if self._params else torch.zeros(0)

    def _scatter(self, flat):
        with torch.no_grad():
            for p, chunk in zip(self._params, torch.split(flat, self._sizes)):
                p.copy_(chunk.view_as(p))

    @property
    def shadow_params(self):
        return [c.view_as(p) for p, c in zip(self._params, torch.split(self.shadow, self._sizes))]

    def update(self, parameters=None):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            cur = self._flatten()
            if self.shadow.device != cur.device:
                self.shadow = self.shadow.to(cur.device)
            self.shadow.sub_((self.shadow - cur) * (1.0 - decay))

    def store(self, parameters=None):
        self.backup = self._flatten().clone()

    def copy_to(self, parameters=None):
        self._scatter(self.shadow)

    def restore(self, parameters=None):
        if self.backup is None:
            raise RuntimeError("restore() called before store()")
        self._scatter(self.backup)

    @property
    def collected_params(self):
        if self.backup is None:
            return []
        return [c.view_as(p) for p, c in zip(self._params, torch.split(self.backup, self._sizes))]

    def state_dict(self):
        """The reference's checkpoint layout (ema_decay.py:148-159): per-parameter lists under `shadow_params` /
        `collected_params` (views into the flat buffers), so trainer checkpoints are interchangeable."""
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def _pack(self, tensors, what):
        if len(tensors) != len(self._params) or any(t.numel() != n for t, n in zip(tensors, self._sizes)):
            raise ValueError(f"{what} does not match the tracked parameters")
        ref = self.shadow
        return torch.cat([t.detach().reshape(-1).to(device=ref.device, dtype=ref.dtype) for t in tensors]).clone()

    def load_state_dict(self, state_dict):
        """Copies (never aliases) the incoming tensors, like the reference's deepcopy (ema_decay.py:168-169).
        Also accepts the flat `shadow` / `backup` keys written by round-1 checkpoints of this package."""
        decay = state_dict["decay"]
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        num_updates = state_dict["num_updates"]
        assert num_updates is None or isinstance(num_updates, int), "Invalid num_updates"
        if "shadow_params" in state_dict:
            shadow = state_dict["shadow_params"]
            assert isinstance(shadow, list) and all(isinstance(t, torch.Tensor) for t in shadow), \
                "shadow_params must be a list of Tensors"
            collected = state_dict.get("collected_params") or []
            assert isinstance(collected, list) and all(isinstance(t, torch.Tensor) for t in collected), \
                "collected_params must be a list of Tensors"
            new_shadow = self._pack(shadow, "shadow_params")
            new_backup = self._pack(collected, "collected_params") if collected else None
        else:
            new_shadow = state_dict["shadow"].detach().to(self.shadow.device).clone()
            new_backup = None if state_dict.get("backup") is None else state_dict["backup"].detach().clone()
        self.decay, self.num_updates = decay, num_updates
        self.shadow, self.backup = new_shadow, new_backup
