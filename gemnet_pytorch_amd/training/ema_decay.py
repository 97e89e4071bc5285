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

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow": self.shadow,
                "backup": self.backup}

    def load_state_dict(self, state_dict):
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        self.shadow = state_dict["shadow"].to(self.shadow.device)
        self.backup = state_dict["backup"]
