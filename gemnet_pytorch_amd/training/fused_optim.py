"""Trainer-step fusion: shared-gradient rescale + global-norm clip + AdamW/Adam(amsgrad) +
EMA of the reference trainer (gemnet/training/trainer.py:115-160,250-278,353-358; ema_decay.py:68-93) as two
launches over one flat fp32 buffer (csrc/optim.hip) instead of ~100 small launches over ~60 parameter tensors.

Parameters and gradients become views into flat buffers (the gradient buffer is the one the data-parallel
all-reduce already uses), so construct this BEFORE capturing a hipGraph of the step.
"""
import torch

from .. import _lib
from .._lib import check, ptr, stream


class FusedAdamWEMA:
    def __init__(self, model, lr=1e-3, weight_decay=2e-6, betas=(0.9, 0.999), eps=1e-7, ema_decay=0.999,
                 grad_clip_max=10.0, use_ema=True):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        pad = lambda k: (k + 63) // 64 * 64   # every tensor starts 256-B aligned (the GEMMs want 16-B aligned rows)
        n = sum(pad(p.numel()) for p in self.params)
        self.n = n
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        gscale = torch.ones(n, device=dev, dtype=torch.float32)
        wd = torch.zeros(n, device=dev, dtype=torch.float32)
        shared = {id(l.weight): 1.0 / model.num_blocks for l in
                  [model.mlp_rbf3, model.mlp_cbf3, model.mlp_rbf_h]
                  + ([] if model.triplets_only else [model.mlp_rbf4, model.mlp_cbf4, model.mlp_sbf4])}
        shared[id(model.mlp_rbf_out.weight)] = 1.0 / (model.num_blocks + 1)
        off = 0
        with torch.no_grad():
            for name, p in named:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                p.grad = self.flat_g[off:off + k].view_as(p)
                gscale[off:off + k] = shared.get(id(p), 1.0)
                if not any(s in name for s in ("atom_emb", "frequencies", "bias")):
                    wd[off:off + k] = weight_decay
                off += pad(k)
        # the parameters moved into the flat buffer: everything keyed by their old addresses is stale
        if getattr(model, "_wcache", None):
            model._wcache.clear()
        if getattr(model, "_packs", None) is not None:
            model._packs.clear()
        self.gscale, self.wd = gscale, wd
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.vmax = torch.zeros_like(self.flat_p)
        self.ema = self.flat_p.clone() if use_ema else None
        self.lib = _lib.load()
        self.partial = torch.zeros(int(self.lib.gn_optim_blocks(n)), device=dev, dtype=torch.float64)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.lr, self.betas, self.eps, self.ema_decay, self.clip = lr, betas, eps, ema_decay, grad_clip_max
        self.steps = 0
        self._model = model

    def zero_grad(self):
        self.flat_g.zero_()

    def step(self, lr=None, flag=None):
        """`flag` (runtime.RangeFlag, optional): a non-finite gradient norm then SKIPS the update on the device and sets the
        flag's GRAD bit (the reference would write NaN into every parameter) — the caller polls it."""
        self.steps += 1
        cache = getattr(self._model, "_wcache", None)
        if cache:
            cache.clear()   # the kernel below rewrites the flat parameter buffer without touching tensor versions
        check(_lib.load().gn_adamw_ema_step_f32(
            ptr(self.flat_p), ptr(self.flat_g), ptr(self.gscale), ptr(self.wd), ptr(self.m), ptr(self.v),
            ptr(self.vmax), ptr(self.ema), self.n, ptr(self.partial), float(self.clip),
            float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
            int(self.steps), float(self.ema_decay), ptr(self.grad_norm), ptr(flag.word) if flag is not None else None,
            flag.GRAD if flag is not None else 0, stream()), "gn_adamw_ema_step_f32")
        if flag is not None:
            flag.mirror()

    def ema_parameters(self):
        """Views of the averaged weights in parameter order (ExponentialMovingAverage.shadow_params)."""
        out, off = [], 0
        for p in self.params:
            out.append(self.ema[off:off + p.numel()].view_as(p))
            off += (p.numel() + 63) // 64 * 64
        return out
