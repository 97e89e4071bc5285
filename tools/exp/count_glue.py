"""CPU: elementwise 'glue' launches of one training step OUTSIDE the kernel launchers (what would be separate ATen
kernels on the GPU), by op and shape: run the model on the float64 emulation, count aten ops while no launcher is active.
    python tools/exp/count_glue.py [t2|t4s|q2s] [0|1 train2]"""
import os
import sys
from collections import Counter

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpu_kernels  # noqa: E402
from oracle import gemnet_oracle as GO  # noqa: E402
from gemnet_pytorch_amd import kernels as K  # noqa: E402
from gemnet_pytorch_amd import ops  # noqa: E402
from test_model_cpu import build  # noqa: E402
from test_oracle_model import load_case  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "t2"
ops.USE_TRAIN2 = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "model2.npz" if tag.endswith("s") else "model.npz")))
cfg, params, inputs = load_case(g, tag)
depth = [0]
cnt, launches = Counter(), Counter()
SKIP = {"view", "_unsafe_view", "reshape", "t", "transpose", "permute", "detach", "alias", "expand", "slice", "select",
        "unsqueeze", "squeeze", "as_strided", "empty", "empty_like", "new_empty", "empty_strided", "_to_copy", "lift_fresh",
        "contiguous", "unbind", "split", "new_empty_strided", "is_same_size", "stride", "sym_size", "numel", "dim"}


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if depth[0] == 0:
            name = func.__name__.split(".")[0]
            if name not in SKIP:
                o = out[0] if isinstance(out, (tuple, list)) and out else out
                shp = tuple(o.shape) if torch.is_tensor(o) else None
                if shp is None or len(shp) == 0 or int(np.prod(shp)) > 0:
                    cnt[(name, shp)] += 1
        return out


with cpu_kernels.emulate():
    saved = {}
    for n in cpu_kernels._NAMES:
        f = getattr(K, n)
        saved[n] = f

        def wrapped(*a, _f=f, _n=n, **k):
            if depth[0] == 0 and not _n.endswith("supported") and _n != "is_angle_form":
                launches[_n] += 1
            depth[0] += 1
            try:
                return _f(*a, **k)
            finally:
                depth[0] -= 1
        setattr(K, n, wrapped)
    model = build(cfg, params).train()
    inp = dict(inputs)
    inp["R"] = inp["R"].double()
    with Mode():
        E, F = model(inp)
        loss = GO.training_loss(E[:, :1], F, torch.tensor(g[f"{tag}.Et"]).double()[:, None], torch.tensor(g[f"{tag}.Ft"]).double())
        loss.backward()
    for n, f in saved.items():
        setattr(K, n, f)
byop = Counter()
for (n, s), c in cnt.items():
    byop[n] += c
print(f"{tag} train2={ops.USE_TRAIN2}: launcher calls {sum(launches.values())}: {dict(launches.most_common(30))}")
print(f"glue aten ops {sum(cnt.values())}: {dict(byop.most_common(25))}")
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1])[:30]:
    print("   ", c, k)
