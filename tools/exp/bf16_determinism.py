"""GPU diagnosis: is the plain-bf16 chain arithmetic (NPL = 1) run-to-run reproducible?  (a) one stack-like chain program
repeated, per mode and tile height; (b) the published GemNet-T / GemNet-Q models (8 x 64 atoms), eager twice, per mode."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gemnet_pytorch_amd import kernels as K  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(*s, generator=g).to(dev)


for M in (1000, 18122, 93156):
    x, W0, W1, W2, res = rnd(M, 128), rnd(128, 128) / 11, rnd(128, 128) / 11, rnd(64, 128) / 11, rnd(M, 128)
    r16, W16 = rnd(M, 16), rnd(128, 16) / 4
    for mode in ("split6", "bf16"):
        outs = []
        for rep in range(6):
            y, z0, z1 = (torch.empty(M, 128, device=dev) for _ in range(3))
            y64 = torch.empty(M, 64, device=dev)
            p = K.ChainProgram(M)
            p.load(0, x)
            p.gemm(W0, a_slot=0, y_slot=1, act=True, pre_out=z0, res=res, beta=0.7)
            p.load(0, r16)
            p.gemm(W16, a_slot=0, y_slot=0, mul=1, alpha=0.9)
            p.gemm(W1, a_slot=0, y_slot=1, act=True, pre_out=z1, res=1, beta=0.7, out=y)
            p.gemm(W2, a_slot=1, y_slot=-1, out=y64)
            K.chain(p, mode=mode)
            outs.append((y.clone(), y64.clone(), z0.clone()))
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(o, outs[0]))
        print(f"chain program M={M} mode={mode}: 6 repeats bitwise equal = {same}", flush=True)

from conftest import SCALE_FILE  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402
from gemnet_pytorch_amd.training.data_container import DataContainer  # noqa: E402
from test_gpu_fullsize import FULL  # noqa: E402

for kind in ("T", "Q"):
    cfg = dict(FULL, triplets_only=kind == "T")
    torch.manual_seed(11)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
    model.requires_grad_(False)
    ds = make_dataset(8, 64, config=4)
    dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=cfg["triplets_only"])
    b = dc[list(range(8))]
    inputs = {k: v.to(dev) for k, v in b.items() if k not in ("E", "F")}
    for mode in (None, "bf16"):
        for overlap in (True, False):
            model.matmul_precision = mode
            model.overlap_output_blocks = overlap
            F = [model(inputs)[1].clone() for _ in range(4)]
            torch.cuda.synchronize()
            d = max(float((f - F[0]).abs().max()) for f in F[1:])
            print(f"GemNet-{kind} 8x64 mode={mode or 'split6'} side-stream={overlap}: max |dF| over 4 eager runs = {d:.3e} "
                  f"(mean|F| {float(F[0].abs().mean()):.3e})", flush=True)
