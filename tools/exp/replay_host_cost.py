"""Host time of one hipGraphLaunch of the forward+force graph (GemNet-T headline shape / GemNet-Q): a single launch on an idle
device, and back-to-back launches (does the call wait for the previous launch of the same executable graph?).
   PYTHONPATH=. python tools/exp/replay_host_cost.py [T|Q]"""
import sys
import time

import torch

import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.runtime import ForceGraphs

which = sys.argv[1] if len(sys.argv) > 1 else "T"
dev = torch.device("cuda")
cfg = dict(bench.GEMNET_T, triplets_only=(which == "T"))
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
fg = ForceGraphs(model, [inputs])
g = fg.graphs[0] if hasattr(fg, "graphs") else None
rep = (lambda: fg.replay()) if g is None else (lambda: g.replay())
for _ in range(5):
    rep()
torch.cuda.synchronize()
single = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    single.append((t1 - t0, t2 - t0))
n = 30
torch.cuda.synchronize()
t0 = time.perf_counter()
calls = []
for _ in range(n):
    a = time.perf_counter()
    rep()
    calls.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"GemNet-{which}: single launch on an idle device: host {1e3 * min(s[0] for s in single):.3f} ms of {1e3 * min(s[1] for s in single):.3f} ms "
      f"to completion; {n} back-to-back launches: host {1e3 * (t1 - t0) / n:.3f} ms per call (first {1e3 * calls[0]:.3f}, median "
      f"{1e3 * sorted(calls)[n // 2]:.3f}), {1e3 * (t2 - t0) / n:.3f} ms per step")
