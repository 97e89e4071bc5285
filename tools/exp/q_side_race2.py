"""GPU diagnosis, part 2: GemNet-Q 8x64, bf16, output blocks on the side stream.  Every fused aggregation launch of one
forward is checked against the GEMM + segmented-sum composition of the SAME inputs on the SAME stream, and checksums of
its inputs / output are kept per run: which launch differs between runs, and is it the kernel or its inputs?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import SCALE_FILE  # noqa: E402
from gemnet_pytorch_amd import kernels as K  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402
from gemnet_pytorch_amd.training.data_container import DataContainer  # noqa: E402
from test_gpu_fullsize import FULL  # noqa: E402

dev = "cuda"
cfg = dict(FULL, triplets_only=False)
torch.manual_seed(11)
model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
ds = make_dataset(8, 64, config=4)
dc = DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=False)
b = dc[list(range(8))]
inputs = {k: v.to(dev) for k, v in b.items() if k not in ("E", "F")}
model.matmul_precision = "bf16"
orig = K.rbf_aggregate_fwd
log = []


def wrapped(m, rbf, W, perm, seg_off, n_atoms, scale):
    out = orig(m, rbf, W, perm, seg_off, n_atoms, scale)
    x = (m.double() * (rbf.double() @ W.double().t())) * scale
    ref = torch.zeros(n_atoms, m.shape[1], dtype=torch.float64, device=m.device)
    idx = torch.repeat_interleave(torch.arange(n_atoms, device=m.device), (seg_off[1:] - seg_off[:-1]).long())
    src = x if perm is None else x[perm.long()]
    ref.index_add_(0, idx, src)
    st = torch.cuda.current_stream().cuda_stream
    log[-1].append((st, m.double().sum(), rbf.double().sum(), out.double().sum(), (out.double() - ref).abs().max()))
    return out


K.rbf_aggregate_fwd = wrapped
Es = []
for rep in range(5):
    log.append([])
    with torch.no_grad():
        pass
    E, F = model(inputs)
    Es.append(E.detach().clone())
torch.cuda.synchronize()
streams = sorted({int(r[0]) for r in log[0]})
print("streams seen:", streams, " launches per forward:", len(log[0]))
for i in range(len(log[0])):
    vals = [[float(v) for v in log[r][i][1:]] for r in range(5)]
    msum = {v[0] for v in vals}
    rsum = {v[1] for v in vals}
    osum = {v[2] for v in vals}
    err = max(v[3] for v in vals)
    print(f"launch {i:2d} stream {streams.index(int(log[0][i][0]))}: inputs m {'SAME' if len(msum) == 1 else 'DIFFER'} "
          f"rbf {'SAME' if len(rsum) == 1 else 'DIFFER'}  output {'SAME' if len(osum) == 1 else 'DIFFER'}  "
          f"max |kernel - composition| over runs = {err:.3e}")
print("E run-to-run max diff:", max(float((e - Es[0]).abs().max()) for e in Es[1:]))
