"""Stand-alone time of the fused geometry + basis kernels at the headline shapes (GEMNET_HIP_LIB selects the library for A/B).
   PYTHONPATH=. python tools/exp/basis_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from gemnet_pytorch_amd import _lib  # noqa: E402
from gemnet_pytorch_amd import kernels as K  # noqa: E402
from gemnet_pytorch_amd.graph import GraphPlan  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(B.GEMNET_T)
inputs, _ = B.make_batch(cfg, 32, 32, first=0, device=dev)
plan = GraphPlan.from_inputs(inputs, True).warm()
model = GemNet(**cfg, scale_file=B.SCALE_FILE).to(dev)
b3 = model.cbf_basis3
R = inputs["R"]
E = plan.n_edges
g = torch.Generator(device="cuda").manual_seed(1)


def t(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


freq = model.rbf_basis.frequencies.detach()
D, V, rbf, rad = K.edge_basis_fwd(R, plan.id_c.idx32, plan.id_a.idx32, freq, b3.z_ln, b3.n_ln, b3.cutoff, b3.p)
g_rbf, g_rad = torch.randn(rbf.shape, device=dev, generator=g), torch.randn(rad.shape, device=dev, generator=g)
print(os.path.basename(_lib.LIB_PATH), "checksums", float(rbf.double().sum()), float(rad.double().sum()))
print("edge_basis_fwd %.1f us" % t(lambda: K.edge_basis_fwd(R, plan.id_c.idx32, plan.id_a.idx32, freq, b3.z_ln, b3.n_ln, b3.cutoff, b3.p)))
W = K.edge_basis_bwd(None, g_rbf, g_rad, R, plan.id_c.idx32, plan.id_a.idx32, freq, b3.z_ln, b3.n_ln, b3.cutoff, b3.p)
print("   bwd checksum", float(W.double().abs().sum()))
print("edge_basis_bwd %.1f us" % t(lambda: K.edge_basis_bwd(None, g_rbf, g_rad, R, plan.id_c.idx32, plan.id_a.idx32, freq, b3.z_ln, b3.n_ln, b3.cutoff, b3.p)))
Y, _ = K.trip_basis_fwd(R, plan.t_c.idx32, plan.t_a.idx32, plan.t_b.idx32, 7)
gY = torch.randn(Y.shape, device=dev, generator=g)
print("trip_basis_fwd %.1f us" % t(lambda: K.trip_basis_fwd(R, plan.t_c.idx32, plan.t_a.idx32, plan.t_b.idx32, 7)))
print("trip_basis_bwd %.1f us" % t(lambda: K.trip_basis_bwd(gY, R, plan.t_c.idx32, plan.t_a.idx32, plan.t_b.idx32)))
