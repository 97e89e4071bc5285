"""Prototype check: the adjoint of the bilinear tail in one launch
(tools/exp/k3fused_bwd.hip: dP kept in LDS) vs the two product launches gn_gemm_f32 + gn_bil_project_bwd_f32(dY = NULL)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import GraphPlan
from gemnet_pytorch_amd.kernels import ptr, stream
from tools.gemm_bench import timeit
import bench

exp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp")
so = os.path.join(exp, "libk3fused_bwd.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(exp, "k3fused_bwd.hip"), "-o", so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
lib.k3fused_bwd.argtypes = [vp] * 6 + [ctypes.c_int64, ctypes.c_float, vp]
cfg = {"cutoff": 5.0, "int_cutoff": 10.0, "triplets_only": True}
inputs, _ = bench.make_batch(cfg, 32, 32, 0, "cuda")
plan = GraphPlan(inputs, True)
sp = plan.trip
E = plan.n_edges
gen = torch.Generator(device="cuda").manual_seed(0)
g = torch.randn(E, 64, device="cuda", generator=gen)
W2 = torch.randn(1024, 64, device="cuda", generator=gen) / 8      # (I*C, O)
Sm = torch.randn(E, 7, 64, device="cuda", generator=gen)
Bm = torch.randn(E, 7, 16, device="cuda", generator=gen)
x = torch.randn(E, 64, device="cuda", generator=gen)
alpha = 0.37


def product():
    dP = K.gemm(g, W2, alpha=alpha).reshape(-1, 16, 64)           # g @ W2^T
    gB, dSm, _ = K.bil_project_bwd(dP, Sm, Bm, x, sp, want_dY=False)
    return gB, dSm


def fused():
    gB = torch.empty(E, 7, 16, device="cuda")
    dSm = torch.empty(E, 7, 64, device="cuda")
    rc = lib.k3fused_bwd(ptr(g), ptr(W2), ptr(Sm), ptr(Bm), ptr(gB), ptr(dSm), E, alpha, stream())
    assert rc == 0, rc
    return gB, dSm


a, b = product(), fused()
torch.cuda.synchronize()
for name, u, v in zip(("gB", "dSm"), a, b):
    print(f"{name}: max err {float((u - v).abs().max()):.2e} (|ref| max {float(u.abs().max()):.2f})")
print(f"product (K3-adjoint GEMM + project_bwd): {timeit(product):7.2f} us")
print(f"fused   (one launch)                   : {timeit(fused):7.2f} us")
