"""Prototype check: K1+K2+K3 of the bilinear layer in one launch (tools/exp/k3fused.hip) vs the two product launches."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import GraphPlan
from gemnet_pytorch_amd.kernels import ptr, stream
from tools.gemm_bench import timeit
import bench

exp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp")
so = os.path.join(exp, "libk3fused.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(exp, "k3fused.hip"), "-o", so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
lib.k3fused_fwd.argtypes = [vp] * 8 + [ctypes.c_int64, ctypes.c_float, vp]
cfg = {"cutoff": 5.0, "int_cutoff": 10.0, "triplets_only": True}
inputs, _ = bench.make_batch(cfg, 32, 32, 0, "cuda")
plan = GraphPlan(inputs, True)
sp = plan.trip
E, T = plan.n_edges, sp.size
g = torch.Generator(device="cuda").manual_seed(0)
Y = torch.randn(T, 7, device="cuda", generator=g)
x = torch.randn(E, 64, device="cuda", generator=g)
Bm = torch.randn(E, 7, 16, device="cuda", generator=g)
W2T = torch.randn(64, 1024, device="cuda", generator=g) / 32
alpha = 0.37


def product():
    Sm, P = K.bil_reduce_project(Y, x, Bm, sp)
    return Sm, K.gemm(P.reshape(-1, 1024), W2T, alpha=alpha)


def fused():
    Sm = torch.empty(E, 7, 64, device="cuda")
    out = torch.empty(E, 64, device="cuda")
    rc = lib.k3fused_fwd(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(Bm), ptr(W2T), ptr(Sm), ptr(out), E,
                         alpha, stream())
    assert rc == 0, rc
    return Sm, out


Sm0, o0 = product()
Sm1, o1 = fused()
torch.cuda.synchronize()
print(f"E={E}: Sm max err {float((Sm0 - Sm1).abs().max()):.2e}   out max err {float((o0 - o1).abs().max()):.2e} (|out| max {float(o0.abs().max()):.2f})")
print(f"product (K1+K2 launch, K3 GEMM): {timeit(product):7.2f} us")
print(f"fused   (one launch)           : {timeit(fused):7.2f} us")
