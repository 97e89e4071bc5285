"""GPU box: the fused triplet bilinear kernels (gn_bil_fused_fwd_f32 / gn_bil_fused_bwd_f32) on the bench batch, stand-alone
(GEMNET_HIP_LIB selects an experiment build of the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import GraphPlan
from tools.gemm_bench import timeit

cfg = {"cutoff": 5.0, "int_cutoff": 10.0, "triplets_only": True}
inputs, _ = bench.make_batch(cfg, 32, 32, 0, "cuda")
sp = GraphPlan(inputs, True).trip
E, T = sp.n_reduce, sp.size
g = torch.Generator(device="cuda").manual_seed(0)
Y = torch.randn(T, 7, device="cuda", generator=g)
x = torch.randn(E, 64, device="cuda", generator=g)
B = torch.randn(E, 7, 16, device="cuda", generator=g)
W2T = torch.randn(64, 1024, device="cuda", generator=g) / 32
W2 = W2T.t().contiguous()
Wp_f = K.pack_weight_split(W2T, fmt=1)
Wp_b = K.pack_weight_split(W2, fmt=1)
Sm, out = K.bil_fused_fwd(Y, x, B, W2T, sp, W2T_planes=Wp_f)
go = torch.randn(E, 64, device="cuda", generator=g)
gB, dSm = K.bil_fused_bwd(go, W2, Sm, B, W2_planes=Wp_b)
print(f"E={E} T={T}: checksum fwd {float(out.double().sum()):.6f} / {float(Sm.double().sum()):.6f}  bwd {float(gB.double().sum()):.6f} / {float(dSm.double().sum()):.6f}")
for _ in range(2):
    print(f"   bil_fused_fwd {timeit(lambda: K.bil_fused_fwd(Y, x, B, W2T, sp, W2T_planes=Wp_f), iters=200):7.2f} us   "
          f"bil_fused_bwd {timeit(lambda: K.bil_fused_bwd(go, W2, Sm, B, W2_planes=Wp_b), iters=200):7.2f} us")
