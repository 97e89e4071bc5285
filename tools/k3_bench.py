"""The two K3 GEMMs of the bilinear layer: (E,1024)x(1024->64) and its input gradient (E,64)x(64->1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit
dev = "cuda"
M = 18122
P = torch.randn(M, 1024, device=dev); W2t = torch.randn(64, 1024, device=dev) / 32   # forward: P @ W2  (B = W2^T (N=64,K=1024))
ref = P @ W2t.t()
print(f"fwd torch.mm: {timeit(lambda: torch.mm(P, W2t.t())):7.2f} us")
for c in (-1, 2, 3, 10, 11):
    out = K.gemm(P, W2t, cfg=c)
    t = timeit(lambda: K.gemm(P, W2t, cfg=c))
    print(f"fwd cfg {c:3d}: {t:7.2f} us ({2.0 * M * 64 * 1024 / t / 1e6:5.1f} TF) err {float((out - ref).abs().max()):.1e}")
g = torch.randn(M, 64, device=dev); W2 = torch.randn(1024, 64, device=dev) / 8         # backward: g @ W2^T  (B = W2 (N=1024,K=64))
ref = g @ W2.t()
print(f"bwd torch.mm: {timeit(lambda: torch.mm(g, W2.t())):7.2f} us")
for c in (-1, 0, 1, 7, 13, 14, 15):
    out = K.gemm(g, W2, cfg=c)
    t = timeit(lambda: K.gemm(g, W2, cfg=c))
    print(f"bwd cfg {c:3d}: {t:7.2f} us ({2.0 * M * 64 * 1024 / t / 1e6:5.1f} TF) err {float((out - ref).abs().max()):.1e}")
W2 = W2t.t().contiguous()     # (K = 1024, N = 64): the k-major 8-wave kernel (x @ B with B (K,N))
ref = P @ W2
out = K.gemm(P, W2, False, True)
t = timeit(lambda: K.gemm(P, W2, False, True))
print(f"fwd k-major (B as (K,N)): {t:7.2f} us ({2.0 * M * 64 * 1024 / t / 1e6:5.1f} TF) err {float((out - ref).abs().max()):.1e}")
