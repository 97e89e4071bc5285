import os, sys, torch
sys.path.insert(0, "/root/repo")
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit
for M, N, Kd in [(18122, 64, 1024), (18122, 1024, 64), (18122, 128, 128), (18122, 64, 128), (18122, 128, 64), (1024, 128, 128)]:
    A = torch.randn(M, Kd, device="cuda"); W = torch.randn(N, Kd, device="cuda"); Wt = W.t().contiguous()
    t_mine = timeit(lambda: K.gemm(A, W))
    t_mm = timeit(lambda: torch.mm(A, Wt))
    t_mm2 = timeit(lambda: torch.mm(A, W.t()))
    print(f"{(M,N,Kd)}: mine {t_mine:.1f} us   torch.mm(A, Wt contiguous) {t_mm:.1f}   torch.mm(A, W.t()) {t_mm2:.1f}")
