"""Chain-kernel microbenchmark (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit

dev = "cuda"
for M in (1024, 18122):
    x = torch.randn(M, 128, device=dev)
    Ws = [torch.randn(128, 128, device=dev) / 11 for _ in range(6)]
    zs = [torch.empty(M, 128, device=dev) for _ in range(6)]
    y = torch.empty(M, 128, device=dev)

    def prog(n_gemm, pre=True, act=True, load=True, store=True):
        p = K.ChainProgram(M)
        if load:
            p.load(0, x)
        cur, oth = 0, 1
        for i in range(n_gemm):
            p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=act, pre_out=zs[i] if pre else None,
                   out=y if (store and i == n_gemm - 1) else None)
            cur, oth = oth, cur
        return p

    print(f"M={M}")
    for n in (1, 2, 5):
        for kw in (dict(), dict(pre=False), dict(pre=False, act=False)):
            p = prog(n, **kw)
            t = timeit(lambda: K.chain(p))
            print(f"   chain of {n} GEMM {kw}: {t:8.2f} us  ({t / n:6.2f} us per GEMM)")
    t = timeit(lambda: K.gemm(x, Ws[0], act=True, pre_out=True))
    print(f"   single gemm_nt_pipe act+pre: {t:8.2f} us")
    p = K.ChainProgram(M); p.load(0, x); p.store(0, y)
    print(f"   load+store only: {timeit(lambda: K.chain(p)):8.2f} us")
