"""Chain-kernel microbenchmark (GPU box): f32-MFMA chain (csrc/chain.hip) vs split-operand bf16 chain (csrc/chain2.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit

dev = "cuda"
for M in (1024, 18122):
    x = torch.randn(M, 128, device=dev)
    Ws = [torch.randn(128, 128, device=dev) / 11 for _ in range(6)]
    Wp = [K.pack_weight_split(w) for w in Ws]
    zs = [torch.empty(M, 128, device=dev) for _ in range(6)]
    y = torch.empty(M, 128, device=dev)
    skip = torch.randn(M, 128, device=dev)

    def prog(n_gemm, pre=True, act=True, load=True, store=True):
        p = K.ChainProgram(M)
        if load:
            p.load(0, x)
        cur, oth = 0, 1
        for i in range(n_gemm):
            p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=act, pre_out=zs[i] if pre else None,
                   out=y if (store and i == n_gemm - 1) else None, packed=Wp[i])
            cur, oth = oth, cur
        return p

    def residual_stack():
        """dense + 2 residual layers (+ skip): the edge stack of an interaction block"""
        p = K.ChainProgram(M)
        p.load(0, x)
        p.gemm(Ws[0], a_slot=0, y_slot=1, act=True, pre_out=zs[0], res=skip, beta=0.7, packed=Wp[0])
        cur, oth = 1, 0
        for k in range(2):
            p.gemm(Ws[1 + 2 * k], a_slot=cur, y_slot=oth, act=True, pre_out=zs[1 + 2 * k], packed=Wp[1 + 2 * k])
            p.gemm(Ws[2 + 2 * k], a_slot=oth, y_slot=cur, act=True, pre_out=zs[2 + 2 * k], res=cur, beta=0.7,
                   res2=skip if k == 0 else None, beta2=0.7, out=y if k == 1 else None, packed=Wp[2 + 2 * k])
        return p

    print(f"M={M}")
    for mode in ("f32", "split6", "split3", "bf16"):
        for n in (1, 2, 5):
            for kw in (dict(), dict(pre=False, act=False)):
                p = prog(n, **kw)
                t = timeit(lambda: K.chain(p, mode=mode))
                print(f"   [{mode:6s}] chain of {n} GEMM {kw}: {t:8.2f} us  ({t / n:6.2f} us per GEMM)")
        p = residual_stack()
        t = timeit(lambda: K.chain(p, mode=mode))
        print(f"   [{mode:6s}] dense + 2 residual layers (5 GEMM, pre-activations, skips): {t:8.2f} us")
        p = K.ChainProgram(M); p.load(0, x); p.store(0, y)
        print(f"   [{mode:6s}] load+store only: {timeit(lambda: K.chain(p, mode=mode)):8.2f} us")
    t = timeit(lambda: K.gemm(x, Ws[0], act=True, pre_out=True))
    print(f"   single gemm_nt_pipe act+pre: {t:8.2f} us")
