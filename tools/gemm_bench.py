"""GEMM microbenchmark on the GPU box: tile variants of our f32-MFMA kernel vs torch.mm per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K

dev = "cuda"


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters // 20):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (iters // 20 * 20)  # us


def main():
    CFGS = {0: "64x128 2x2 k32", 1: "32x128 1x4 k32", 6: "64x128 2x2 k64", 7: "32x128 1x4 k64", 8: "64x128 2x4 k32",
            9: "64x128 2x4 k64", 12: "128x128 4x2 k32", 13: "128x128 4x4 k32", 14: "32x128 8w 16x16 k32", 15: "32x128 8w 16x16 k64",
            2: "64x64 2x2 k32", 3: "32x64 1x2 k32", 10: "32x64 1x2 k64", 11: "64x64 2x2 k64", 4: "128x32 4x1", 5: "32x32 1x1"}
    for M, N, Kd in [(18122, 128, 128), (1024, 128, 128), (18122, 128, 64), (9061, 128, 128), (65536, 128, 128)]:
        A = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev); Z = torch.randn(M, Kd, device=dev)
        Wt = W.t().contiguous()
        tt = timeit(lambda: torch.mm(A, Wt))
        ref = A @ Wt
        print(f"shape {(M, N, Kd)}: torch.mm {tt:.2f} us")
        cands = [c for c in CFGS if (N > 64 and c in (0, 1, 7, 13, 14, 15)) or (32 < N <= 64 and c in (2, 3, 10, 11))
                 or (N <= 32 and c in (4, 5))]
        for c in cands:
            out = K.gemm(A, W, cfg=c)
            err = float((out - ref).abs().max())
            t0 = timeit(lambda: K.gemm(A, W, cfg=c))
            t1 = timeit(lambda: K.gemm(A, W, act=True, pre_out=True, cfg=c))
            t2 = timeit(lambda: K.gemm(A, W, a_dact_pre=Z, cfg=c))
            print(f"    cfg {c:2d} {CFGS[c]:>16s}: plain {t0:7.2f} us ({2.0 * M * N * Kd / t0 / 1e6:5.1f} TF)  act+pre {t1:7.2f}  dact {t2:7.2f}  maxerr {err:.1e}")


if __name__ == "__main__":
    main()
