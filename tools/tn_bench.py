"""Time the weight-gradient shaped product (A^T B, K = edges/atoms) : new split-K kernel vs torch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemnet_pytorch_amd.kernels as K

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for Kd, M, N in [(1024, 128, 128), (17800, 128, 128), (17800, 128, 16), (17800, 128, 64), (17800, 1024, 64), (320000, 64, 64)]:
    A = torch.randn(Kd, M, device="cuda"); B = torch.randn(Kd, N, device="cuda")
    row = [f"K={Kd} M={M} N={N}", f"torch {t(lambda: A.t() @ B):.1f}us", f"auto(s={K._lib.load().gn_gemm_tn_splitk(M, N, Kd)}) {t(lambda: K.gemm_tn(A, B)):.1f}us"]
    for s in (1, 8, 32, 64, 128, 256):
        row.append(f"s{s} {t(lambda: K.gemm_tn(A, B, splitk=s)):.1f}")
    print("  ".join(row), flush=True)
