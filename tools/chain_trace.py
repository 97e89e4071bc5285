"""Phase timing inside gn_chain_f32 (diagnosis build with -DGN_CHAIN_TRACE): shader-clock stamps per GEMM op."""
import ctypes, os, sys, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gemnet_pytorch_amd", "csrc")
TRACE_LIB = os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_trace.so")
if not os.path.exists(TRACE_LIB) or "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-DGN_CHAIN_TRACE", "-I", os.path.join(ROOT, "include")] + sorted(glob.glob(CSRC + "/*.hip"))
                          + ["-o", TRACE_LIB])
    if "--build" in sys.argv:
        sys.exit(0)
import numpy as np, torch
from gemnet_pytorch_amd import _lib
_lib.LIB_PATH = TRACE_LIB
from gemnet_pytorch_amd import kernels as K
lib = _lib.load()
lib.gn_chain_trace_read.argtypes = [ctypes.c_void_p]
for M in (1024, 18122):
    for pre in (0, 1):
        n = 5
        x = torch.randn(M, 128, device="cuda")
        Ws = [torch.randn(128, 128, device="cuda") / 11 for _ in range(n)]
        zs = [torch.empty(M, 128, device="cuda") for _ in range(n)]
        y = torch.empty(M, 128, device="cuda")
        p = K.ChainProgram(M); p.load(0, x)
        cur, oth = 0, 1
        for i in range(n):
            p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=bool(pre), pre_out=zs[i] if pre else None, out=y if i == n - 1 else None)
            cur, oth = oth, cur
        for _ in range(5):
            K.chain(p)
        torch.cuda.synchronize()
        buf = np.zeros((2, 20, 8), dtype=np.uint64)
        lib.gn_chain_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
        print(f"M={M} pre/act={pre}: cycles per phase [wait-W, mfma, epilogue, barrier | op total] (block 0 / block 100)")
        for b in range(2):
            t = buf[b].astype(np.int64)
            for oi in range(1, n + 1):
                d = [int(t[oi, i + 1] - t[oi, i]) for i in range(4)]
                gap = int(t[oi, 0] - t[oi - 1, 4]) if oi > 1 else 0
                print(f"   blk{b*100:3d} op{oi}: gap {gap:6d}  {d}  total {int(t[oi,4]-t[oi,0])}   epi: start+{int(t[oi,6]-t[oi,2])} stages {int(t[oi,5]-t[oi,6])} lds/out {int(t[oi,3]-t[oi,5])}")
