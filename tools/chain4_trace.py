"""Phase timing inside the row-resident chain kernel (csrc/chain4.hip; diagnosis build with -DGN_CHAIN_TRACE): shader-clock
stamps per op of compute wave 0 and of the loader wave of one workgroup.
    python tools/chain4_trace.py [--adj]        GN_TRACE_DEFS="-DGN4_EXP=1" for experiment builds"""
import ctypes, os, sys, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gemnet_pytorch_amd", "csrc")
DEFS = os.environ.get("GN_TRACE_DEFS", "").split()
os.makedirs(os.path.join(ROOT, "tools", "exp", "bin"), exist_ok=True)
TRACE_LIB = os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_trace4%s.so" % "".join(d.replace("-D", "_").replace("=", "") for d in DEFS))
if not os.path.exists(TRACE_LIB) or "--build" in sys.argv:
    import shutil
    import __graft_entry__ as ge
    base = os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_trace4.so")
    if DEFS and os.path.isdir(base + ".obj") and not os.path.isdir(TRACE_LIB + ".obj"):
        # an experiment build differs from the plain trace build in chain4.hip only: reuse the other objects
        shutil.copytree(base + ".obj", TRACE_LIB + ".obj")
        os.remove(os.path.join(TRACE_LIB + ".obj", "chain4.o"))
    ge.build_hip_library(lib=TRACE_LIB, extra=["-DGN_CHAIN_TRACE"] + DEFS, objdir=TRACE_LIB + ".obj")
    if "--build" in sys.argv:
        sys.exit(0)
import numpy as np, torch
from gemnet_pytorch_amd import _lib
_lib.LIB_PATH = TRACE_LIB
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit
lib = _lib.load()
lib.gn_chain4_trace_read.argtypes = [ctypes.c_void_p]
K.CHAIN_LAYOUT = "row"
ADJ = "--adj" in sys.argv
for M in (18122, 1024):
    for pre in (0, 1):
        n = 5
        x = torch.randn(M, 128, device="cuda")
        Ws = [torch.randn(128, 128, device="cuda") / 11 for _ in range(n)]
        Wp = [K.pack_weight_split(w, fmt=2) for w in Ws]
        zs = [torch.randn(M, 128, device="cuda") for _ in range(n)]
        zs2 = [torch.empty(M, 128, device="cuda") for _ in range(n)]
        y = torch.empty(M, 128, device="cuda")
        p = K.ChainProgram(M); p.load(0, x)
        cur, oth = 0, 1
        for i in range(n):
            if ADJ:
                p.gemm(Ws[i], a_slot=cur, y_slot=oth, mul=zs[i], mul_mode=2, pre_out=zs2[i] if pre else None,
                       out=y if i == n - 1 else None, packed=Wp[i])
            else:
                p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=bool(pre), pre_out=zs[i] if pre else None, pre_deriv=bool(pre),
                       out=y if i == n - 1 else None, packed=Wp[i])
            cur, oth = oth, cur
        for _ in range(5):
            K.chain(p, mode="h3")
        torch.cuda.synchronize()
        buf = np.zeros((2, 21, 8), dtype=np.uint64)
        lib.gn_chain4_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
        t = buf[0].astype(np.int64); l = buf[1].astype(np.int64)
        us = timeit(lambda: K.chain(p, mode="h3"), iters=50)
        print(f"M={M} pre/act={pre} adj={ADJ}: {us:.1f} us per launch; compute wave 0 of one workgroup, cycles")
        print(f"   LOAD: {int(t[1,0]-t[0,0])};  first op start -> kernel end {int(t[20,0]-t[0,0])}")
        for oi in range(1, n + 1):
            nxt = t[oi + 1, 0] if oi < n else t[20, 0]
            print(f"   op{oi}: desc+split {int(t[oi,1]-t[oi,0]):6d}  barrier {int(t[oi,2]-t[oi,1]):6d}  mfma {int(t[oi,3]-t[oi,2]):6d}  "
                  f"epilogue {int(t[oi,4]-t[oi,3]):6d}  -> next op {int(nxt-t[oi,4]):5d} | total {int(nxt-t[oi,0]):6d}"
                  f"   || loader g{oi-1}: issue {int(l[oi-1,1]-l[oi-1,0]):6d} wait {int(l[oi-1,2]-l[oi-1,1]):6d} barrier {int(l[oi-1,3]-l[oi-1,2]):6d}")
