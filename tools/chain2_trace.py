"""Phase timing inside gn_chain_split_f32 (diagnosis build with -DGN_CHAIN_TRACE): shader-clock stamps per op.

    python tools/chain2_trace.py [--quick [--small]] [--modes=split6,h3]   all modes / only split6 at M = 18122 (M = 1024)
    GN_TRACE_DEFS="-DGN_EXP=1" python tools/chain2_trace.py --quick   MFMA phase without its LDS reads (2: reads only)"""
import ctypes, os, sys, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gemnet_pytorch_amd", "csrc")
DEFS = os.environ.get("GN_TRACE_DEFS", "").split()     # extra -D flags (experiments), e.g. GN_TRACE_DEFS="-DGN_EXP=1"
TRACE_LIB = os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_trace%s.so" % "".join(d.replace("-D", "_").replace("=", "") for d in DEFS))
QUICK = "--quick" in sys.argv
if not os.path.exists(TRACE_LIB) or "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-DGN_CHAIN_TRACE", "-I", os.path.join(ROOT, "include")] + DEFS + sorted(glob.glob(CSRC + "/*.hip"))
                          + ["-o", TRACE_LIB])
    if "--build" in sys.argv:
        sys.exit(0)
import numpy as np, torch
from gemnet_pytorch_amd import _lib
_lib.LIB_PATH = TRACE_LIB
from gemnet_pytorch_amd import kernels as K
lib = _lib.load()
lib.gn_chain2_trace_read.argtypes = [ctypes.c_void_p]
MODES = [a.split("=")[1].split(",") for a in sys.argv if a.startswith("--modes=")]
for mode in (MODES[0] if MODES else (("split6",) if QUICK else ("split6", "bf16"))):
  for M in (((1024,) if "--small" in sys.argv else (18122,)) if QUICK else (1024, 18122)):
    for pre in ((0, 1) if (MODES or not QUICK) else (0,)):
        n = 5
        x = torch.randn(M, 128, device="cuda")
        Ws = [torch.randn(128, 128, device="cuda") / 11 for _ in range(n)]
        Wp = [K.pack_weight_split(w, fmt=K.SPLIT_FORMAT[mode]) for w in Ws]
        zs = [torch.randn(M, 128, device="cuda") for _ in range(n)]
        zs2 = [torch.empty(M, 128, device="cuda") for _ in range(n)]
        y = torch.empty(M, 128, device="cuda")
        p = K.ChainProgram(M); p.load(0, x)
        cur, oth = 0, 1
        ADJ = "--adj" in sys.argv     # adjoint-like ops: a global f'(z) factor on every GEMM, pre-activation adjoint stored
        for i in range(n):
            if ADJ:
                p.gemm(Ws[i], a_slot=cur, y_slot=oth, mul=zs[i], mul_mode=2, pre_out=zs2[i] if pre else None,
                       out=y if i == n - 1 else None, packed=Wp[i])
            else:
                p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=bool(pre), pre_out=zs[i] if pre else None, out=y if i == n - 1 else None,
                       packed=Wp[i])
            cur, oth = oth, cur
        for _ in range(5):
            K.chain(p, mode=mode)
        torch.cuda.synchronize()
        buf = np.zeros((2, 20, 8), dtype=np.uint64)
        lib.gn_chain2_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
        print(f"[{mode}] M={M} pre/act={pre}: cycles per phase [wait-W, mfma, epilogue, barrier | op total] (wave 0 / wave 7 of block 100)")
        t = buf[0].astype(np.int64)
        print(f"   prologue (entry -> op table staged, first barrier): {int(t[19, 1] - t[19, 0])}; entry -> first op {int(t[0, 0] - t[19, 0])}; "
              f"entry -> end of the last op {int(t[n, 4] - t[19, 0])}")
        for b in range(2):
            t = buf[b].astype(np.int64)
            print(f"   wave{b*7} LOAD: work {int(t[0,3]-t[0,0])} barrier {int(t[0,4]-t[0,3])}")
            for oi in range(1, n + 1):
                d = [int(t[oi, i + 1] - t[oi, i]) for i in range(4)]
                gap = int(t[oi, 0] - t[oi - 1, 4])
                if b == 1:
                    t0 = buf[0].astype(np.int64)
                    d.append(("vs wave0 at stamps", [int(t[oi, i] - t0[oi, i]) for i in range(5)]))
                sub = [int(t[oi, 5] - t[oi, 2]), int(t[oi, 6] - t[oi, 5]), int(t[oi, 7] - t[oi, 6]), int(t[oi, 3] - t[oi, 7])]
                print(f"   wave{b*7} op{oi}: gap {gap:6d}  {d}  total {int(t[oi,4]-t[oi,0])}   epilogue = [to act done, mul/res/store, "
                      f"split + LDS write, row scales] {sub}")
