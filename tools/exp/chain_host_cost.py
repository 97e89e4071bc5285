"""Host time of one chain launch WITHOUT a GPU (the library call stubbed out): ChainProgram construction and the marshalling of
its gn_chain_args block in kernels.chain — what an eager (non-captured) step pays per chain program on the Python side.

    python tools/exp/chain_host_cost.py        # runs anywhere
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gemnet_pytorch_amd import _lib, kernels as K   # noqa: E402


class _Stub:
    def __getattr__(self, name):
        return lambda *a: 0


_lib._lib = _Stub()
_lib.load = lambda: _lib._lib
_lib.stream = lambda: ctypes.c_void_p(0)
K.stream = _lib.stream
K.require_device = lambda *t: None
K._mat = lambda t, cols=None: t.data_ptr()
M = 18122
x, y = torch.zeros(M, 128), torch.zeros(M, 128)
Ws = [torch.zeros(128, 128) for _ in range(5)]
packed = [torch.zeros(8 * 4 * 2 * 64 * 16, dtype=torch.uint8) for _ in range(5)]
for p in packed:
    p._gn_fmt = 1
zs = [torch.zeros(M, 128) for _ in range(5)]


def build():
    p = K.ChainProgram(M)
    p.load(0, x)
    cur, oth = 0, 1
    for i in range(5):
        p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=True, pre_out=zs[i], out=y if i == 4 else None, packed=packed[i])
        cur, oth = oth, cur
    return p


prog = build()
K.chain(prog, mode="h3")
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    K.chain(prog, mode="h3")
t1 = time.perf_counter()
for _ in range(n):
    build()
t2 = time.perf_counter()
print(f"LOAD + 5 GEMM ops: kernels.chain (checks + marshalling + call) {(t1 - t0) / n * 1e6:.1f} us, "
      f"ChainProgram construction {(t2 - t1) / n * 1e6:.1f} us")
