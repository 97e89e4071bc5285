"""A/B of the fused edge -> atom aggregation (gn_rbf_aggregate_fwd/bwd_f32): the product library (round-5 kernels: scalar loads of
the wave-uniform operands, W through LDS, one resident round + next-edge prefetch in the adjoint) against a library built with
-DGN_AGG_V1 (the round-2 kernels; tools/exp/bin/libgemnet_hip_aggv1.so, or any libgemnet_hip_agg*.so there): bitwise comparison
+ stand-alone time at the headline shapes.   PYTHONPATH=. python tools/exp/agg_v2_bench.py   (-> profiles/r5_agg_v2.txt)"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from gemnet_pytorch_amd import _lib  # noqa: E402
from gemnet_pytorch_amd.graph import GraphPlan  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(B.GEMNET_T)
inputs, _ = B.make_batch(cfg, 32, 32, first=0, device=dev)
plan = GraphPlan.from_inputs(inputs, True).warm()
E, A = plan.n_edges, plan.n_atoms
g = torch.Generator(device="cuda").manual_seed(1)
m, rbf, W = (torch.randn(E, 128, device=dev, generator=g), torch.randn(E, 16, device=dev, generator=g),
             torch.randn(128, 16, device=dev, generator=g) / 4)
go = torch.randn(A, 128, device=dev, generator=g)
ida = plan.id_a.idx32
libs = {"product": _lib.LIB_PATH}
for p in sorted(glob.glob(os.path.join(ROOT, "tools", "exp", "bin", "libgemnet_hip_agg*.so"))):
    libs[os.path.basename(p)[len("libgemnet_hip_"):-3]] = p
vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


ref = None
for name, path in libs.items():
    lib = ctypes.CDLL(path)
    f = lib.gn_rbf_aggregate_bwd_f32
    f.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, ci, ci, cf, ci, vp]
    f.restype = ci
    st = vp(torch.cuda.current_stream().cuda_stream)
    gm, gr = torch.empty_like(m), torch.empty_like(rbf)
    run = torch.zeros_like(m)

    def call(gm_=gm, gr_=gr, acc=0):
        rc = f(go.data_ptr(), m.data_ptr(), rbf.data_ptr(), W.data_ptr(), ida.data_ptr(), gm_.data_ptr() if gm_ is not None else None,
               gr_.data_ptr() if gr_ is not None else None, E, 128, 16, 0.5, acc, st)
        assert rc == 0, rc
    call()
    torch.cuda.synchronize()
    out = (gm.clone(), gr.clone())
    if ref is None:
        ref = out
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    t_both = timeit(call)
    t_m = timeit(lambda: call(gm, None))
    t_acc = timeit(lambda: call(run, gr, 1))
    ff = lib.gn_rbf_aggregate_fwd_f32
    ff.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, ci, cf, vp]
    ff.restype = ci
    perm, seg = plan.id_a.csr
    fo = torch.empty(A, 128, device=dev)

    def fwd():
        rc = ff(m.data_ptr(), rbf.data_ptr(), W.data_ptr(), perm.data_ptr() if perm is not None else None, seg.data_ptr(),
                fo.data_ptr(), A, 128, 16, 0.5, st)
        assert rc == 0, rc
    fwd()
    torch.cuda.synchronize()
    if name == "product":
        ref_f = fo.clone()
    same_f = torch.equal(fo, ref_f)
    t_f = timeit(fwd)
    print(f"{name:14s}: adjoint bit-identical to the product kernel: {same}, forward: {same_f};  forward {t_f:6.1f} us;  adjoint g_m + "
          f"g_rbf {t_both:6.1f} us, g_m only {t_m:6.1f} us, accumulate m {t_acc:6.1f} us")
