"""What bounds the atom-grouped triplet adjoint?  Times the product kernels and the variants of
tools/exp/bilt_variants.hip (built here: hipcc ... -o tools/exp/libbilt.so) on the bench batch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import GraphPlan, SegmentPlan
from gemnet_pytorch_amd.kernels import ptr, stream
from tools.gemm_bench import timeit
import bench

_exp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp")
if not os.path.exists(os.path.join(_exp, "libbilt.so")):   # cross-compiles without a GPU: build before `gpurun`
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(_exp, "bilt_variants.hip"), "-o", os.path.join(_exp, "libbilt.so")])
lib = ctypes.CDLL(os.path.join(_exp, "libbilt.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lib.bilt_grouped.argtypes = [i32, i32] + [vp] * 8 + [i32, i32, vp]
lib.bilt_staged.argtypes = [i32] + [vp] * 8 + [i32, i32, i32, vp]

cfg = {"cutoff": 5.0, "int_cutoff": 10.0, "triplets_only": True}
inputs, _ = bench.make_batch(cfg, 32, 32, 0, "cuda")
plan = GraphPlan(inputs, True)
sp = plan.trip
rows, off, kseg, rposT, max_rows = sp.groups
permT, segT = sp.expand.csr
E, T, A = plan.n_edges, sp.size, plan.n_atoms
print(f"E={E} T={T} atoms={A} max_rows={max_rows}")
g = torch.Generator(device="cuda").manual_seed(0)
Y = torch.randn(T, 7, device="cuda", generator=g)
D = torch.randn(E, 7, 64, device="cuda", generator=g)
dx = torch.empty(E, 64, device="cuda")

plain = SegmentPlan(inputs["id3_reduce_ca"], inputs["id3_expand_ba"], E, E)
ref = K.bil_reduce_t(Y, D, plain)
print(f"ungrouped product kernel : {timeit(lambda: K.bil_reduce_t(Y, D, plain)):7.2f} us")
print(f"grouped product kernel   : {timeit(lambda: K.bil_reduce_t(Y, D, sp)):7.2f} us   err {float((K.bil_reduce_t(Y, D, sp) - ref).abs().max()):.2e}")


def grouped(mode, nt, lds_rows):
    rc = lib.bilt_grouped(mode, nt, ptr(Y), ptr(D), ptr(rows), ptr(off), ptr(kseg), ptr(permT), ptr(rposT), ptr(dx), A,
                          lds_rows, stream())
    assert rc == 0, rc


for nt in (1024, 512, 256):
    for lr in (max_rows, 45, 64, 88):
        t = timeit(lambda: grouped(0, nt, lr))
        err = float((dx - ref).abs().max())
        print(f"grouped nt={nt:4d} lds_rows={lr:3d} ({lr * 1792 / 1024:5.1f} KB): {t:7.2f} us  err {err:.1e}")
for mode, name in ((1, "no compute"), (2, "no tile fill"), (3, "no Y fetch")):
    print(f"grouped nt=1024 {name:13s}: {timeit(lambda: grouped(mode, 1024, max_rows)):7.2f} us")

# staged-Y variant: per group row the reduce segment [t0,t1) and its base inside the group's Y buffer
seg = sp.seg_off.to(torch.int64)
t0, t1 = seg[:-1][rows.long()], seg[1:][rows.long()]
cnt = t1 - t0
csum = torch.cumsum(cnt, 0) - cnt
grp_of_row = torch.repeat_interleave(torch.arange(A, device="cuda"), (off[1:] - off[:-1]).long())
ybase = csum - csum[off[:-1].long()][grp_of_row]
yseg = torch.stack([t0, t1, ybase, torch.zeros_like(t0)], 1).to(torch.int32).contiguous()
y_rows = int((torch.zeros(A, device="cuda", dtype=torch.int64).index_add_(0, grp_of_row, cnt)).max())
# per transposed entry k: triplet t = permT[k], reduce row r(t) at group position rposT[k]
tt = permT.long()
red = inputs["id3_reduce_ca"][tt]
grp = inputs["id_a"][red]
ypos_row = ybase[off[:-1].long()[grp] + rposT.long()]
yloc = ypos_row + (tt - seg[red])
packT = ((yloc << 8) | rposT.long()).to(torch.int32).contiguous()
print(f"staged: y_rows={y_rows} -> LDS {(max_rows * 1792 + y_rows * 28) / 1024:.1f} KB")
for nt in (1024, 512):
    def staged():
        rc = lib.bilt_staged(nt, ptr(Y), ptr(D), ptr(rows), ptr(off), ptr(kseg), ptr(yseg), ptr(packT), ptr(dx), A,
                             max_rows, y_rows, stream())
        assert rc == 0, rc
    t = timeit(staged)
    print(f"staged nt={nt:4d}: {t:7.2f} us  err {float((dx - ref).abs().max()):.1e}")

lib.bilt_scalar.argtypes = [i32, i32] + [vp] * 8 + [i32, i32, vp]
for nt, un in ((1024, 1), (1024, 2), (1024, 4), (512, 2), (512, 4)):
    def scalar():
        rc = lib.bilt_scalar(nt, un, ptr(Y), ptr(D), ptr(rows), ptr(off), ptr(kseg), ptr(permT), ptr(rposT), ptr(dx), A,
                             max_rows, stream())
        assert rc == 0, rc
    t = timeit(scalar)
    print(f"scalar-Y nt={nt:4d} unroll={un}: {t:7.2f} us  err {float((dx - ref).abs().max()):.1e}")


# fused K1+K2 and its adjoint on the same batch (S = 7, C = 64, I = 16)
x = torch.randn(E, 64, device="cuda", generator=g)
Bm = torch.randn(E, 7, 16, device="cuda", generator=g)
dP = torch.randn(E, 16, 64, device="cuda", generator=g)
Sm, P = K.bil_reduce_project(Y, x, Bm, sp)
print(f"bil_reduce_project (7,64,16): {timeit(lambda: K.bil_reduce_project(Y, x, Bm, sp)):7.2f} us")
print(f"bil_project_bwd    (7,64,16): {timeit(lambda: K.bil_project_bwd(dP, Sm, Bm, x, sp)):7.2f} us")
print(f"bil_project_bwd no dY       : {timeit(lambda: K.bil_project_bwd(dP, Sm, Bm, x, sp, want_dY=False)):7.2f} us")

lib.bwd7.argtypes = [i32] + [vp] * 9 + [ctypes.c_int64, vp]
gB_, dSm_, dY_ = torch.empty(E, 7, 16, device="cuda"), torch.empty(E, 7, 64, device="cuda"), torch.empty(T, 7, device="cuda")
for mode, name in ((0, "full"), (1, "no x loads"), (2, "no dY stores"), (3, "no phase-3 MFMA"), (4, "no gB/dSm stores")):
    def run():
        rc = lib.bwd7(mode, ptr(dP), ptr(Sm), ptr(Bm), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(gB_), ptr(dSm_),
                      ptr(dY_), E, stream())
        assert rc == 0, rc
    print(f"bwd7 variant {name:18s}: {timeit(run):7.2f} us")
print(f"edge order, no atom grouping: fwd {timeit(lambda: K.bil_reduce_project(Y, x, Bm, plain)):7.2f} us   "
      f"bwd {timeit(lambda: K.bil_project_bwd(dP, Sm, Bm, x, plain)):7.2f} us")
dS4 = [torch.randn(E, 7, 64, device="cuda", generator=g) for _ in range(4)]
x4 = [torch.randn(E, 64, device="cuda", generator=g) for _ in range(4)]
print(f"bil_dy_multi (7,64) 4 blocks: {timeit(lambda: K.bil_dy_multi(dS4, x4, sp)):7.2f} us")
