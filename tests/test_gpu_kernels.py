"""-m gpu: every C-ABI entry point against its CPU restatement (tests/cpu_kernels.py, float64) on the
same seeded inputs — through the real libgemnet_hip.so on a MI355X.  Shapes include ragged /
non-multiple-of-tile sizes, empty inputs and unaligned leading dimensions."""
import numpy as np
import pytest
import torch

import cpu_kernels as CK
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.graph import RowIndex, SegmentPlan
from oracle import basis_oracle as B

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float64)


def close(out, ref, rtol=2e-5, atol=None):
    ref = ref.to(torch.float64)
    out = out.detach().cpu().to(torch.float64)
    assert out.shape == ref.shape, (out.shape, ref.shape)
    if atol is None:
        atol = 2e-5 * max(1.0, float(ref.abs().max())) if ref.numel() else 0.0
    err = (out - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"max err {float(err.max()):.3e} (atol {atol:.1e}) at {int(bad.sum())} of {ref.numel()}"


def f32(t):
    return t.to(torch.float32).to(DEV)


@pytest.mark.parametrize("M,N,Kd", [(64, 128, 32), (300, 128, 128), (1000, 64, 128), (257, 32, 64),
                                    (130, 16, 6), (77, 1, 128), (513, 128, 384), (95, 100, 42),
                                    (2048, 64, 1024), (5, 7, 3), (1024, 128, 1), (300, 200, 6), (1000, 1, 130)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_plain(M, N, Kd, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + Kd)
    A = rnd(g, *((Kd, M) if ta else (M, Kd)))
    Bm = rnd(g, *((Kd, N) if tb else (N, Kd)))
    ref = CK.gemm(A, Bm, ta, tb)
    out = K.gemm(f32(A), f32(Bm), ta, tb)
    close(out, ref, rtol=1e-5, atol=1e-5 * np.sqrt(Kd) * 4)


def test_gemm_is_transpose_detecting():
    # asymmetric operands: A = identity-like, B = i*100 + j pattern
    M = N = Kd = 64
    A = torch.eye(M, dtype=torch.float64)
    Bm = (torch.arange(N)[:, None] * 100 + torch.arange(Kd)[None, :]).to(torch.float64)
    out = K.gemm(f32(A), f32(Bm), False, False)  # A @ B^T
    close(out, Bm.t(), rtol=0, atol=1e-3)


def test_gemm_fused_epilogue_and_prologue():
    g = torch.Generator().manual_seed(5)
    M, N, Kd, A_rows = 333, 128, 128, 50
    A, W = rnd(g, M, Kd), rnd(g, N, Kd) / np.sqrt(Kd)
    mul, res = rnd(g, M, N), rnd(g, M, N)
    g1, g2 = rnd(g, A_rows, N), rnd(g, A_rows, N)
    i1 = torch.randint(0, A_rows, (M,), generator=g, dtype=torch.int32)
    i2 = torch.randint(0, A_rows, (M,), generator=g, dtype=torch.int32)
    pre = rnd(g, M, Kd)
    kw = dict(act=True, pre_out=True, alpha=0.7, beta=0.5)
    ref_y, ref_z = CK.gemm(A, W, a_dact_pre=pre, mul=mul, res=res, gadd1=g1, gidx1=i1, gadd2=g2, gidx2=i2, **kw)
    y, z = K.gemm(f32(A), f32(W), a_dact_pre=f32(pre), mul=f32(mul), res=f32(res), gadd1=f32(g1),
                  gidx1=i1.to(DEV), gadd2=f32(g2), gidx2=i2.to(DEV), **kw)
    close(z, ref_z, atol=5e-5)
    close(y, ref_y, atol=5e-5)


@pytest.mark.parametrize("Kd,N,tb", [(6, 128, False), (6, 100, True), (42, 112, True), (1, 128, False),
                                     (6, 16, False), (6, 16, True), (42, 32, True), (6, 32, False), (7, 20, True), (2, 1, True)])
def test_gemm_small_k_row_kernel_with_fused_epilogue(Kd, N, tb):
    """K <= 64 and tiny / not a multiple of 4 (edge embedding: 6 radial functions + two gathered atom rows + ScaledSiLU;
    the 42-column circular basis; K = 1 outer products) run the row kernel gemm_smallk with the same epilogue stages; N <= 32
    (the radial projections onto 16 columns) its one-column-per-thread form."""
    g = torch.Generator().manual_seed(Kd * N)
    M, A_rows = 1037, 60
    A = rnd(g, M, Kd)
    W = rnd(g, *((Kd, N) if tb else (N, Kd)))
    mul, res, res2 = rnd(g, M, N), rnd(g, M, N), rnd(g, M, N)
    g1, g2 = rnd(g, A_rows, N), rnd(g, A_rows, N)
    i1 = torch.randint(0, A_rows, (M,), generator=g, dtype=torch.int32)
    i2 = torch.randint(0, A_rows, (M,), generator=g, dtype=torch.int32)
    kw = dict(act=True, pre_out=True, alpha=0.7, beta=0.5, beta2=1.25)
    ref_y, ref_z = CK.gemm(A, W, False, tb, mul=mul, res=res, res2=res2, gadd1=g1, gidx1=i1, gadd2=g2, gidx2=i2, **kw)
    y, z = K.gemm(f32(A), f32(W), False, tb, mul=f32(mul), res=f32(res), res2=f32(res2), gadd1=f32(g1),
                  gidx1=i1.to(DEV), gadd2=f32(g2), gidx2=i2.to(DEV), **kw)
    close(z, ref_z, atol=5e-5)
    close(y, ref_y, atol=5e-5)
    Wb = rnd(g, N, 262)   # a K-column slice of a wider weight (row pitch 262: the edge-embedding matrix)
    if not tb:
        close(K.gemm(f32(A), f32(Wb)[:, 256:256 + Kd]), CK.gemm(A, Wb[:, 256:256 + Kd]), atol=5e-5)


def test_gemm_strided_weight_slices():
    g = torch.Generator().manual_seed(6)
    W = rnd(g, 64, 2 * 32 + 6)  # like edge_emb: (out, 2*atom + rbf); row stride 70 -> scalar loads
    x = rnd(g, 91, 6)
    Wd = f32(W)
    close(K.gemm(f32(x), Wd[:, 64:], False, False), x @ W[:, 64:].t())
    h = rnd(g, 40, 32)
    close(K.gemm(f32(h), Wd[:, 32:64], False, False), h @ W[:, 32:64].t())


@pytest.mark.parametrize("b,m,n,k", [(100, 16, 64, 7), (33, 32, 32, 49), (7, 5, 3, 2), (50, 7, 16, 64), (18122, 16, 64, 7),
                                     (31, 49, 32, 32)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_bmm(b, m, n, k, ta, tb):
    g = torch.Generator().manual_seed(b + m)
    A = rnd(g, *((b, k, m) if ta else (b, m, k)))
    Bm = rnd(g, *((b, n, k) if tb else (b, k, n)))
    close(K.bmm(f32(A), f32(Bm), ta, tb), CK.bmm(A, Bm, ta, tb))


@pytest.mark.parametrize("N,T", [(97, 1000), (5000, 30000)])   # workgroup-per-row (N <= 4096) / wave-per-row kernels
@pytest.mark.parametrize("C", [1, 3, 32, 128, 7 * 6, 320])
def test_gather_segsum(C, N, T):
    g = torch.Generator().manual_seed(C)
    idx = torch.randint(0, N, (T,), generator=g)
    x, y = rnd(g, N, C), rnd(g, T, C)
    ri = RowIndex(idx.to(DEV), N)
    close(K.gather(f32(x), ri.idx32), x[idx])
    perm, seg = ri.csr
    ref = torch.zeros(N, C, dtype=torch.float64).index_add(0, idx, y)
    close(K.segsum(f32(y), perm, seg, N), ref)
    sidx = torch.sort(idx).values
    rs = RowIndex(sidx.to(DEV), N, is_sorted=True)
    ref = torch.zeros(N, C, dtype=torch.float64).index_add(0, sidx, y)
    close(K.segsum(f32(y), None, rs.csr[1], N), ref)


def test_empty_inputs_are_noops():
    e = torch.zeros(0, 8, device=DEV)
    ri = RowIndex(torch.zeros(0, dtype=torch.long, device=DEV), 5)
    assert K.gather(torch.zeros(5, 8, device=DEV), ri.idx32).shape == (0, 8)
    out = K.segsum(e, *ri.csr, 5)
    assert out.shape == (5, 8) and float(out.abs().sum()) == 0.0


def _segplan(g, E, J, mean_k):
    counts = torch.randint(0, 2 * mean_k, (E,), generator=g)
    red = torch.repeat_interleave(torch.arange(E), counts)
    exp = torch.randint(0, J, (red.shape[0],), generator=g)
    cpu = SegmentPlan(red, exp, E, J)
    dev = SegmentPlan(red.to(DEV), exp.to(DEV), E, J)
    return cpu, dev


@pytest.mark.parametrize("S,C,E,J,mk", [(7, 64, 500, 500, 18), (7, 32, 77, 77, 5), (49, 32, 60, 300, 80)])
def test_bilinear_kernels(S, C, E, J, mk):
    g = torch.Generator().manual_seed(S * C)
    cpu, dev = _segplan(g, E, J, mk)
    T = cpu.size
    Y, x, D = rnd(g, T, S), rnd(g, J, C), rnd(g, E, S, C)
    close(K.bil_reduce(f32(Y), f32(x), dev), CK.bil_reduce(Y, x, cpu), atol=1e-4)
    close(K.bil_reduce_t(f32(Y), f32(D), dev), CK.bil_reduce_t(Y, D, cpu), atol=1e-4)
    close(K.bil_dot(f32(D), f32(x), dev), CK.bil_dot(D, x, cpu), atol=1e-4)


@pytest.mark.parametrize("A,deg,C", [(40, 9, 64), (7, 30, 64), (25, 4, 32), (3, 80, 64), (2, 100, 64)])
def test_bilinear_adjoint_grouped_by_target_atom(A, deg, C):
    """Triplets c->a<-b: reduce and expand edge share the target atom, so gn_bil_reduce_t_grouped_f32 parks the
    atom's dSm blocks in LDS; same result as the ungrouped adjoint (and the >160 KB case falls back to it)."""
    g = torch.Generator().manual_seed(A * deg)
    S = 7
    # ragged in-degrees, edges of one atom NOT contiguous in edge order
    degs = torch.randint(0, deg + 1, (A,), generator=g)
    degs[0] = deg
    tgt = torch.repeat_interleave(torch.arange(A), degs)
    tgt = tgt[torch.randperm(tgt.shape[0], generator=g)]
    E = int(tgt.shape[0])
    red, exp = [], []
    for a in range(A):
        es = torch.nonzero(tgt == a).flatten().tolist()
        for r in es:
            for x in es:
                if r != x and (A > 5 or float(torch.rand((), generator=g)) < 0.85):   # small cases: incomplete pair sets
                    red.append(r), exp.append(x)
    red, exp = torch.tensor(red, dtype=torch.int64), torch.tensor(exp, dtype=torch.int64)
    order = torch.argsort(red, stable=True)
    red, exp = red[order], exp[order]
    cpu = SegmentPlan(red, exp, E, E)
    dev = SegmentPlan(red.to(DEV), exp.to(DEV), E, E)
    dev.set_row_groups(tgt.to(DEV), A)
    assert dev.groups[4] == deg
    Y, D = rnd(g, cpu.size, S), rnd(g, E, S, C)
    got = K.bil_reduce_t(f32(Y), f32(D), dev)
    close(got, CK.bil_reduce_t(Y, D, cpu), atol=1e-4)
    plain = SegmentPlan(red.to(DEV), exp.to(DEV), E, E)
    close(got, K.bil_reduce_t(f32(Y), f32(D), plain).double().cpu(), atol=1e-5)


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_ssilu(k):
    x = torch.linspace(-12, 12, 1001, dtype=torch.float64)
    close(K.ssilu(f32(x), k), CK.ssilu(x, k), atol=2e-6)


@pytest.mark.parametrize("kd,kf", [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1)])
def test_bessel_rbf(kd, kf, golden_basis):
    d = torch.tensor(golden_basis["d"])
    f = torch.tensor(golden_basis["freq"])
    close(K.bessel_rbf(f32(d), f32(f), 5.0, 5, kd, kf), CK.bessel_rbf(d, f, 5.0, 5, kd, kf), atol=3e-6 * (1 + 9 * kd))
    if (kd, kf) == (0, 0):
        close(K.bessel_rbf(f32(d), f32(f), 5.0, 5, 0, 0), torch.tensor(golden_basis["bessel_rbf"]), atol=3e-6)


@pytest.mark.parametrize("cutoff,dkey,gkey", [(5.0, "d", "c5"), (10.0, "d10", "c10")])
@pytest.mark.parametrize("kd", [0, 1, 2])
def test_sph_radial(cutoff, dkey, gkey, kd, golden_basis):
    d = torch.tensor(golden_basis[dkey])
    z = torch.tensor(B.jn_zeros(7, 6))
    nrm = torch.tensor(B.sph_bessel_normalizer(7, 6))
    out = K.sph_radial(f32(d), z.to(DEV), nrm.to(DEV), cutoff, 5, kd)
    # inputs are rounded to f32 before evaluation: compare against the restatement on the same f32 d
    d32 = d.to(torch.float32).to(torch.float64)
    close(out, CK.sph_radial(d32, z, nrm, cutoff, 5, kd), rtol=1e-5, atol=2e-6 * (1 + 20 * kd))
    if kd == 0:
        close(out, torch.tensor(golden_basis[f"radial_{gkey}"]), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ylm0(k, golden_basis):
    th = torch.tensor(golden_basis["theta"])
    th32 = th.to(torch.float32).to(torch.float64)
    close(K.ylm0(f32(th), 7, k), CK.ylm0(th32, 7, k), atol=1e-5)
    if k == 0:
        close(K.ylm0(f32(th), 7, 0), torch.tensor(golden_basis["y_l0"]), atol=1e-5)


@pytest.mark.parametrize("kt,kp", [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)])
def test_ylm(kt, kp, golden_basis):
    th, ph = torch.tensor(golden_basis["theta"]), torch.tensor(golden_basis["phi"])
    th32, ph32 = th.to(torch.float32).to(torch.float64), ph.to(torch.float32).to(torch.float64)
    ref = CK.ylm(th32, ph32, 7, kt, kp)
    close(K.ylm(f32(th), f32(ph), 7, kt, kp), ref, atol=2e-6 * max(1.0, float(ref.abs().max())))
    if (kt, kp) == (0, 0):
        close(K.ylm(f32(th), f32(ph), 7, 0, 0), torch.tensor(golden_basis["y_lm"]), atol=1e-5)


def test_gemm_row_gathered_and_second_residual():
    g = torch.Generator().manual_seed(9)
    M, N, Kd = 515, 128, 64
    A, W = rnd(g, M, Kd), rnd(g, N, Kd) / np.sqrt(Kd)
    res, res2 = rnd(g, M, N), rnd(g, M, N)
    ridx = torch.randperm(M, generator=g).to(torch.int32)
    kw = dict(act=True, alpha=1.3, beta=0.7, beta2=0.6)
    ref = CK.gemm(A, W, res=res, ridx=ridx, res2=res2, **kw)
    out = K.gemm(f32(A), f32(W), res=f32(res), ridx=ridx.to(DEV), res2=f32(res2), **kw)
    close(out, ref, atol=5e-5)


@pytest.mark.parametrize("M,N,Kd", [(1024, 128, 128), (1000, 64, 128), (300, 32, 64), (18000, 128, 128),
                                    (18001, 64, 1024), (5000, 32, 128)])
def test_gemm_pipelined_nt_all_tile_configs(M, N, Kd):
    g = torch.Generator().manual_seed(M + N)
    A, W, pre = rnd(g, M, Kd), rnd(g, N, Kd) / np.sqrt(Kd), rnd(g, M, Kd)
    close(K.gemm(f32(A), f32(W)), CK.gemm(A, W), atol=1e-4)
    close(K.gemm(f32(A), f32(W), a_dact_pre=f32(pre), alpha=0.5), CK.gemm(A, W, a_dact_pre=pre, alpha=0.5), atol=1e-4)


@pytest.mark.parametrize("act,has_mul", [(True, True), (True, False), (False, True)])
def test_dact_mul(act, has_mul):
    g = torch.Generator().manual_seed(3)
    gr, z, mul = rnd(g, 777, 64), rnd(g, 777, 64), rnd(g, 777, 64)
    ref_dz, ref_gm = CK.dact_mul(gr, z, act, mul if has_mul else None, 0.37, want_gmul=True)
    dz, gm = K.dact_mul(f32(gr), f32(z), act, f32(mul) if has_mul else None, 0.37, want_gmul=True)
    close(dz, ref_dz, atol=1e-5)
    close(gm, ref_gm, atol=1e-5)
    # float4 body + scalar tail (n % 4 = 3), and the all-scalar path of a misaligned operand
    gr, z, mul = rnd(g, 333, 7), rnd(g, 333, 7), rnd(g, 333, 7)
    ref_dz, _ = CK.dact_mul(gr, z, act, mul if has_mul else None, 1.5, want_gmul=False)
    dz, none = K.dact_mul(f32(gr), f32(z), act, f32(mul) if has_mul else None, 1.5, want_gmul=False)
    assert none is None
    close(dz, ref_dz, atol=1e-5)
    gz = f32(torch.cat([z.flatten()[:1], z.flatten()]))[1:].reshape(333, 7)   # same values, 4-byte aligned only
    dz, _ = K.dact_mul(f32(gr), gz, act, f32(mul) if has_mul else None, 1.5, want_gmul=False)
    close(dz, ref_dz, atol=1e-5)


def _rand_geometry(g, n_atoms=40, n_edges=300, n_trip=900):
    R = (torch.rand(n_atoms, 3, generator=g, dtype=torch.float64) * 6.0)
    ic = torch.randint(0, n_atoms, (n_edges,), generator=g)
    ia = (ic + 1 + torch.randint(0, n_atoms - 1, (n_edges,), generator=g)) % n_atoms  # != ic
    tc = torch.randint(0, n_atoms, (n_trip,), generator=g)
    ta = (tc + 1 + torch.randint(0, n_atoms - 1, (n_trip,), generator=g)) % n_atoms
    tb = torch.randint(0, n_atoms, (n_trip,), generator=g)
    ok = (tb != ta) & (tb != tc)
    return R, ic.int(), ia.int(), tc[ok].int(), ta[ok].int(), tb[ok].int()


def test_edge_basis_fused_fwd_bwd():
    g = torch.Generator().manual_seed(21)
    R, ic, ia, *_ = _rand_geometry(g)
    R32 = R.float().double()  # the kernel sees f32 positions
    z, nrm = torch.tensor(B.jn_zeros(7, 6)), torch.tensor(B.sph_bessel_normalizer(7, 6))
    freq = torch.arange(1, 7, dtype=torch.float64) * np.pi
    D, V, rbf, rad = K.edge_basis_fwd(f32(R), ic.to(DEV), ia.to(DEV), f32(freq), z.to(DEV), nrm.to(DEV), 8.0, 5, True, True)
    rD, rV, rrbf, rrad = CK.edge_basis_fwd(R32, ic, ia, freq, z, nrm, 8.0, 5, True, True)
    close(D, rD, atol=2e-6); close(V, rV, atol=2e-6); close(rbf, rrbf, atol=1e-5); close(rad, rrad, rtol=1e-4, atol=2e-5)
    gD, grbf, grad = rnd(g, ic.shape[0]), rnd(g, ic.shape[0], 6), rnd(g, ic.shape[0], 7, 6)
    W = K.edge_basis_bwd(f32(gD), f32(grbf), f32(grad), f32(R), ic.to(DEV), ia.to(DEV), f32(freq), z.to(DEV), nrm.to(DEV), 8.0, 5)
    rW = CK.edge_basis_bwd(gD, grbf, grad, R32, ic, ia, freq, z, nrm, 8.0, 5)
    close(W, rW, rtol=2e-4, atol=2e-4 * float(rW.abs().max()))


def test_trip_basis_fused_fwd_bwd_including_collinear():
    g = torch.Generator().manual_seed(22)
    R, _, _, tc, ta, tb = _rand_geometry(g)
    # make atoms 0,1,2 exactly collinear and add that triplet: exercises the 1e-9 clamp
    R[0] = torch.tensor([0.0, 0.0, 0.0]); R[1] = torch.tensor([1.0, 0.0, 0.0]); R[2] = torch.tensor([2.5, 0.0, 0.0])
    tc = torch.cat([tc, torch.tensor([1], dtype=torch.int32)]); ta = torch.cat([ta, torch.tensor([0], dtype=torch.int32)])
    tb = torch.cat([tb, torch.tensor([2], dtype=torch.int32)])
    R32 = R.float().double()
    Y, th = K.trip_basis_fwd(f32(R), tc.to(DEV), ta.to(DEV), tb.to(DEV), 7, want_theta=True)
    rY, rth = CK.trip_basis_fwd(R32, tc, ta, tb, 7, want_theta=True)
    close(th, rth, atol=5e-6); close(Y, rY, atol=2e-5)
    gY = rnd(g, tc.shape[0], 7)
    Gc, Gb = K.trip_basis_bwd(f32(gY), f32(R), tc.to(DEV), ta.to(DEV), tb.to(DEV))
    rGc, rGb = CK.trip_basis_bwd(gY, R32, tc, ta, tb)
    # near-collinear triplets amplify f32 rounding of the angle: compare with a scale-aware tolerance
    close(Gc[:-1], rGc[:-1], rtol=2e-3, atol=2e-3 * float(rGc.abs().median()))
    close(Gb[:-1], rGb[:-1], rtol=2e-3, atol=2e-3 * float(rGb.abs().median()))
    assert torch.isfinite(Gc).all() and torch.isfinite(Gb).all()


def _stack_programs(M, width, g, dev, park=True):
    """A residual-stack-like program with every op kind, slots aliasing and slot/global operands.
    park=False: the same without the register parking slot (such programs take the wide kernel layout, csrc/chain3.hip)."""
    def mk(*shape):
        return rnd(g, *shape)
    x, W0, W1, W2 = mk(M, 64), mk(width, 64) / 8, mk(width, width) / 11, mk(width, width) / 11
    res, skip, ga = mk(M, width), mk(M, width), mk(50, width)
    gi = torch.randint(0, 50, (M,), generator=g, dtype=torch.int32)
    rows = torch.randperm(M, generator=g).to(torch.int32)
    Z = mk(M, width)

    def build(conv, idx):
        t = {k: conv(v) for k, v in dict(x=x, W0=W0, W1=W1, W2=W2, res=res, skip=skip, ga=ga, Z=Z).items()}
        outs = {k: conv(torch.zeros(M, width, dtype=torch.float64)) for k in ("z0", "z1", "z2", "y", "sc", "st")}
        p = K.ChainProgram(M)
        p.load(0, t["x"], rows=idx(rows))
        p.gemm(t["W0"], a_slot=0, y_slot=1, act=True, gadd1=t["ga"], gidx1=idx(gi), pre_out=outs["z0"],
               res=t["res"], beta=0.7)
        p.gemm(t["W1"], a_slot=1, y_slot=0, act=True, pre_out=outs["z1"])
        p.gemm(t["W2"], a_slot=0, y_slot=1, act=True, pre_out=outs["z2"], res=1, beta=0.7, res2=t["skip"], beta2=0.6,
               out=outs["y"], mul=None)
        if park:
            p.scale(2, 1, 0.25, width=width)                      # park a scaled copy in the third slot ...
        p.scale(0, 1, 0.3, Z=t["Z"], out=outs["sc"])
        p.gemm(t["W1"], a_slot=0, y_slot=0, act=False, mul=1, alpha=1.5, res=2 if park else t["skip"], beta=1.0)   # ... and add it back here
        p.store(0, outs["st"])
        return p, outs
    return build


def _adjoint_programs(M, width, g, dev, park=True):
    """The fused-epilogue features of the adjoint programs: LOAD with scale + second tensor, a global mul with
    activation-derivative mode, second outputs (from the final value and from the pre-mul value, to a slot aliasing
    the A operand and to global memory).  park=False: without the register parking slot (wide kernel layout)."""
    def mk(*shape):
        return rnd(g, *shape)
    gin, W1, W2 = mk(M, width), mk(width, width) / 11, mk(width, width) / 11
    z1, z2, z3, r = mk(M, width), mk(M, width), mk(M, width), mk(M, width)

    def build(conv, idx):
        t = {k: conv(v) for k, v in dict(g=gin, W1=W1, W2=W2, z1=z1, z2=z2, z3=z3, r=r).items()}
        outs = {k: conv(torch.zeros(M, width, dtype=torch.float64)) for k in ("a", "b", "o2", "c", "d")}
        p = K.ChainProgram(M)
        p.load(0, t["g"], alpha=0.7, y2=1, alpha2=1.3, Z2=t["z2"], mode2=0)         # G, dz2 = G * 1.3 * f'(z2)
        if park:
            p.scale(2, 0, 0.5, width=width)
        p.gemm(t["W2"], a_slot=1, y_slot=1, mul=t["z1"], mul_mode=2, out=None)        # dz1 = (dz2 W2) f'(z1)
        p.gemm(t["W1"], a_slot=1, y_slot=0, res=0, alpha=2.0, beta=0.5, y2=1, alpha2=0.9, Z2=t["z3"], mode2=2,
               out2=outs["o2"])                                                       # G' and G' * 0.9 * f(z3)
        p.store(0, outs["a"])
        p.store(1, outs["b"])
        p.gemm(t["W2"], a_slot=1, y_slot=1, mul=t["r"], mul_mode=1, alpha=1.1, res=2 if park else None, beta=1.0,
               y2=0, y2_src=1, alpha2=0.8, Z2=t["z1"], mode2=1)                      # two products of one GEMM
        p.store(0, outs["c"])
        p.store(1, outs["d"])
        return p, outs
    return build


def _source_programs(M, width, g, dev):
    """The second-order source terms (gn_chain_op.src_*) at every place they may sit: on a SCALE, after the mul stage of
    a GEMM (ssilu'' of the global mul operand), on the second output of a GEMM and of a LOAD, with and without Q."""
    def mk(*shape):
        return rnd(g, *shape)
    gin, W1, W2 = mk(M, width), mk(width, width) / 11, mk(width, width) / 11
    z1, z2, z3 = mk(M, width), mk(M, width), mk(M, width)
    P1, Q1, P2, Q2, P3 = mk(M, width), mk(M, width), mk(M, width), mk(M, width), mk(M, width)

    def build(conv, idx):
        t = {k: conv(v) for k, v in dict(g=gin, W1=W1, W2=W2, z1=z1, z2=z2, z3=z3, P1=P1, Q1=Q1, P2=P2, Q2=Q2,
                                         P3=P3).items()}
        outs = {k: conv(torch.zeros(M, width, dtype=torch.float64)) for k in ("a", "b", "c", "d", "e", "f")}
        S = K.ChainProgram.source
        p = K.ChainProgram(M)
        # LOAD: slot 0 <- g, slot 1 <- g f'(z2) + P1 f''(z2) Q1
        p.load(0, t["g"], y2=1, alpha2=1.0, Z2=t["z2"], mode2=0, add2=S(t["P1"], t["Q1"], d2=True))
        p.store(1, outs["a"])
        # GEMM stage-1 source: y = (x W2^T) f'(z1) * 0.9 + 1.3 P2 f''(z1) Q2, stored
        p.gemm(t["W2"], a_slot=1, y_slot=1, mul=t["z1"], mul_mode=2, alpha=0.9, add=S(t["P2"], t["Q2"], d2=True, alpha=1.3),
               out=outs["b"])
        # GEMM second output with a plain (no ssilu'') source and no Q: y2 = y * 0.7 * z3 + 0.5 P3
        p.gemm(t["W1"], a_slot=1, y_slot=0, res=0, beta=0.5, y2=1, alpha2=0.7, Z2=t["z3"], mode2=1,
               add2=S(t["P3"], None, alpha=0.5), out2=outs["c"])
        p.store(0, outs["d"])
        # SCALE with a source: slot 0 <- slot 1 * 1.1 * f(z2) + P1 Q2 * 2
        p.scale(0, 1, 1.1, Z=t["z2"], mode=2, add=S(t["P1"], t["Q2"], alpha=2.0), out=outs["e"])
        # SCALE with the ssilu'' source of its own Z
        p.scale(1, 0, 1.0, Z=t["z1"], mode=0, add=S(t["P2"], t["Q1"], d2=True), out=outs["f"])
        return p, outs
    return build


# relative error budget of one chain program per arithmetic: f32 MFMA and the six-product split are fp32-equivalent
CHAIN_TOL = {"f32": 1e-4, "split6": 1e-4, "h3": 1e-4, "split3": 2e-3, "bf16": 1e-1}


@pytest.mark.parametrize("mode", ["f32", "split6", "h3", "h3/row", "split3", "bf16"])
@pytest.mark.parametrize("M,width", [(1000, 128), (33, 64), (18122, 128), (5000, 128), (9000, 64), (13000, 128), (120000, 128)])
def test_chain_kernel_vs_interpreter(M, width, mode, monkeypatch):
    """Every op kind, aliasing slots, slot / register / global operands, all five tile heights (RT = 1..5), on the f32
    MFMA kernel, on the split-operand bf16 kernel (packed weights) and on both layouts of the two-plane fp16 arithmetic
    ("h3" = csrc/chain2.hip, the default; "h3/row" = the row-resident csrc/chain4.hip, 1 .. 7 compute waves per workgroup,
    which also takes the programs chain2's row-scaled form refuses), against the float64 interpreter."""
    if "/" in mode:
        mode, layout = mode.split("/")
        monkeypatch.setattr(K, "CHAIN_LAYOUT", layout)
    if M > 20000 and K.CHAIN_LAYOUT != "row":
        pytest.skip("the 7-wave workgroups of the row-resident layout only")
    tol = CHAIN_TOL[mode]
    worst = 0.0
    import functools
    nopark = (functools.partial(_stack_programs, park=False), functools.partial(_adjoint_programs, park=False))
    for maker in (_stack_programs, _adjoint_programs) + nopark + ((_source_programs,) if mode != "f32" else ()):
        g = torch.Generator().manual_seed(M)
        build = maker(M, width, g, DEV)
        p_ref, o_ref = build(lambda t: t.clone(), lambda i: i)
        p_dev, o_dev = build(lambda t: f32(t), lambda i: i.to(DEV))
        if mode == "h3" and K.CHAIN_LAYOUT != "row" and K.h3_hazards(p_dev):
            # second-order source terms / gathered adds into LDS-resident rows: refused, never computed out of range
            with pytest.raises(RuntimeError):
                K.chain(p_dev, mode=mode)
            continue
        CK.chain(p_ref, mode=mode)
        K.chain(p_dev, mode=mode)
        for k in o_ref:
            scale = max(1.0, float(o_ref[k].abs().max()))
            err = float((o_dev[k].double().cpu() - o_ref[k]).abs().max()) / scale
            worst = max(worst, err)
            assert err <= 2 * tol, (getattr(maker, '__name__', 'no-park variant'), k, err)
    print(f"chain {mode} M={M} width={width}: max err / scale = {worst:.2e}")


@pytest.mark.parametrize("tile_rows", [0, 8, 24, 40, 48])
@pytest.mark.parametrize("M,width", [(1000, 128), (33, 64), (18122, 128), (5000, 128), (9000, 64)])
def test_wide_chain_layout_equals_tall_layout_bitwise(M, width, tile_rows):
    """csrc/chain3.hip (4 waves x 32 columns, row tiles of any multiple of 8 rows up to 48, two workgroups per CU) against
    csrc/chain2.hip (8 waves x 16 columns): the same products accumulated in the same order — every output bit for bit, for
    forward stacks (activations, gathered adds, pre-activation / stored-derivative outputs) and first-order adjoint programs
    (row scales, second outputs), ragged last tiles and half-used MFMA row blocks included."""
    checked = 0
    default_layout, default_rows = K.CHAIN_LAYOUT, K.WIDE_TILE_ROWS
    try:
        import functools
        for maker in (functools.partial(_stack_programs, park=False), functools.partial(_adjoint_programs, park=False),
                      _adjoint_programs):
            outs = {}
            for layout in ("tall", "wide"):
                g = torch.Generator().manual_seed(M)
                build = maker(M, width, g, DEV)
                p_dev, o_dev = build(lambda t: f32(t), lambda i: i.to(DEV))
                if K.h3_hazards(p_dev):
                    break
                K.CHAIN_LAYOUT = layout
                K.WIDE_TILE_ROWS = tile_rows if layout == "wide" else 0     # GN_CHAIN_WIDE_ROWS bits of this launch's `nprod`
                K.chain(p_dev, mode="h3")
                torch.cuda.synchronize()
                outs[layout] = o_dev
            if len(outs) == 2:
                for k in outs["tall"]:
                    assert torch.equal(outs["tall"][k], outs["wide"][k]), (k, tile_rows)
                    checked += 1
    finally:
        K.CHAIN_LAYOUT, K.WIDE_TILE_ROWS = default_layout, default_rows
    assert checked >= 10


def test_source_terms_are_rejected_by_the_f32_chain_kernel():
    """The f32-MFMA chain kernel has no second-order source terms: it must refuse such a program, never drop the term."""
    g = torch.Generator().manual_seed(3)
    build = _source_programs(64, 128, g, DEV)
    p_dev, _ = build(lambda t: f32(t), lambda i: i.to(DEV))
    with pytest.raises(RuntimeError):
        K.chain(p_dev, mode="f32")


@pytest.mark.parametrize("fmt", [0, 1], ids=["bf16x3", "f16x2"])
def test_grouped_weight_pack_equals_single_packs(fmt):
    """gn_pack_weight_split_grouped (all weights of a training step in one launch) == gn_pack_weight_split per weight,
    bit for bit, for plain, transposed, sliced (row pitch > K) and ragged (N % 16, K % 32 != 0) matrices; both plane
    formats, and a table that mixes them (the S1 / S2 and S3 / S4 entries of one weight in "h3" mode)."""
    g = torch.Generator().manual_seed(9)
    big = f32(rnd(g, 128, 390))
    mats = [(f32(rnd(g, 128, 128)), False), (f32(rnd(g, 128, 128)), True), (f32(rnd(g, 64, 16)), False),
            (f32(rnd(g, 16, 64)), True), (big[:, 128:256], False), (big[:, 256:384], True), (f32(rnd(g, 128, 40)), False),
            (f32(rnd(g, 48, 128)), False)]
    fmts = [fmt if i % 3 else 1 - fmt for i in range(len(mats))]
    singles = [K.pack_weight_split(W, trans=t, fmt=f) for (W, t), f in zip(mats, fmts)]
    outs = [torch.zeros_like(sp) for sp in singles]
    for o, f in zip(outs, fmts):
        o._gn_fmt = f
    assert singles[1].numel() * (3 if fmts[1] else 2) == singles[0].numel() * (3 if fmts[0] else 2)   # 2 vs 3 planes
    table, units = K.pack_job_table([(W, t, o) for (W, t), o in zip(mats, outs)])
    K.pack_weight_split_grouped(table, len(mats), units)
    torch.cuda.synchronize()
    for sp, o in zip(singles, outs):
        assert torch.equal(sp, o)


def test_split_six_products_is_fp32_equivalent():
    """One 128x128 layer: the six-product split reproduces the f32-MFMA result to fp32 rounding level, and the plane
    decomposition is exact (LOAD -> STORE round trip is bit-identical)."""
    g = torch.Generator().manual_seed(5)
    M = 4096
    x, W = f32(rnd(g, M, 128)), f32(rnd(g, 128, 128) / 11)
    ref = x.double() @ W.double().t()
    outs = {}
    for mode in ("f32", "split6"):
        y = torch.empty(M, 128, device=DEV)
        p = K.ChainProgram(M)
        p.load(0, x)
        p.gemm(W, a_slot=0, y_slot=1, out=y)
        K.chain(p, mode=mode)
        outs[mode] = float((y.double() - ref).abs().max())
    assert outs["split6"] <= 2.0 * outs["f32"] + 1e-7, outs
    rt = torch.empty(M, 128, device=DEV)
    p = K.ChainProgram(M)
    p.load(0, x)
    p.store(0, rt)
    K.chain(p, mode="split6")
    assert torch.equal(rt, x)


def test_two_plane_fp16_form_accuracy_and_range():
    """Mode "h3" (two fp16 planes, three products).  A LINEAR program (adjoint sweeps) carries a power-of-two scale per
    row: rows from 1e-8 to 1e4 all come out with the same relative accuracy (2^-22 operand rounding: within 4x of the
    f32-MFMA result), the LOAD -> STORE round trip keeps 21 bits of every row's maximum.  A program with an activation
    is unscaled (activations are O(1) by the model's normalisation): fp32-like accuracy at O(1), and a value beyond the
    fp16 range comes out non-finite — never silently wrong."""
    g = torch.Generator().manual_seed(6)
    M = 4096
    x, W = f32(rnd(g, M, 128)), f32(rnd(g, 128, 128) / 11)
    scale = 10.0 ** (torch.rand(M, generator=g) * 12 - 8)
    x = x * scale[:, None].to(DEV)
    ref = x.double() @ W.double().t()
    rel = {}
    for mode in ("f32", "h3"):
        y = torch.empty(M, 128, device=DEV)
        p = K.ChainProgram(M)
        p.load(0, x)
        p.gemm(W, a_slot=0, y_slot=1, out=y)
        K.chain(p, mode=mode)
        rel[mode] = ((y.double() - ref).norm(dim=1) / ref.norm(dim=1)).cpu()
    print(f"row-wise relative error, rows of scale 1e-8 .. 1e4: f32 MFMA max {float(rel['f32'].max()):.2e}, "
          f"h3 max {float(rel['h3'].max()):.2e} median {float(rel['h3'].median()):.2e}")
    assert float(rel["h3"].max()) <= 4.0 * float(rel["f32"].max())
    rt = torch.empty(M, 128, device=DEV)
    p = K.ChainProgram(M)
    p.load(0, x)
    p.store(0, rt)
    K.chain(p, mode="h3")
    assert float(((rt - x).abs().amax(dim=1) / x.abs().amax(dim=1)).max()) <= 2.0 ** -21
    # two chained layers with the running row scale inherited through the first output and a slot residual
    W2 = f32(rnd(g, 128, 128) / 11)
    ref2 = (ref @ W2.double().t() + x.double()) * 0.5
    y = torch.empty(M, 128, device=DEV)
    p = K.ChainProgram(M)
    p.load(0, x)
    p.gemm(W, a_slot=0, y_slot=1)
    p.gemm(W2, a_slot=1, y_slot=1, res=0, beta=0.5, out=y)
    K.chain(p, mode="h3")
    assert float(((y.double() - ref2).norm(dim=1) / ref2.norm(dim=1)).max()) <= 1e-6
    # non-linear program: no row scale
    xs = f32(rnd(g, M, 128))
    refa = torch.nn.functional.silu(xs.double() @ W.double().t()) / 0.6
    big = xs.clone()
    big[5, 7] = 7.0e4
    for inp, finite in ((xs, True), (big, False)):
        y = torch.empty(M, 128, device=DEV)
        p = K.ChainProgram(M)
        p.load(0, inp)
        p.gemm(W, a_slot=0, y_slot=1, act=True, out=y)
        K.chain(p, mode="h3")
        if finite:
            close(y, refa.cpu(), rtol=1e-5, atol=1e-5)
        else:
            assert not bool(torch.isfinite(y[5]).all()) and bool(torch.isfinite(y[6]).all())


@pytest.mark.parametrize("mode", ["split6", "h3", "bf16"])
@pytest.mark.parametrize("M", [77, 18122])
def test_chain_gemm_stores_the_activation_derivative(mode, M):
    """act bit 1 of a chain GEMM: `pre_out` receives ssilu'(z) (what a first-order adjoint multiplies by) instead of z,
    from the same sigmoid as the activation; the f32-MFMA kernel refuses the flag."""
    g = torch.Generator().manual_seed(M)
    x, W = f32(rnd(g, M, 128)), f32(rnd(g, 64, 128) / 6)
    z = x.double().cpu() @ W.double().cpu().t()
    sg = torch.sigmoid(z)
    y, d = torch.empty(M, 64, device=DEV), torch.empty(M, 64, device=DEV)
    p = K.ChainProgram(M)
    p.load(0, x)
    p.gemm(W, a_slot=0, y_slot=1, act=True, pre_out=d, pre_deriv=True, out=y)
    K.chain(p, mode=mode)
    tol = dict(rtol=2e-5, atol=2e-5) if mode != "bf16" else dict(rtol=5e-2, atol=5e-2)
    close(y, z * sg / 0.6, **tol)
    close(d, sg * (1 + z * (1 - sg)) / 0.6, **tol)
    with pytest.raises(RuntimeError):
        K.chain(p, mode="f32")


def test_packed_weight_format_must_match_the_mode():
    g = torch.Generator().manual_seed(6)
    x, W = f32(rnd(g, 64, 128)), f32(rnd(g, 128, 128))
    p = K.ChainProgram(64)
    p.load(0, x)
    p.gemm(W, a_slot=0, y_slot=1, out=torch.empty(64, 128, device=DEV), packed=K.pack_weight_split(W, fmt=0))
    with pytest.raises(RuntimeError):
        K.chain(p, mode="h3")


@pytest.mark.parametrize("mode", ["f32", "split6", "h3"])
@pytest.mark.parametrize("M", [300, 18122])
def test_chain_gemm_accumulates_into_its_residual(mode, M):
    """A chain GEMM whose global residual IS its output (res / res2 = out): the running-gradient form used by
    ops.accumulate_gradient; every element is read and rewritten by one thread — same bits as the two-tensor form."""
    g = torch.Generator().manual_seed(M)
    x, W, W16 = f32(rnd(g, M, 128)), f32(rnd(g, 128, 128) / 11), f32(rnd(g, 16, 128) / 11)
    base, base16 = f32(rnd(g, M, 128)), f32(rnd(g, M, 16))
    outs = []
    for inplace in (False, True):
        run, run16 = base.clone(), base16.clone()
        y = run if inplace else torch.empty_like(run)
        y16 = run16 if inplace else torch.empty_like(run16)
        p = K.ChainProgram(M)
        p.load(0, x)
        p.gemm(W16, a_slot=0, y_slot=-1, out=y16, res=run16, beta=1.0)
        # (the row-scaled fp16 form takes a global residual only on a result that leaves LDS: kernels.h3_hazards)
        p.gemm(W, a_slot=0, y_slot=-1 if mode == "h3" else 1, out=y, res2=run, beta2=1.0)
        K.chain(p, mode=mode)
        outs.append((y.clone(), y16.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(outs[1][0], base.double().cpu() + x.double().cpu() @ W.double().cpu().t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,Kd,ta,tb", [(128, 128, 18122, True, True), (64, 128, 5000, True, True),
                                          (16, 6, 20000, True, True), (128, 1024, 18122, True, True)])
def test_gemm_splitk_weight_gradient_shapes(M, N, Kd, ta, tb):
    g = torch.Generator().manual_seed(M + N)
    A = rnd(g, *((Kd, M) if ta else (M, Kd))) / 10
    Bm = rnd(g, *((Kd, N) if tb else (N, Kd))) / 10
    ref = CK.gemm(A, Bm, ta, tb, alpha=0.5)
    out = K.gemm(f32(A), f32(Bm), ta, tb, alpha=0.5)
    close(out, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    assert torch.equal(out, K.gemm(f32(A), f32(Bm), ta, tb, alpha=0.5))  # deterministic reduction order


@pytest.mark.parametrize("S,C,I,E,J,mk", [(7, 64, 16, 500, 500, 18), (7, 32, 16, 77, 77, 5), (49, 32, 32, 60, 300, 80)])
def test_bilinear_fused_project_kernels(S, C, I, E, J, mk):
    g = torch.Generator().manual_seed(S * C + 1)
    cpu, dev = _segplan(g, E, J, mk)
    T = cpu.size
    Y, x, Bm, dP = rnd(g, T, S), rnd(g, J, C), rnd(g, E, S, I), rnd(g, E, I, C)
    Sm, P = K.bil_reduce_project(f32(Y), f32(x), f32(Bm), dev)
    rSm, rP = CK.bil_reduce_project(Y, x, Bm, cpu)
    close(Sm, rSm, atol=1e-4); close(P, rP, atol=2e-4 * float(rP.abs().max()))
    gB, dSm, dY = K.bil_project_bwd(f32(dP), f32(rSm), f32(Bm), f32(x), dev)
    rgB, rdSm, rdY = CK.bil_project_bwd(dP, rSm, Bm, x, cpu)
    close(gB, rgB, atol=2e-4 * float(rgB.abs().max())); close(dSm, rdSm, atol=1e-4)
    close(dY, rdY, atol=2e-4 * float(rdY.abs().max()))
    gB2, dSm2, none = K.bil_project_bwd(f32(dP), f32(rSm), f32(Bm), f32(x), dev, want_dY=False)
    assert none is None and torch.equal(gB2, gB) and torch.equal(dSm2, dSm)
    base = f32(rnd(g, E, S, I))        # running gradient of the shared radial basis: gB joins it in the same launch
    run = base.clone()
    gB3, _, _ = K.bil_project_bwd(f32(dP), f32(rSm), f32(Bm), f32(x), dev, want_dY=False, gB_accum=run)
    assert gB3 is run and torch.equal(run, base + gB)


@pytest.mark.parametrize("E", [500, 18122, 7])
def test_bilinear_tail_adjoint_one_launch(E):
    """gn_bil_fused_bwd_f32 (dP = alpha g W2^T kept in LDS, then gB and dSm) is bit-identical to the K = 64 / N = 1024 GEMM
    followed by gn_bil_project_bwd_f32 without the Y gradient; also in the accumulate-into-gB form."""
    g = torch.Generator().manual_seed(E)
    S, C, I, O = 7, 64, 16, 64
    gr, W2, Sm, Bm = f32(rnd(g, E, O)), f32(rnd(g, I * C, O) / 8), f32(rnd(g, E, S, C)), f32(rnd(g, E, S, I))
    cpu, dev = _segplan(g, E, max(E, 8), 4)
    x = f32(rnd(g, max(E, 8), C))
    dP = K.gemm(gr, W2, alpha=0.7).reshape(E, I, C)
    gB0, dSm0, _ = K.bil_project_bwd(dP, Sm, Bm, x, dev, want_dY=False)
    gB1, dSm1 = K.bil_fused_bwd(gr, W2, Sm, Bm, 0.7)
    assert torch.equal(gB1, gB0) and torch.equal(dSm1, dSm0)
    ref = CK.bil_fused_bwd(gr.double().cpu(), W2.double().cpu(), Sm.double().cpu(), Bm.double().cpu(), 0.7)
    close(gB1, ref[0], atol=2e-4 * float(ref[0].abs().max())); close(dSm1, ref[1], atol=2e-4 * float(ref[1].abs().max()))
    base = f32(rnd(g, E, S, I))
    run = base.clone()
    gB2, _ = K.bil_fused_bwd(gr, W2, Sm, Bm, 0.7, gB_accum=run)
    assert gB2 is run and torch.equal(run, base + gB0)
    # the first product on the fp16 matrix pipe (W2 pre-split, g under one power-of-two scale per edge row): the same result to
    # fp32 rounding for cotangents of ANY magnitude, rows of very different size mixed
    planes = K.pack_weight_split(W2, fmt=1)
    e32 = [float((gB1.double().cpu() - ref[0]).abs().max()), float((dSm1.double().cpu() - ref[1]).abs().max())]
    for scale in (1.0, 1e-9, 3e-6, 1e5):
        rows = torch.logspace(0, -4, E, dtype=torch.float64)[:, None]
        gs = gr.double().cpu() * scale * rows
        rs = CK.bil_fused_bwd(gs, W2.double().cpu(), Sm.double().cpu(), Bm.double().cpu(), 0.7)
        gB3, dSm3 = K.bil_fused_bwd(f32(gs), W2, Sm, Bm, 0.7, W2_planes=planes)
        # per edge: error relative to that edge's own magnitude (the row scale makes it independent of the others)
        for got, want, bar in ((gB3, rs[0], e32[0]), (dSm3, rs[1], e32[1])):
            err = (got.double().cpu() - want).abs().flatten(1).max(dim=1).values
            mag = want.abs().flatten(1).max(dim=1).values.clamp_min(1e-300)
            ref_mag = float(ref[0 if got is gB3 else 1].abs().max())
            assert float((err / mag).max()) <= 4 * bar / ref_mag + 2e-6, (scale, float((err / mag).max()))


def test_quad_basis_fused_fwd_bwd():
    """Explicit (Q,49) harmonics + adjoint (the path of non-published tensor-basis widths; the published ones use the
    angle form).  Quadruplets are drawn WELL-CONDITIONED — bond lengths >= 0.8 A, sin of the polar angle c-a-b and of
    a-b-d >= 0.25 (the chain rule through atan2 / the projections divides by them) — so the bar is elementwise, for
    every quadruplet, not a quantile."""
    g = torch.Generator().manual_seed(31)
    n_atoms, Q = 30, 6000
    R = torch.rand(n_atoms, 3, generator=g, dtype=torch.float64) * 5.0
    idx = torch.stack([torch.randperm(n_atoms, generator=g)[:4] for _ in range(Q)])  # 4 distinct atoms each
    R32 = R.float().double()
    c, a, b, d = (R32[idx[:, i]] for i in range(4))

    def sin_between(u, v):
        return torch.linalg.cross(u, v).norm(dim=1) / (u.norm(dim=1) * v.norm(dim=1))
    ok = ((c - a).norm(dim=1) >= 0.8) & ((b - a).norm(dim=1) >= 0.8) & ((d - b).norm(dim=1) >= 0.8) \
        & (sin_between(c - a, b - a) >= 0.25) & (sin_between(a - b, d - b) >= 0.25)
    idx = idx[ok][:1500]
    Q = idx.shape[0]
    assert Q >= 1000
    qc, qa, qb, qd = (idx[:, i].contiguous().int() for i in range(4))
    Y = K.quad_basis_fwd(f32(R), qc.to(DEV), qa.to(DEV), qb.to(DEV), qd.to(DEV), 7)
    rY = CK.quad_basis_fwd(R32, qc, qa, qb, qd, 7)
    close(Y, rY, rtol=1e-4, atol=2e-5)
    gY = rnd(g, Q, 49)
    G = K.quad_basis_bwd(f32(gY), f32(R), qc.to(DEV), qa.to(DEV), qb.to(DEV), qd.to(DEV), 7)
    rG = CK.quad_basis_bwd(gY, R32, qc, qa, qb, qd, 7)
    for name, out, ref in zip("cbd", G, rG):
        err = (out.cpu().double() - ref).abs()
        scale = float(ref.abs().median())
        print(f"quad_basis_bwd G{name}: median err {float(err.median()) / scale:.2e}, p99 {float(torch.quantile(err, 0.99)) / scale:.2e}, "
              f"max {float(err.max()) / scale:.2e} (x median |ref| = {scale:.3e})")
        assert torch.isfinite(out).all()
        assert float(err.median()) <= 2e-5 * scale
        assert float(err.max()) <= 2e-3 * scale


@pytest.mark.parametrize("Kd,M,N", [(1024, 128, 128), (17800, 128, 128), (17801, 128, 16), (5003, 64, 128),
                                    (18000, 1024, 64), (1024, 1, 128), (3000, 16, 6), (2500, 42, 112),
                                    (33, 128, 128), (1, 5, 3), (100000, 128, 128)])
def test_gemm_tn_weight_gradient_shapes(Kd, M, N):
    """gn_gemm_tn_f32: A^T B over the edge/atom dimension for every (M, N) the model's weight gradients have."""
    g = torch.Generator().manual_seed(Kd + M)
    A, Bm = rnd(g, Kd, M), rnd(g, Kd, N)
    ref = 0.5 * (A.t() @ Bm)
    tol = dict(rtol=1e-5, atol=2e-5 * np.sqrt(Kd))
    close(K.gemm(f32(A), f32(Bm), True, True, alpha=0.5), ref, **tol)  # routed to the split-K kernel
    for splitk in (1, 2, 7, 64):
        close(K.gemm_tn(f32(A), f32(Bm), alpha=0.5, splitk=splitk), ref, **tol)
    again = K.gemm_tn(f32(A), f32(Bm), alpha=0.5)
    assert torch.equal(again, K.gemm_tn(f32(A), f32(Bm), alpha=0.5))  # deterministic (no atomics)


def test_gemm_tn_strided_operands_and_empty_contraction():
    g = torch.Generator().manual_seed(77)
    X = rnd(g, 4000, 384)  # column slices of a concatenated activation: row stride 384
    Xd = f32(X)
    close(K.gemm_tn(Xd[:, 128:256], Xd[:, 256:320]), X[:, 128:256].t() @ X[:, 256:320], rtol=1e-5, atol=2e-3)
    close(K.gemm_tn(Xd[:, 1:43], Xd[:, 130:258]), X[:, 1:43].t() @ X[:, 130:258], rtol=1e-5, atol=2e-3)
    out = K.gemm_tn(f32(torch.zeros(0, 128)), f32(torch.zeros(0, 16)))
    assert out.shape == (128, 16) and float(out.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,Kd", [(18000, 128, 128), (18001, 64, 1024), (5000, 32, 128), (18122, 16, 128),
                                    (1024, 128, 64), (700, 384, 128), (4097, 128, 20)])
def test_gemm_k_major_weight_operand(M, N, Kd):
    """x @ B with B (K,N) — the input-gradient product of every Dense — on the k-major-staged 8-wave kernel,
    plain and with the fused prologue/epilogue."""
    g = torch.Generator().manual_seed(M + N + Kd)
    A, Bm = rnd(g, M, Kd), rnd(g, Kd, N) / np.sqrt(Kd)
    close(K.gemm(f32(A), f32(Bm), False, True), A @ Bm, rtol=1e-5, atol=2e-5 * np.sqrt(Kd))
    pre, mul, res = rnd(g, M, Kd), rnd(g, M, N), rnd(g, M, N)
    kw = dict(act=True, pre_out=True, alpha=0.7, beta=0.5)
    ref_y, ref_z = CK.gemm(A, Bm, False, True, a_dact_pre=pre, mul=mul, res=res, **kw)
    y, z = K.gemm(f32(A), f32(Bm), False, True, a_dact_pre=f32(pre), mul=f32(mul), res=f32(res), **kw)
    close(z, ref_z, atol=1e-4)
    close(y, ref_y, atol=1e-4)


@pytest.mark.parametrize("n", [1, 7, 64, 1000, 128 * 1001 + 3])
@pytest.mark.parametrize("k", [-1, 0, 1, 2, 3])
def test_pointwise_product_primitive(n, k):
    """gn_pm_f32: c * ssilu^(k)(z) * a * b * d with optional factors (the composite path's only pointwise op)."""
    g = torch.Generator().manual_seed(n + k)
    z, a, b, d = (rnd(g, n) * 2 for _ in range(4))
    for fs in ((a,), (a, b), (a, b, d)) + (((),) if k >= 0 else ()):
        ref = CK.pm(z, k, *fs, c=0.7) if fs else CK.pm(z, k, c=0.7)
        out = K.pm(f32(z), k, *[f32(t) for t in fs], c=0.7)
        # f'' and f''' change sign: compare against the magnitude of the factors, not of the (cancelling) result
        scale = float(torch.stack([t.abs() for t in fs]).prod(0).max()) if fs else 1.0
        close(out, ref, rtol=2e-5, atol=3e-6 * max(1.0, scale))
    # unaligned views take the scalar path
    if n > 8:
        zz, aa = f32(z)[1:], f32(a)[1:]
        close(K.pm(zz, max(k, 0), aa), CK.pm(z[1:], max(k, 0), a[1:]), rtol=2e-5, atol=3e-6 * max(1.0, float(a.abs().max())))


@pytest.mark.parametrize("S,C,I,E,J,mk", [(7, 64, 16, 300, 300, 12), (49, 32, 32, 40, 200, 70)])
def test_bilinear_adjoint_accumulates_y_gradient(S, C, I, E, J, mk):
    """gn_bil_project_bwd_acc_f32: the Y gradient of a second consumer is added into the first one's buffer."""
    g = torch.Generator().manual_seed(S + E)
    seg = torch.randint(0, mk, (E,), generator=g)
    T = int(seg.sum())
    red = torch.repeat_interleave(torch.arange(E), seg)
    exp = torch.randint(0, J, (T,), generator=g)
    from gemnet_pytorch_amd.graph import SegmentPlan
    dev, cpu = SegmentPlan(red.to(DEV), exp.to(DEV), E, J), SegmentPlan(red, exp, E, J)
    dP1, dP2, Sm, Bm, x = rnd(g, E, I, C), rnd(g, E, I, C), rnd(g, E, S, C), rnd(g, E, S, I), rnd(g, J, C)
    _, _, dY = K.bil_project_bwd(f32(dP1), f32(Sm), f32(Bm), f32(x), dev)
    _, _, dY2 = K.bil_project_bwd(f32(dP2), f32(Sm), f32(Bm), f32(x), dev, dY_accum=dY)
    assert dY2.data_ptr() == dY.data_ptr()
    ref = CK.bil_project_bwd(dP1, Sm, Bm, x, cpu)[2] + CK.bil_project_bwd(dP2, Sm, Bm, x, cpu)[2]
    close(dY2, ref, atol=2e-4 * max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("E,mk", [(500, 18), (77, 5), (16, 30), (1, 3)])
def test_bilinear_forward_fused_with_the_final_projection(E, mk):
    """gn_bil_fused_fwd_f32: K1 + K2 + K3 in one launch (P parked in LDS) == gn_bil_reduce_project_f32 + gn_gemm_f32."""
    S, C, I, O = 7, 64, 16, 64
    g = torch.Generator().manual_seed(E)
    cpu, dev = _segplan(g, E, E, mk)
    Y, x, Bm = rnd(g, cpu.size, S), rnd(g, E, C), rnd(g, E, S, I)
    W2T = rnd(g, O, I * C) / 32
    Sm, out = K.bil_fused_fwd(f32(Y), f32(x), f32(Bm), f32(W2T), dev, alpha=0.6)
    rSm, rout = CK.bil_fused_fwd(Y, x, Bm, W2T, cpu, alpha=0.6)
    close(Sm, rSm, atol=1e-4)
    close(out, rout, atol=3e-4 * max(1.0, float(rout.abs().max())))
    Sm2, P2 = K.bil_reduce_project(f32(Y), f32(x), f32(Bm), dev)
    assert torch.equal(Sm, Sm2)
    # K3 on the fp16 matrix pipe (the weight given as two fp16 planes in fragment order, P split in LDS): the same product to
    # fp32 rounding, Sm bit for bit; also with activations of 1e3 (P ~ 1e5 would leave fp16: 2e4 still inside)
    planes = K.pack_weight_split(f32(W2T), fmt=1)
    Sm3, out3 = K.bil_fused_fwd(f32(Y), f32(x), f32(Bm), f32(W2T), dev, alpha=0.6, W2T_planes=planes)
    assert torch.equal(Sm3, Sm)
    close(out3, rout, atol=3e-4 * max(1.0, float(rout.abs().max())))
    err16 = float((out3.double().cpu() - rout).abs().max())
    err32 = float((out.double().cpu() - rout).abs().max())
    print(f"K3 f32 MFMA max err {err32:.2e}, split fp16 {err16:.2e} (|out| max {float(rout.abs().max()):.2e})")
    assert err16 <= 4 * err32 + 1e-6 * float(rout.abs().max())
    _, out4 = K.bil_fused_fwd(f32(Y), f32(x * 300.0), f32(Bm), f32(W2T), dev, alpha=0.6, W2T_planes=planes)
    close(out4, rout * 300.0, atol=3e-4 * 300.0 * max(1.0, float(rout.abs().max())))
    with pytest.raises(RuntimeError):
        K.bil_fused_fwd(f32(Y), f32(x)[:, :32].contiguous(), f32(Bm), f32(W2T)[:, :512].contiguous(), dev)


@pytest.mark.parametrize("shape", [(49, 32, 32, 50, 260, 75), (7, 64, 16, 90, 90, 40)])
@pytest.mark.parametrize("nb", [1, 2, 4])
def test_bilinear_deferred_y_gradient_of_several_blocks(nb, shape):
    """gn_bil_project_bwd with dY == NULL + gn_bil_dy_multi_f32: the Y gradient of nb blocks that share one basis
    (tensor basis of the quadruplets, spherical basis of the triplets) in one pass."""
    S, C, I, E, J, mk = shape
    g = torch.Generator().manual_seed(nb)
    seg = torch.randint(0, mk, (E,), generator=g)
    seg[3] = 0
    T = int(seg.sum())
    red = torch.repeat_interleave(torch.arange(E), seg)
    exp = torch.randint(0, J, (T,), generator=g)
    from gemnet_pytorch_amd.graph import SegmentPlan
    dev, cpu = SegmentPlan(red.to(DEV), exp.to(DEV), E, J), SegmentPlan(red, exp, E, J)
    ref = 0
    dS_dev, x_dev = [], []
    for _ in range(nb):
        dP, Sm, Bm, x = rnd(g, E, I, C), rnd(g, E, S, C), rnd(g, E, S, I), rnd(g, J, C)
        gB, dSm, dY = K.bil_project_bwd(f32(dP), f32(Sm), f32(Bm), f32(x), dev, want_dY=False)
        rB, rS, rY = CK.bil_project_bwd(dP, Sm, Bm, x, cpu)
        assert dY is None
        close(gB, rB, atol=2e-4 * float(rB.abs().max()))
        close(dSm, rS, atol=2e-4 * float(rS.abs().max()))
        ref = ref + rY
        dS_dev.append(dSm)
        x_dev.append(f32(x))
    close(K.bil_dy_multi(dS_dev, x_dev, dev), ref, atol=3e-4 * max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("n_atoms,deg", [(1024, 18), (37, 5), (3, 0)])
def test_rbf_aggregate_fused_vs_reference_ops(n_atoms, deg):
    """gn_rbf_aggregate_fwd/bwd (Dense over rbf + Hadamard + scatter-add of AtomUpdateBlock, one pass each) against the
    float64 composition, including atoms without incoming edges and an unsorted target index."""
    g = torch.Generator().manual_seed(n_atoms)
    E = max(n_atoms * deg, 1)
    id_a = torch.randint(0, n_atoms, (E,), generator=g)
    if n_atoms > 5:
        id_a[id_a == 2] = 3                       # atom 2 receives nothing
    m, rbf, W, go = rnd(g, E, 128), rnd(g, E, 16), rnd(g, 128, 16) / 4, rnd(g, n_atoms, 128)
    perm = torch.argsort(id_a, stable=True)
    seg = torch.searchsorted(id_a[perm].contiguous(), torch.arange(n_atoms + 1)).to(torch.int32)
    ref = torch.zeros(n_atoms, 128, dtype=torch.float64).index_add(0, id_a, m * (rbf @ W.t())) * 0.37
    out = K.rbf_aggregate_fwd(f32(m), f32(rbf), f32(W), perm.to(torch.int32).to(DEV), seg.to(DEV), n_atoms, 0.37)
    close(out, ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))
    gm, gr = K.rbf_aggregate_bwd(f32(go), f32(m), f32(rbf), f32(W), id_a.to(torch.int32).to(DEV), 0.37)
    gsel = go[id_a] * 0.37
    close(gm, gsel * (rbf @ W.t()), rtol=1e-5, atol=1e-5)
    close(gr, (gsel * m) @ W, rtol=1e-5, atol=2e-5 * max(1.0, float(((gsel * m) @ W).abs().max())))
    gm2, gr2 = K.rbf_aggregate_bwd(f32(go), f32(m), f32(rbf), f32(W), id_a.to(torch.int32).to(DEV), 0.37, want_m=False)
    assert gm2 is None and torch.equal(gr2, gr)
    # accumulate form (ops.accumulate_gradient): the contribution joins a running gradient in the same pass, in place
    base_m, base_r = rnd(g, E, 128), rnd(g, E, 16)
    run_m, run_r = f32(base_m), f32(base_r)
    gm3, gr3 = K.rbf_aggregate_bwd(f32(go), f32(m), f32(rbf), f32(W), id_a.to(torch.int32).to(DEV), 0.37, acc_m=run_m, acc_rbf=run_r)
    assert gm3 is run_m and gr3 is run_r
    assert torch.equal(gm3, f32(base_m) + gm) and torch.equal(gr3, f32(base_r) + gr)


@pytest.mark.parametrize("N,T,C", [(1024, 300000, 3), (37, 500, 3), (5, 0, 1), (200, 9000, 4)])
def test_segsum_multi_matches_separate_sums(N, T, C):
    """gn_segsum_multi_f32 (the dE/dR assembly: up to four signed CSR sums over the same atoms in one launch) against the
    float64 composition, with unsorted and sorted (perm = None) index lists and empty rows; deterministic."""
    g = torch.Generator().manual_seed(N + T)
    terms_cpu, terms_dev = [], []
    for k, sign in enumerate((1.0, 1.0, -1.0, -0.5)):
        idx = torch.randint(0, N, (T,), generator=g)
        if k == 1:
            idx = torch.sort(idx).values
        if N > 5 and T:
            idx[idx == 2] = 3                      # row 2 empty
        y = rnd(g, T, C)
        if k == 1:
            perm = None
            seg = torch.searchsorted(idx.contiguous(), torch.arange(N + 1)).to(torch.int32)
        else:
            perm = torch.argsort(idx, stable=True)
            seg = torch.searchsorted(idx[perm].contiguous(), torch.arange(N + 1)).to(torch.int32)
            perm = perm.to(torch.int32)
        terms_cpu.append((y, idx, sign))
        terms_dev.append((f32(y), None if perm is None else perm.to(DEV), seg.to(DEV), sign))
    ref = sum(sg * torch.zeros(N, C, dtype=torch.float64).index_add(0, idx, y) for y, idx, sg in terms_cpu)
    out = K.segsum_multi(terms_dev, N)
    close(out, ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))
    assert torch.equal(out, K.segsum_multi(terms_dev, N))
    out2 = K.segsum_multi(terms_dev[:2], N)
    close(out2, sum(sg * torch.zeros(N, C, dtype=torch.float64).index_add(0, idx, y) for y, idx, sg in terms_cpu[:2]),
          rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))


def test_gemm_accumulates_in_place():
    """K.gemm(out = res2): C += A @ W^T element by element in one launch (the running gradient of a tensor with several
    fused consumers); same bits as the separate add."""
    g = torch.Generator().manual_seed(5)
    A, W, base = f32(rnd(g, 777, 64)), f32(rnd(g, 128, 64)), f32(rnd(g, 777, 128))
    ref = K.gemm(A, W, res2=base)
    run = base.clone()
    out = K.gemm(A, W, res2=run, out=run)
    assert out is run and torch.equal(run, ref)


@pytest.fixture(params=[7, 5, 0], ids=["f16-all (default)", "f16-K1+expand", "f32-mfma"])
def ang_arithmetic(request, monkeypatch):
    """kernels.ANG_F16_MASK: which of the angle-form kernels get `arith = GN_ANG_F16` (products on the fp16 matrix pipe with
    split operands) — an argument of each launch since ABI 13."""
    monkeypatch.setattr(K, "ANG_F16_MASK", request.param)
    yield request.param


@pytest.mark.parametrize("E,J,mk", [(60, 300, 80), (9, 40, 700), (33, 120, 3)])
def test_tensor_basis_angle_form_kernels(E, J, mk, ang_arithmetic):
    """The *_ang kernels (Y_lm rebuilt in LDS from (sin, cos) of the two angles, 16 B per quadruplet) against the
    float64 restatements working on the explicit (Q,49) harmonics: K1 + K2 forward, the x-adjoint, and the gradient
    w.r.t. the two angles of several blocks at once; segments of 0 .. 700 quadruplets (tiles of 64 and 16)."""
    g = torch.Generator().manual_seed(E * mk)
    S, C, I = 49, 32, 32
    cpu, dev = _segplan(g, E, J, mk)
    Q = cpu.size
    th, ph = torch.rand(Q, generator=g, dtype=torch.float64) * 3.1, torch.rand(Q, generator=g, dtype=torch.float64) * 3.1
    ang = torch.stack([torch.sin(th), torch.cos(th), torch.sin(ph), torch.cos(ph)], 1)
    x, B_, D = rnd(g, J, C), rnd(g, E, S, I), rnd(g, E, S, C)
    Sm, P = K.bil_reduce_project(f32(ang), f32(x), f32(B_), dev)
    Sm_ref, P_ref = CK.bil_reduce_project(ang, x, B_, cpu)
    close(Sm, Sm_ref, atol=2e-5 * max(1.0, float(Sm_ref.abs().max())))
    close(P, P_ref, atol=2e-5 * max(1.0, float(P_ref.abs().max())))
    dx_ref = CK.bil_reduce_t(ang, D, cpu)
    close(K.bil_reduce_t(f32(ang), f32(D), dev), dx_ref, atol=2e-5 * max(1.0, float(dx_ref.abs().max())))
    for scale in (1e-9, 3e-6, 1e5):     # a cotangent of any magnitude: the split-fp16 product scales the block per edge
        rows = torch.logspace(0, -3, E, dtype=torch.float64)[:, None, None]       # and edges of very different size
        close(K.bil_reduce_t(f32(ang), f32(D * scale * rows), dev), CK.bil_reduce_t(ang, D * scale * rows, cpu),
              atol=2e-5 * scale * float(dx_ref.abs().max()), rtol=2e-4)
    Ds, xs = [rnd(g, E, S, C) for _ in range(3)], [rnd(g, J, C) for _ in range(3)]
    for nb in (1, 3):
        g_ref = CK.bil_dy_multi(Ds[:nb], xs[:nb], cpu, ang=ang)
        got = K.bil_dy_multi([f32(d) for d in Ds[:nb]], [f32(v) for v in xs[:nb]], dev, ang=f32(ang))
        close(got, g_ref, atol=3e-5 * max(1.0, float(g_ref.abs().max())))
        # the cotangent blocks may have ANY magnitude (1e-6 on the quadruplet path of a force pass): the split-fp16 form scales
        # them per edge by a power of two — same relative accuracy from 1e-9 to 1e5, blocks of very different size mixed
        for scale in (1e-9, 3e-6, 1e5):
            sc = [scale * (10.0 ** (-2 * i)) for i in range(nb)]
            got_s = K.bil_dy_multi([f32(d * c) for d, c in zip(Ds[:nb], sc)], [f32(v) for v in xs[:nb]], dev, ang=f32(ang))
            ref_s = CK.bil_dy_multi([d * c for d, c in zip(Ds[:nb], sc)], xs[:nb], cpu, ang=ang)
            close(got_s, ref_s, atol=3e-5 * float(ref_s.abs().max()), rtol=0)


@pytest.mark.parametrize("E,J,mk", [(60, 300, 80), (9, 40, 700), (33, 120, 3)])
def test_tensor_basis_tangent_kernels_of_force_training(E, J, mk):
    """gn_bil_reduce_project_ang_tan_f32 / gn_bil_expand_ang_tan_f32 — the S3 / S4 sweeps of GemNet-Q force training with the
    tangent rows dY = Y_theta dPhi + Y_phi dTheta rebuilt in-kernel by dual numbers — against the float64 restatements that
    differentiate the oracle's (Q,49) harmonics, for every combination of optional operands and tangents of any magnitude
    (they carry the scale of the caller's loss)."""
    g = torch.Generator().manual_seed(E * mk + 1)
    S, C, I = 49, 32, 32
    cpu, dev = _segplan(g, E, J, mk)
    Q = cpu.size
    th, ph = torch.rand(Q, generator=g, dtype=torch.float64) * 3.0 + 0.05, torch.rand(Q, generator=g, dtype=torch.float64) * 3.1
    ang = torch.stack([torch.sin(th), torch.cos(th), torch.sin(ph), torch.cos(ph)], 1)
    x, tx, B_, tB, Sm = rnd(g, J, C), rnd(g, J, C), rnd(g, E, S, I), rnd(g, E, S, I), rnd(g, E, S, C)
    D1, D2 = rnd(g, E, S, C), rnd(g, E, S, C)
    for scale in (1.0, 3e-6, 1e4):
        tang = torch.zeros(Q, 4, dtype=torch.float64)
        tang[:, 0:2] = rnd(g, Q, 2) * scale
        for use_t, use_tx, use_tB in ((True, True, True), (True, False, False), (False, True, True), (True, True, False)):
            a = (tang if use_t else None, x, tx * scale if use_tx else None, B_, tB * scale if use_tB else None, Sm)
            Smd_ref, Pd_ref = CK.bil_reduce_project_tan(ang, a[0], a[1], a[2], a[3], a[4], a[5], cpu)
            Smd, Pd = K.bil_reduce_project_tan(f32(ang), None if a[0] is None else f32(a[0]), f32(a[1]),
                                               None if a[2] is None else f32(a[2]), f32(a[3]),
                                               None if a[4] is None else f32(a[4]), f32(a[5]), dev)
            close(Smd, Smd_ref, atol=3e-5 * max(1e-30, float(Smd_ref.abs().max())), rtol=0)
            close(Pd, Pd_ref, atol=3e-5 * max(1e-30, float(Pd_ref.abs().max())), rtol=0)
        Smd_only, none = K.bil_reduce_project_tan(f32(ang), f32(tang), f32(x), None, f32(B_), None, None, dev, want_P=False)
        assert none is None
        ref_only = CK.bil_reduce_project_tan(ang, tang, x, None, B_, None, None, cpu, want_P=False)[0]
        close(Smd_only, ref_only, atol=3e-5 * float(ref_only.abs().max()), rtol=0)
        for use_D1 in (True, False):
            ref = CK.bil_reduce_t_tan(ang, tang, D1 * scale if use_D1 else None, D2, cpu)
            got = K.bil_reduce_t_tan(f32(ang), f32(tang), f32(D1 * scale) if use_D1 else None, f32(D2), dev)
            close(got, ref, atol=3e-5 * float(ref.abs().max()), rtol=0)


def test_quad_angles_tangent_kernel():
    """gn_quad_angles_jvp_f32: (dPhi_cab, dTheta_cabd) along a position tangent — the double backward of gn_quad_angles_bwd —
    against forward-mode differentiation of the float64 geometry (gemnet.py:334-418), and consistent with the first adjoint:
    <tang, g_ang> == <tR, G(g_ang)> for random g_ang (the two kernels are transposes of one Jacobian)."""
    g = torch.Generator().manual_seed(5)
    A, Q = 40, 3000
    R = rnd(g, A, 3) * 2.0
    tR = rnd(g, A, 3)
    idx = [torch.randint(0, A, (Q,), generator=g) for _ in range(4)]
    keep = (idx[0] != idx[1]) & (idx[1] != idx[2]) & (idx[2] != idx[3]) & (idx[0] != idx[2]) & (idx[1] != idx[3])
    qc, qa, qb, qd = (i[keep].to(torch.int32) for i in idx)
    dev_idx = [t.to(DEV) for t in (qc, qa, qb, qd)]
    ref = CK.quad_angles_jvp(R, tR, qc, qa, qb, qd)
    got = K.quad_angles_jvp(f32(R), f32(tR), *dev_idx)
    assert bool((got[:, 2:] == 0).all())
    ok = ref.abs().max(dim=1).values < 50          # (nearly collinear quadruplets: derivatives of order 1 / sin)
    close(got[ok.to(DEV)], ref[ok], atol=2e-4 * float(ref[ok].abs().max()), rtol=2e-3)
    g_ang = torch.zeros(qc.shape[0], 4, dtype=torch.float64)
    g_ang[:, 0:2] = rnd(g, qc.shape[0], 2)
    g_ang[~ok] = 0
    Gc, Gb, Gd = K.quad_angles_bwd(f32(g_ang), f32(R), *dev_idx)
    Gc, Gb, Gd = (t.double().cpu() for t in (Gc, Gb, Gd))
    lhs = float((got.double().cpu() * g_ang).sum())
    rhs = float((Gc * tR[qc.long()]).sum() + (Gb * tR[qb.long()]).sum() + (Gd * tR[qd.long()]).sum()
                - ((Gc + Gb + Gd) * tR[qa.long()]).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


@pytest.mark.parametrize("n_mol,n_atoms", [(3, 12), (2, 32)])
def test_fused_per_atom_x_adjoint_of_the_tensor_basis(n_mol, n_atoms, monkeypatch):
    """gn_bil_expand_atoms_ang_f32 — one workgroup per target atom, the atom's expand rows summed in LDS edge by edge, no
    per-quadruplet rows in memory — on the real quadruplet structure of a GemNet-Q batch (the only structure it is defined
    for: reduce edge and intermediate triplet of a quadruplet end in the same atom) against the float64 restatement on the
    explicit (Q, 49) harmonics and against the two-pass form (gn_bil_expand_ang_f32 + segmented sum); run twice: bitwise."""
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.data_container import DataContainer
    ds = make_dataset(n_mol, n_atoms, config=2)
    b = DataContainer.from_arrays(dict(ds), 5.0, 10.0, triplets_only=False)[list(range(n_mol))]
    inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
    monkeypatch.setattr(K, "USE_ATOM_BLOCKS", True)      # (the plan only builds the per-atom structure for its one consumer)
    cpu = GraphPlan.from_inputs(dict(inputs), False).quad
    dev_plan = GraphPlan.from_inputs({k: v.to(DEV) for k, v in inputs.items()}, False).warm()
    dev = dev_plan.quad
    a_perm, a_seg, j_off, max_J = dev.atom_blocks
    assert 0 < max_J <= K.ATOM_BLOCK_MAX_ROWS and int(j_off[-1]) == dev.n_expand
    g = torch.Generator().manual_seed(n_atoms)
    Q, S, C = cpu.size, 49, 32
    th, ph = torch.rand(Q, generator=g, dtype=torch.float64) * 3.1, torch.rand(Q, generator=g, dtype=torch.float64) * 3.1
    ang = torch.stack([torch.sin(th), torch.cos(th), torch.sin(ph), torch.cos(ph)], 1)
    D = rnd(g, dev.n_reduce, S, C)
    ref = CK.bil_reduce_t(ang, D, cpu)
    monkeypatch.setattr(K, "USE_ATOM_BLOCKS", True)
    got = K.bil_reduce_t(f32(ang), f32(D), dev)
    close(got, ref, atol=2e-5 * max(1.0, float(ref.abs().max())))
    assert torch.equal(got, K.bil_reduce_t(f32(ang), f32(D), dev))
    monkeypatch.setattr(K, "USE_ATOM_BLOCKS", False)
    two_pass = K.bil_reduce_t(f32(ang), f32(D), dev)
    close(got, two_pass.double().cpu(), atol=2e-5 * max(1.0, float(ref.abs().max())))
    print(f"{n_mol} x {n_atoms}: {Q} quadruplets, max |J_a| {max_J}; fused vs float64 {float((got.double().cpu() - ref).abs().max()):.2e}, "
          f"two-pass vs float64 {float((two_pass.double().cpu() - ref).abs().max()):.2e}")


@pytest.mark.parametrize("arith_f16", [True, False], ids=["fp16-planes", "f32-mfma"])
@pytest.mark.parametrize("n_mol,n_atoms", [(3, 12), (2, 32), (1, 64), (5, 7)])
def test_row_stationary_x_adjoint_of_the_tensor_basis(n_mol, n_atoms, arith_f16, monkeypatch):
    """gn_bil_expand_rows_ang_f32 (round 6; the default form of the quadruplet x-adjoint) — a wave owns 32 expand rows of one
    target atom and walks the atom's reduce edges with the accumulators in registers; the quadruplet of (edge, row) from the
    dense per-atom grid of graph.SegmentPlan.row_grid — on the real quadruplet structure of GemNet-Q batches (ragged: 64-atom
    molecules with > 32-row tiles per atom, 7-atom molecules with a single partial tile) against the float64 restatement on the
    explicit (Q, 49) harmonics and against the two-pass form (gn_bil_expand_ang_f32 + segmented sum); run twice: bitwise.
    Cotangent blocks of very different magnitude per edge (1e-6 .. 1e3): the per-edge scale of the fp16 planes."""
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.data_container import DataContainer
    ds = make_dataset(n_mol, n_atoms, config=2)
    b = DataContainer.from_arrays(dict(ds), 5.0, 10.0, triplets_only=False)[list(range(n_mol))]
    inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
    monkeypatch.setattr(K, "USE_ROW_GRID", True)
    monkeypatch.setattr(K, "ANG_F16_MASK", 7 if arith_f16 else 0)
    cpu = GraphPlan.from_inputs(dict(inputs), False).quad
    dev = GraphPlan.from_inputs({k: v.to(DEV) for k, v in inputs.items()}, False).warm().quad
    rg = dev.row_grid
    assert rg is not None
    a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0, n_tasks = rg
    # the grid holds every quadruplet exactly once
    qm = qmap.cpu()
    assert int((qm >= 0).sum()) == cpu.size and torch.equal(torch.sort(qm[qm >= 0]).values, torch.arange(cpu.size, dtype=torch.int32))
    assert int(j_off[-1]) == dev.n_expand and n_tasks == int(task_atom.shape[0]) > 0
    g = torch.Generator().manual_seed(n_atoms)
    Q, S, C = cpu.size, 49, 32
    th, ph = torch.rand(Q, generator=g, dtype=torch.float64) * 3.1, torch.rand(Q, generator=g, dtype=torch.float64) * 3.1
    ang = torch.stack([torch.sin(th), torch.cos(th), torch.sin(ph), torch.cos(ph)], 1)
    D = rnd(g, dev.n_reduce, S, C) * (10.0 ** torch.randint(-6, 4, (dev.n_reduce, 1, 1), generator=g).double())
    ref = CK.bil_reduce_t(ang, D, cpu)
    got = K.bil_reduce_t(f32(ang), f32(D), dev)
    # error relative to what a row sums up (its largest contribution sets the rounding level)
    row_scale = CK.bil_reduce_t(ang.abs() * 0 + ang, D.abs(), cpu).abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    err = float(((got.double().cpu() - ref).abs() / row_scale).max())
    assert torch.equal(got, K.bil_reduce_t(f32(ang), f32(D), dev))
    monkeypatch.setattr(K, "USE_ROW_GRID", False)
    two_pass = K.bil_reduce_t(f32(ang), f32(D), dev)
    err2 = float(((two_pass.double().cpu() - ref).abs() / row_scale).max())
    print(f"{n_mol} x {n_atoms} [{'fp16 planes' if arith_f16 else 'f32 MFMA'}]: {Q} quadruplets, {n_tasks} row tiles, grid {int(qmap.shape[0])}; "
          f"row-stationary vs float64 {err:.2e} of the row scale, two-pass {err2:.2e}")
    assert err <= 5e-5 and err <= 4 * max(err2, 1e-6)
    if not arith_f16:
        # the tangent form of the training step's S4 (gn_bil_expand_rows_ang_tan_f32: Y D1 + dY D2 in one pass) on the same grid
        tang = torch.cat([rnd(g, Q, 2), torch.zeros(Q, 2, dtype=torch.float64)], 1)
        D1 = rnd(g, dev.n_reduce, S, C)
        for use_D1 in (True, False):
            ref_t = CK.bil_reduce_t_tan(ang, tang, D1 if use_D1 else None, D, cpu)
            monkeypatch.setattr(K, "USE_ROW_GRID", True)
            got_t = K.bil_reduce_t_tan(f32(ang), f32(tang), f32(D1) if use_D1 else None, f32(D), dev)
            assert torch.equal(got_t, K.bil_reduce_t_tan(f32(ang), f32(tang), f32(D1) if use_D1 else None, f32(D), dev))
            monkeypatch.setattr(K, "USE_ROW_GRID", False)
            two_t = K.bil_reduce_t_tan(f32(ang), f32(tang), f32(D1) if use_D1 else None, f32(D), dev)
            sc = float(ref_t.abs().max())
            e1, e2 = float((got_t.double().cpu() - ref_t).abs().max()) / sc, float((two_t.double().cpu() - ref_t).abs().max()) / sc
            print(f"    tangent form (D1 {'given' if use_D1 else 'absent'}): row-stationary {e1:.2e}, two-pass {e2:.2e} of max|dx|")
            assert e1 <= 3e-5 and e1 <= 4 * max(e2, 1e-6)


def test_quad_angles_geometry_fwd_bwd():
    """gn_quad_angles_fwd / bwd: (sin, cos) of Phi_cab, Theta_cabd per quadruplet and the force contributions of a
    gradient given w.r.t. the two angles, against autograd on the float64 geometry (gemnet.py:334-418)."""
    g = torch.Generator().manual_seed(3)
    A, Q = 40, 3000
    R = rnd(g, A, 3) * 2.0
    idx = [torch.randint(0, A, (Q,), generator=g) for _ in range(4)]
    keep = (idx[0] != idx[1]) & (idx[1] != idx[2]) & (idx[2] != idx[3]) & (idx[0] != idx[2]) & (idx[1] != idx[3])
    qc, qa, qb, qd = (i[keep].to(torch.int32) for i in idx)
    ang_ref = CK.quad_angles_fwd(R, qc, qa, qb, qd)
    dev_idx = [t.to(DEV) for t in (qc, qa, qb, qd)]
    ang = K.quad_angles_fwd(f32(R), *dev_idx)
    close(ang, ang_ref, atol=2e-5)
    g_ang = torch.zeros(qc.shape[0], 4, dtype=torch.float64)
    g_ang[:, :2] = rnd(g, qc.shape[0], 2)
    ref = CK.quad_angles_bwd(g_ang, R, qc, qa, qb, qd)
    got = K.quad_angles_bwd(f32(g_ang), f32(R), *dev_idx)
    # exclude the nearly collinear configurations (sin of either angle < 0.05: the angle gradient ~ 1 / sin is
    # ill-conditioned in fp32 there; the clamp case itself is covered by test_trip_basis / the linear-molecule goldens)
    okq = (ang_ref[:, 0].abs() > 0.05) & (ang_ref[:, 2].abs() > 0.05)
    assert int(okq.sum()) > 0.8 * okq.shape[0]
    for a, b in zip(got, ref):
        err = (a.double().cpu() - b).abs()[okq]
        scale = b.abs()[okq].clamp(min=1.0)
        assert float((err / scale).median()) <= 1e-5 and float((err / scale).max()) <= 5e-3
    Gc, Gbd = K.quad_angles_bwd(f32(g_ang), f32(R), *dev_idx, packed=True)
    assert torch.equal(Gc, got[0]) and torch.equal(Gbd[:, 0:3], got[1]) and torch.equal(Gbd[:, 4:7], got[2])


def test_bilinear_training_kernels_extended_forms():
    """The two kernel extensions of the fused training step's bilinear layer (ops_train._Bilinear2):
    gn_bil_reduce_project2_f32 (Sm starts from Sm_init, P takes a second K2 term B2^T Sm2, P optional) and the dSm
    accumulation flag of gn_bil_project_bwd_acc_f32, against the float64 restatement on a ragged CSR plan."""
    g = torch.Generator().manual_seed(77)
    E, J, S, C, I = 301, 120, 7, 64, 16
    counts = torch.randint(0, 40, (E,), generator=g)
    counts[5] = 0
    reduce_idx = torch.repeat_interleave(torch.arange(E), counts)
    T = int(reduce_idx.shape[0])
    expand_idx = torch.randint(0, J, (T,), generator=g)
    sp_c = SegmentPlan(reduce_idx, expand_idx, E, J)
    sp_d = SegmentPlan(reduce_idx.to(DEV), expand_idx.to(DEV), E, J)
    Y, Y2, x, x2 = rnd(g, T, S), rnd(g, T, S), rnd(g, J, C), rnd(g, J, C)
    B, B2, Sm2 = rnd(g, E, S, I), rnd(g, E, S, I), rnd(g, E, S, C)
    # (a) two-call tangent form: Sm = K1(Y2, x) + K1(Y, x2), P = B^T Sm + B2^T Sm2
    ref_a, _ = CK.bil_reduce_project(Y2, x, B, sp_c, want_P=False)
    ref_Sm, ref_P = CK.bil_reduce_project(Y, x2, B, sp_c, Sm_init=ref_a, B2=B2, Sm2=Sm2)
    Sm_a, none = K.bil_reduce_project(f32(Y2), f32(x), f32(B), sp_d, want_P=False)
    assert none is None
    close(Sm_a, ref_a, atol=2e-4 * float(ref_a.abs().max()))
    Sm, P = K.bil_reduce_project(f32(Y), f32(x2), f32(B), sp_d, Sm_init=Sm_a, B2=f32(B2), Sm2=f32(Sm2))
    close(Sm, ref_Sm, atol=2e-4 * float(ref_Sm.abs().max()))
    close(P, ref_P, atol=2e-4 * float(ref_P.abs().max()))
    # the plain form is unchanged by the template split
    Sm0, P0 = K.bil_reduce_project(f32(Y), f32(x), f32(B), sp_d)
    r0 = CK.bil_reduce_project(Y, x, B, sp_c)
    close(Sm0, r0[0], atol=2e-4 * float(r0[0].abs().max())); close(P0, r0[1], atol=2e-4 * float(r0[1].abs().max()))
    # (b) gB and dSm accumulated in place
    dP = rnd(g, E, I, C)
    base_gB, base_dSm = rnd(g, E, S, I), rnd(g, E, S, C)
    rg, rd, _ = CK.bil_project_bwd(dP, ref_Sm, B, x, sp_c, want_dY=False, gB_accum=base_gB.clone(), dSm_accum=base_dSm.clone())
    run_gB, run_dSm = f32(base_gB), f32(base_dSm)
    gB, dSm, _ = K.bil_project_bwd(f32(dP), f32(ref_Sm), f32(B), f32(x), sp_d, want_dY=False, gB_accum=run_gB, dSm_accum=run_dSm)
    assert gB is run_gB and dSm is run_dSm
    close(gB, rg, atol=2e-4 * float(rg.abs().max())); close(dSm, rd, atol=2e-4 * float(rd.abs().max()))


def test_gather_mul():
    g = torch.Generator().manual_seed(8)
    x, m = rnd(g, 50, 128), rnd(g, 700, 128)
    idx = torch.randint(0, 50, (700,), generator=g, dtype=torch.int32)
    close(K.gather_mul(f32(x), idx.to(DEV), f32(m), 0.37), x[idx.long()] * m * 0.37, rtol=1e-6, atol=1e-6)


def test_twice_differentiable_geometry_kernels():
    """csrc/geometry2.hip: distance and angle value / first adjoint / tangent kernels (the tangent pass differentiates
    the first adjoint with dual numbers) against float64 autograd of the reference formulas, including a collinear
    triplet (the max(|u x v|, 1e-9) clamp: zero gradient through y)."""
    g = torch.Generator().manual_seed(21)
    n = 40
    R = torch.rand(n, 3, generator=g, dtype=torch.float64) * 4.0
    R[3] = R[1] + 2.0 * (R[2] - R[1])                       # atoms 1, 2, 3 on a line
    R32 = R.float().double()
    tR = rnd(g, n, 3)
    E, T = 500, 1500
    idx = torch.stack([torch.randperm(n, generator=g)[:2] for _ in range(E)]).int()
    ec, ea = idx[:, 0].contiguous(), idx[:, 1].contiguous()
    tri = torch.stack([torch.randperm(n, generator=g)[:3] for _ in range(T)]).int()
    tri[0] = torch.tensor([2, 1, 3])                        # collinear: c = 2, a = 1, b = 3
    tc, ta, tb = (tri[:, i].contiguous() for i in range(3))
    d = lambda t: t.to(DEV)                                  # noqa: E731
    # distances
    gD = rnd(g, E)
    close(K.dist_fwd(f32(R), d(ec), d(ea)), CK.dist_fwd(R32, ec, ea), rtol=1e-6, atol=1e-6)
    close(K.dist_bwd(f32(gD), f32(R), d(ec), d(ea)), CK.dist_bwd(gD, R32, ec, ea), rtol=1e-5, atol=1e-5)
    Dd, H = K.dist_jvp(f32(R), f32(tR), f32(gD), d(ec), d(ea))
    rDd, rH = CK.dist_jvp(R32, tR, gD, ec, ea)
    close(Dd, rDd, rtol=1e-5, atol=1e-5)
    close(H, rH, rtol=1e-4, atol=1e-4 * float(rH.abs().max()))
    assert K.dist_jvp(f32(R), f32(tR), None, d(ec), d(ea), want_H=False)[1] is None
    # angles
    gth = rnd(g, T)
    close(K.angle_fwd(f32(R), d(tc), d(ta), d(tb)), CK.angle_fwd(R32, tc, ta, tb), rtol=1e-5, atol=2e-6)
    Gc, Gb = K.angle_bwd(f32(gth), f32(R), d(tc), d(ta), d(tb))
    rGc, rGb = CK.angle_bwd(gth, R32, tc, ta, tb)
    # derivatives of nearly straight / nearly folded angles divide by sin(theta): compared elementwise on the
    # well-conditioned triplets (sin(theta) >= 0.3, most of them), finite everywhere — incl. the collinear triplet 0,
    # whose gradient through y is cut by the clamp
    th = CK.angle_fwd(R32, tc, ta, tb)
    well = torch.sin(th) >= 0.3
    well[0] = False
    assert int(well.sum()) > 1000
    wd = well.to(DEV)
    sc = float(rGc[well].abs().max())
    close(Gc[wd], rGc[well], rtol=1e-3, atol=2e-4 * sc)
    close(Gb[wd], rGb[well], rtol=1e-3, atol=2e-4 * sc)
    assert torch.isfinite(Gc).all() and torch.isfinite(Gb).all()
    thd, Hc, Hb = K.angle_jvp(f32(R), f32(tR), f32(gth), d(tc), d(ta), d(tb))
    rthd, rHc, rHb = CK.angle_jvp(R32, tR, gth, tc, ta, tb)
    close(thd[wd], rthd[well], rtol=1e-3, atol=2e-4 * float(rthd[well].abs().max()))
    hs = float(rHc[well].abs().max())
    close(Hc[wd], rHc[well], rtol=2e-3, atol=5e-4 * hs)
    close(Hb[wd], rHb[well], rtol=2e-3, atol=5e-4 * hs)
    assert torch.isfinite(Hc).all() and torch.isfinite(Hb).all() and torch.isfinite(thd).all()
    t2, none_c, none_b = K.angle_jvp(f32(R), f32(tR), None, d(tc), d(ta), d(tb), want_H=False)
    assert none_c is None and none_b is None and torch.equal(t2, thd)


@pytest.mark.parametrize("masked", [False, True])
def test_force_loss_and_its_cotangents_in_one_launch(masked):
    """gn_force_loss_f32 against the ATen composite of trainer.py:330-343 ((1 - rho) MAE(E) + rho mean L2(F)) and its
    autograd gradients; a row with F = Ft (zero norm) gets a zero gradient, masked rows (padded training step) no share."""
    g = torch.Generator().manual_seed(7 + masked)
    n_mol, A, rho = 37, 1501, 0.999
    E, Et = (torch.randn(n_mol, 1, generator=g, dtype=torch.float64) for _ in range(2))
    F, Ft = (torch.randn(A, 3, generator=g, dtype=torch.float64) for _ in range(2))
    Ft[5] = F[5].float().double()
    F[5] = Ft[5]
    mask = (torch.rand(A, generator=g) > 0.3).double() if masked else None
    n_at = float(mask.sum()) if masked else float(A)
    E64, F64 = E.clone().requires_grad_(True), F.clone().requires_grad_(True)
    m = mask if masked else torch.ones(A, dtype=torch.float64)
    d = torch.where(m[:, None] > 0, F64 - Ft, torch.ones_like(F64))
    ref = (1 - rho) * (E64 - Et).abs().sum() / n_mol + rho * (torch.norm(d, p=2, dim=1) * m).sum() / n_at
    ref.backward()
    inv = torch.tensor(1.0 / n_at, device=DEV, dtype=torch.float32)
    loss, gE, gF = K.force_loss(f32(E), f32(Et), f32(F), f32(Ft), (1 - rho) / n_mol, rho if masked else rho / n_at,
                                mask=f32(mask) if masked else None, w_f_dev=inv if masked else None)
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    close(gE, E64.grad, atol=1e-9)
    close(gF, F64.grad, atol=2e-7 * float(F64.grad.abs().max()))
    assert float(gF[5].abs().max()) == 0.0
    loss2, gE2, gF2 = K.force_loss(f32(E), f32(Et), f32(F), f32(Ft), (1 - rho) / n_mol, rho if masked else rho / n_at,
                                   mask=f32(mask) if masked else None, w_f_dev=inv if masked else None)
    assert torch.equal(loss, loss2) and torch.equal(gF, gF2)


@pytest.mark.parametrize("S,R,N", [(7, 6, 16), (7, 6, 9), (3, 4, 16)])
def test_cbf_project_forward_and_adjoint_against_the_composite(S, R, N):
    """gn_cbf_project_{fwd,bwd}_f32 against basis_layers.py:119-131 + a bias-free Dense in float64: rad[ie] * Y_l0 -> (I, S R)
    -> @ W^T, and the autograd gradients w.r.t. rad and y; rows sorted by interaction edge, some edges without rows; twice:
    bitwise."""
    g = torch.Generator().manual_seed(S * 100 + N)
    E = 301
    counts = torch.randint(0, 9, (E,), generator=g)
    counts[17] = 0
    counts[E - 1] = 0
    ie = torch.repeat_interleave(torch.arange(E), counts)
    I = int(ie.shape[0])
    seg = torch.zeros(E + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    rad, y, W = rnd(g, E, S, R).requires_grad_(True), rnd(g, I, S).requires_grad_(True), rnd(g, N, S * R)
    ref = (rad[ie] * y[:, :, None]).reshape(I, S * R) @ W.t()
    go = rnd(g, I, N)
    ref.backward(go)
    out = K.cbf_project_fwd(f32(rad.detach()), ie.to(torch.int32).to(DEV), f32(y.detach()), f32(W))
    close(out, ref.detach(), atol=2e-5)
    g_rad, g_y = K.cbf_project_bwd(f32(go), f32(rad.detach()), seg.to(DEV), f32(y.detach()), f32(W))
    close(g_rad, rad.grad, atol=5e-5)
    close(g_y, y.grad, atol=5e-5)
    g_rad2, g_y2 = K.cbf_project_bwd(f32(go), f32(rad.detach()), seg.to(DEV), f32(y.detach()), f32(W))
    assert torch.equal(g_rad, g_rad2) and torch.equal(g_y, g_y2)
