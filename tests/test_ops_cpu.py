"""Host logic above the C ABI, exercised on CPU with the launchers emulated (tests/cpu_kernels.py):
autograd closure of gemnet_pytorch_amd.ops (gradcheck + gradgradcheck in float64) and the index
plans of gemnet_pytorch_amd.graph."""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck, gradgradcheck

from gemnet_pytorch_amd import ops
from gemnet_pytorch_amd.graph import RowIndex, SegmentPlan
import cpu_kernels


@pytest.fixture(autouse=True)
def _emulated():
    with cpu_kernels.emulate():
        yield


def rnd(*shape):
    return torch.randn(*shape, dtype=torch.float64, requires_grad=True)


def test_rowindex_csr():
    idx = torch.tensor([3, 1, 3, 0, 1, 3])
    ri = RowIndex(idx, 5)
    perm, seg = ri.csr
    assert seg.tolist() == [0, 1, 3, 3, 6, 6]
    assert idx[perm.long()].tolist() == [0, 1, 1, 3, 3, 3]
    assert perm.tolist() == [3, 1, 4, 0, 2, 5]  # stable
    rs = RowIndex(torch.tensor([0, 0, 2, 2, 2]), 4, is_sorted=True)
    assert rs.csr[0] is None and rs.csr[1].tolist() == [0, 2, 2, 5, 5]


def test_gather_segsum_adjoint():
    idx = torch.tensor([3, 1, 3, 0, 1, 3])
    ri = RowIndex(idx, 5)
    x = rnd(5, 4)
    assert gradcheck(lambda t: ops.gather_rows(t, ri), (x,))
    assert gradgradcheck(lambda t: ops.gather_rows(t, ri), (x,))
    y = rnd(6, 4)
    out = ops.segsum_rows(y, ri)
    ref = torch.zeros(5, 4, dtype=torch.float64).index_add(0, idx, y)
    assert torch.allclose(out, ref)
    assert gradcheck(lambda t: ops.segsum_rows(t, ri), (y,))


def test_swap_involution():
    swap = torch.tensor([2, 3, 0, 1])
    ri = RowIndex(swap, 4)
    ri.inverse = ri
    x = rnd(4, 3)
    assert gradcheck(lambda t: ops.gather_rows(t, ri), (x,))


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("tb", [False, True])
def test_mm(ta, tb):
    A = rnd(*((5, 7) if ta else (7, 5)))
    Bm = rnd(*((5, 3) if tb else (3, 5)))
    a = A.t() if ta else A
    b = Bm if tb else Bm.t()
    assert torch.allclose(ops.mm(A, Bm, ta, tb), a @ b)
    assert gradcheck(lambda x, y: ops.mm(x, y, ta, tb), (A, Bm))
    assert gradgradcheck(lambda x, y: ops.mm(x, y, ta, tb), (A, Bm))


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("tb", [False, True])
def test_bmm(ta, tb):
    A = rnd(*((4, 3, 2) if ta else (4, 2, 3)))
    Bm = rnd(*((4, 5, 3) if tb else (4, 3, 5)))
    assert gradcheck(lambda x, y: ops.bmm(x, y, ta, tb), (A, Bm))
    assert gradgradcheck(lambda x, y: ops.bmm(x, y, ta, tb), (A, Bm))


def test_param_grads_switch():
    A, W = rnd(4, 3), rnd(2, 3)
    with ops.param_grads(False):
        gA, gW = torch.autograd.grad(ops.linear(A, W).sum(), (A, W), allow_unused=True)
    assert gA is not None and gW is None


def test_ssilu_chain():
    x = rnd(6)
    assert torch.allclose(ops.ssilu(x), torch.nn.functional.silu(x) / 0.6)
    assert gradcheck(ops.ssilu, (x,))
    assert gradgradcheck(ops.ssilu, (x,))


def _segplan():
    red = torch.tensor([0, 0, 0, 2, 2, 3])
    exp = torch.tensor([1, 3, 2, 0, 1, 1])
    return SegmentPlan(red, exp, 4, 4)


def test_bilinear_closure():
    sp = _segplan()
    Y, x = rnd(6, 3), rnd(4, 5)
    f = lambda a, b: ops.bil_reduce(a, b, sp)
    ref = torch.zeros(4, 3, 5, dtype=torch.float64).index_add(
        0, sp.reduce.idx64, Y[:, :, None] * x[sp.expand.idx64][:, None, :])
    assert torch.allclose(f(Y, x), ref)
    assert gradcheck(f, (Y, x))
    assert gradgradcheck(f, (Y, x))


def test_basis_ops_first_and_second_order():
    d = (torch.rand(5, dtype=torch.float64) * 3.5 + 1.0).requires_grad_(True)
    freq = (torch.arange(1, 4, dtype=torch.float64) * np.pi).requires_grad_(True)
    f = lambda a, b: ops.bessel_rbf(a, b, 5.0, 5)
    assert gradcheck(f, (d, freq))
    assert gradgradcheck(lambda a: ops.bessel_rbf(a, freq.detach(), 5.0, 5), (d,))
    th = (torch.rand(5, dtype=torch.float64) * 3.0 + 0.05).requires_grad_(True)
    ph = (torch.rand(5, dtype=torch.float64) * 3.0 + 0.05).requires_grad_(True)
    assert gradcheck(lambda t: ops.ylm0(t, 4), (th,))
    assert gradgradcheck(lambda t: ops.ylm0(t, 4), (th,))
    assert gradcheck(lambda t, p: ops.ylm(t, p, 3), (th, ph))
    assert gradgradcheck(lambda t, p: ops.ylm(t, p, 3), (th, ph))
    from oracle import basis_oracle as B
    z = torch.tensor(B.jn_zeros(3, 2))
    nrm = torch.tensor(B.sph_bessel_normalizer(3, 2))
    assert gradcheck(lambda t: ops.sph_radial(t, z, nrm, 5.0, 5), (d,))
    assert gradgradcheck(lambda t: ops.sph_radial(t, z, nrm, 5.0, 5), (d,))


def test_triplet_groups_by_target_atom():
    """SegmentPlan.groups (consumed by gn_bil_reduce_t_grouped_f32): every triplet's reduce and expand edge sit in the
    group of their common target atom, and rposT / grp_kseg address them."""
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training import data_container as DC
    ds = make_dataset(3, 20, config=2)
    idx = DC.build_indices(ds["R"], ds["N"], 5.0, 10.0, True)
    inputs = {k: torch.as_tensor(v) for k, v in idx.items()}
    inputs["Z"] = torch.as_tensor(ds["Z"]).long()
    inputs["N"] = torch.as_tensor(ds["N"]).long()
    inputs["batch_seg"] = torch.repeat_interleave(torch.arange(len(ds["N"])), inputs["N"])
    plan = GraphPlan(inputs, True)
    rows, off, kseg, rposT, max_rows = plan.trip.groups
    permT, segT = plan.trip.expand.csr
    id_a, red, exp = inputs["id_a"], inputs["id3_reduce_ca"], inputs["id3_expand_ba"]
    assert torch.equal(id_a[red], id_a[exp])
    assert sorted(rows.tolist()) == list(range(plan.n_edges))
    assert max_rows == int(torch.bincount(id_a, minlength=plan.n_atoms).max())
    seen = 0
    for g in range(plan.n_atoms):
        for i in range(int(off[g]), int(off[g + 1])):
            j = int(rows[i])
            assert int(id_a[j]) == g
            k0, k1 = kseg[i].tolist()
            assert (k0, k1) == (int(segT[j]), int(segT[j + 1]))
            for k in range(k0, k1):
                t = int(permT[k])
                assert int(exp[t]) == j and int(rows[int(off[g]) + int(rposT[k])]) == int(red[t])
                seen += 1
    assert seen == plan.trip.size


def test_fuse_program_preserves_semantics():
    """K.fuse_program moves SCALE ops into the producing GEMM / LOAD epilogue: the fused program must compute the
    same tensors (the adjoint of a Dense + two residual layers with a parked skip gradient and a residual output)."""
    import gemnet_pytorch_amd.kernels as K
    g = torch.Generator().manual_seed(0)
    M, W = 37, 64

    def mk(*s):
        return torch.randn(*s, generator=g, dtype=torch.float64)
    gin, Ws = mk(M, W), [mk(W, W) / 8 for _ in range(5)]
    zs = [mk(M, W) for _ in range(5)]

    def build():
        outs = [torch.zeros(M, W, dtype=torch.float64) for _ in range(3)]
        p = K.ChainProgram(M)
        p.load(0, gin)
        cur, oth = 0, 1
        for k in (1, 0):
            if k == 0:
                p.scale(2, cur, 0.6, width=W)
            p.scale(cur, cur, 0.7 if k else 0.42, width=W)
            p.scale(oth, cur, 1.0, Z=zs[2 * k + 1])
            p.gemm(Ws[2 * k + 1], a_slot=oth, y_slot=oth)
            p.scale(oth, oth, 1.0, Z=zs[2 * k])
            p.gemm(Ws[2 * k], a_slot=oth, y_slot=cur, res=cur, beta=1.0)
        p.scale(cur, cur, 0.9, out=outs[0], width=W)
        p.scale(oth, cur, 1.0, Z=zs[4], out=outs[1], width=W)
        p.gemm(Ws[4], a_slot=oth, y_slot=-1, out=outs[2], res=2, beta=1.0)
        return p, outs
    p0, o0 = build()
    cpu_kernels.chain(p0)
    p1, o1 = build()
    f = K.fuse_program(p1)
    kinds = [o["kind"] for o in f.ops]
    assert kinds.count("scale") == 1 and kinds.count("gemm") == 5, kinds       # only the park survives
    cpu_kernels.chain(f)
    for a, b in zip(o0, o1):
        assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(a.abs().max()))


def _shared_consumer_graph(mark):
    """x -> three fused consumers (Dense, Dense with an activation, the radial aggregation) -> scalar."""
    torch.manual_seed(3)
    E_, A = 40, 7
    x0 = torch.randn(E_, 128, dtype=torch.float64, requires_grad=True)
    rbf = torch.randn(E_, 16, dtype=torch.float64)
    W1, W2 = torch.randn(32, 128, dtype=torch.float64) / 11, torch.randn(64, 128, dtype=torch.float64) / 11
    Wr = torch.randn(128, 16, dtype=torch.float64) / 4
    ri = RowIndex(torch.randint(0, A, (E_,)), A)
    x = x0 * 1.0          # a non-leaf, like every activation of the model
    if mark:
        ops.accumulate_gradient(x)
    ys = [ops.dense(x, W1), ops.dense(x, W2, True), ops.rbf_aggregate(x, rbf, Wr, ri, 0.5)]
    return x0, ys


def test_accumulate_gradient_matches_autograd_sum():
    """ops.accumulate_gradient: the consumers' in-kernel running sum equals autograd's own accumulation; a backward
    pass that prunes a registered consumer fails loudly instead of returning a partial gradient."""
    with cpu_kernels.emulate(), ops.fused_first_order(True), ops.param_grads(False):
        grads = {}
        for mark in (False, True):
            x0, ys = _shared_consumer_graph(mark)
            assert all(y is not None for y in ys)
            loss = sum((y * y).sum() for y in ys)
            grads[mark], = torch.autograd.grad(loss, x0)
        torch.testing.assert_close(grads[True], grads[False], rtol=1e-12, atol=1e-12)
        # two passes over one graph (several energy targets): the running state is per pass
        x0, ys = _shared_consumer_graph(True)
        loss = sum((y * y).sum() for y in ys)
        g1, = torch.autograd.grad(loss, x0, retain_graph=True)
        g2, = torch.autograd.grad(loss, x0)
        torch.testing.assert_close(g1, grads[False], rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(g2, grads[False], rtol=1e-12, atol=1e-12)
        x0, ys = _shared_consumer_graph(True)
        with pytest.raises(RuntimeError, match="fused consumers did not run"):
            torch.autograd.grad((ys[0] * ys[0]).sum(), x0)


def test_accumulate_gradient_is_per_forward():
    """A long-lived tensor marked again by every forward (a leaf fed to one block repeatedly): each forward gets its own
    running sum, consumers of an earlier forward are not counted (bench.py extra.interaction_block_fwd_bwd)."""
    with cpu_kernels.emulate(), ops.fused_first_order(True), ops.param_grads(False):
        torch.manual_seed(5)
        x = (torch.randn(30, 128, dtype=torch.float64)).requires_grad_(True) * 1.0
        x.retain_grad()
        W1, W2 = torch.randn(32, 128, dtype=torch.float64) / 11, torch.randn(64, 128, dtype=torch.float64) / 11
        ref = None
        for _ in range(3):
            ops.accumulate_gradient(x)
            loss = (ops.dense(x, W1) ** 2).sum() + (ops.dense(x, W2, True) ** 2).sum()
            g, = torch.autograd.grad(loss, x)
            ref = g if ref is None else ref
            torch.testing.assert_close(g, ref, rtol=1e-12, atol=1e-12)


def test_accumulate_gradient_recovers_after_an_aborted_pass():
    """A backward pass that dies between two consumers leaves a half-counted sink behind; the next pass (another graph
    task) starts from scratch instead of handing out the stale partial sum."""
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            if Boom.armed:
                raise ValueError("boom")
            return g
    with cpu_kernels.emulate(), ops.fused_first_order(True), ops.param_grads(False):
        x0, ys = _shared_consumer_graph(True)
        loss = (ys[0] ** 2).sum() + (Boom.apply(ys[1]) ** 2).sum() + (ys[2] ** 2).sum()
        Boom.armed = True
        with pytest.raises(ValueError, match="boom"):
            torch.autograd.grad(loss, x0, retain_graph=True)
        Boom.armed = False
        g, = torch.autograd.grad(loss, x0)
        x1, ys1 = _shared_consumer_graph(False)
        ref, = torch.autograd.grad(sum((y ** 2).sum() for y in ys1), x1)
        torch.testing.assert_close(g, ref, rtol=1e-12, atol=1e-12)


def test_plan_groupings_expanded_csr_and_expand_positions():
    """graph.expanded_csr (the triplets grouped by the atoms of their reduce edge WITHOUT sorting T keys; its device form is
    gn_expanded_csr_i32) equals the stable sort of the item keys, and SegmentPlan.expand_pos is the inverse of the expand CSR's
    permutation."""
    from gemnet_pytorch_amd.graph import RowIndex, SegmentPlan, expanded_csr
    g = torch.Generator().manual_seed(3)
    E, n_rows = 500, 41
    row_of_edge = torch.randint(0, n_rows, (E,), generator=g)
    cnt = torch.randint(0, 9, (E,), generator=g)
    so = torch.zeros(E + 1, dtype=torch.int64)
    so[1:] = torch.cumsum(cnt, 0)
    T = int(so[-1])
    perm, seg = expanded_csr(RowIndex(row_of_edge, n_rows), so, T)
    item_key = torch.repeat_interleave(row_of_edge, cnt)
    want = torch.argsort(item_key, stable=True)
    assert torch.equal(perm.long(), want)
    assert torch.equal(seg.long(), torch.searchsorted(item_key[want].contiguous(), torch.arange(n_rows + 1)))
    reduce_idx = torch.repeat_interleave(torch.arange(E), cnt)
    expand_idx = torch.randint(0, 77, (T,), generator=g)
    sp = SegmentPlan(reduce_idx, expand_idx, E, 77)
    p, _ = sp.expand.csr
    pos = sp.expand_pos
    assert torch.equal(pos[p.long()].long(), torch.arange(T)) and torch.equal(p[pos.long()].long(), torch.arange(T))
