"""Padding of a batch to fixed capacities (gemnet_pytorch_amd/padded.py): the fabricated index rows are a valid graph of
their own, and — on the CPU emulation of the launchers, float64 — the padded batch gives the real molecules the energies
and forces of the unpadded one."""
import numpy as np
import torch

import cpu_kernels
from gemnet_pytorch_amd.padded import dummy_positions, pad_indices
from test_model_cpu import build
from test_oracle_model import load_case


def _idx(inputs):
    return {k: inputs[k] for k in ("id_c", "id_a", "id_swap", "id_undir", "id3_reduce_ca", "id3_expand_ba")}


def test_pad_rows_form_a_valid_graph(golden_model2):
    cfg, params, inputs = load_case(golden_model2, "t2s")
    idx = _idx(inputs)
    A = int(inputs["Z"].shape[0])
    E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
    for e_cap, t_cap, G in ((E + 8, T + 2, 1), (E + 40, T + 300, 3), (E + 4, T, 2), (E, T, 1), (E + 6, T + 4, 2), (E + 2, T, 1)):
        o = pad_indices(idx, A, e_cap, t_cap, G)
        assert all(o[k].shape[0] == e_cap for k in ("id_c", "id_a", "id_swap", "id_undir"))
        assert o["id3_reduce_ca"].shape[0] == t_cap == o["id3_expand_ba"].shape[0]
        for k in idx:                                   # the real rows are untouched and come first
            assert torch.equal(o[k][:idx[k].shape[0]], idx[k].to(torch.int64))
        sw = o["id_swap"]
        assert torch.equal(sw[sw], torch.arange(e_cap))                      # involution
        assert torch.equal(o["id_c"][sw], o["id_a"]) and torch.equal(o["id_a"][sw], o["id_c"])
        assert torch.equal(o["id_undir"][sw], o["id_undir"])                 # both directions share the undirected id
        assert int(o["id_undir"].max()) == e_cap // 2 - 1 and torch.bincount(o["id_undir"]).eq(2).all()
        r, x = o["id3_reduce_ca"], o["id3_expand_ba"]
        assert bool((r[1:] >= r[:-1]).all())                                 # sorted by reduce edge
        assert torch.equal(o["id_a"][r], o["id_a"][x])                       # c -> a <- b: same target atom
        assert bool((o["id_c"][r] != o["id_c"][x]).all())                    # b != c: a proper angle
        pad_atoms = torch.cat([o["id_c"][E:], o["id_a"][E:]])
        assert pad_atoms.numel() == 0 or (int(pad_atoms.min()) >= A and int(pad_atoms.max()) < A + 3 * G)
        assert bool((r[T:] >= E).all()) and bool((x[T:] >= E).all())         # pad triplets live on pad edges only


def test_padded_batch_reproduces_the_unpadded_molecules(golden_model2):
    cfg, params, inputs = load_case(golden_model2, "t2s")
    A = int(inputs["Z"].shape[0])
    with cpu_kernels.emulate():
        model = build(cfg, params).eval()
        base = dict(inputs, R=inputs["R"].double())
        E0, F0 = model(base)
        G = 2
        idx = _idx(inputs)
        E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
        pad = pad_indices(idx, A, E + 24, T + 50, G)
        R = torch.cat([inputs["R"].double(), dummy_positions(G, inputs["R"].double(), offset=50.0)])
        n_mol = int(inputs["N"].shape[0])
        padded = dict(Z=torch.cat([inputs["Z"], torch.ones(3 * G, dtype=inputs["Z"].dtype)]), R=R,
                      N=torch.cat([inputs["N"], torch.tensor([3 * G])]),
                      batch_seg=torch.cat([inputs["batch_seg"], torch.full((3 * G,), n_mol, dtype=inputs["batch_seg"].dtype)]),
                      max_in_degree=64, **pad)
        E1, F1 = model(padded)
    assert E1.shape[0] == n_mol + 1 and F1.shape[0] == A + 3 * G
    np.testing.assert_allclose(E1[:n_mol].detach().numpy(), E0.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(F1[:A].detach().numpy(), F0.detach().numpy(), rtol=1e-11, atol=1e-12)
    assert torch.isfinite(E1).all() and torch.isfinite(F1).all()


def test_runner_fill_equals_pad_indices(golden_model2):
    """The runner writes real rows and precomputed pad patterns straight into its static buffers: same arrays as the
    reference construction `pad_indices`, for two batches of different sizes in turn (stale rows must not survive)."""
    from gemnet_pytorch_amd.padded import PaddedGraphRunner

    class _M:       # the runner only looks at these two attributes before a capture
        triplets_only, direct_forces = True, False
    cfg, params, inputs = load_case(golden_model2, "t2s")
    idx = _idx(inputs)
    E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
    runner = PaddedGraphRunner(_M(), inputs["Z"], inputs["N"], E + 42, T + 600, max_in_degree=64, n_groups=3)
    fewer = dict(idx, id3_reduce_ca=idx["id3_reduce_ca"][:T - 6], id3_expand_ba=idx["id3_expand_ba"][:T - 6])
    for batch in (idx, fewer, idx):
        runner._fill(inputs["R"].float(), batch, Z=inputs["Z"])
        ref = pad_indices(batch, runner.A, runner.e_cap, runner.t_cap, runner.G)
        got = runner.padded_inputs()
        for k, v in ref.items():
            assert torch.equal(got[k], v), k
        assert torch.equal(got["R"][:runner.A], inputs["R"].float())
    with __import__("pytest").raises(ValueError):
        runner._fill(inputs["R"].float(), dict(idx, id_c=torch.cat([idx["id_c"]] * 3), id_a=torch.cat([idx["id_a"]] * 3),
                                               id_swap=torch.cat([idx["id_swap"]] * 3), id_undir=torch.cat([idx["id_undir"]] * 3)))


def test_triplet_atom_csr_by_expansion_equals_the_stable_sort(golden_model2):
    """GraphPlan builds the CSR of the triplets' atoms c and a (adjoint of the angle kernel) by expanding the atoms' edge
    lists into triplet ranges instead of sorting T keys: same permutation and offsets as the stable sort, also on a padded
    batch (pad edges are not grouped by target atom)."""
    from gemnet_pytorch_amd.graph import GraphPlan, RowIndex
    for tag in ("t2s", "t4s"):
        cfg, params, inputs = load_case(golden_model2, tag)
        A = int(inputs["Z"].shape[0])
        idx = _idx(inputs)
        E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
        pad = pad_indices(idx, A, E + 26, T + 90, 3)
        n_mol = int(inputs["N"].shape[0])
        padded = dict(Z=torch.cat([inputs["Z"], torch.ones(9, dtype=inputs["Z"].dtype)]),
                      N=torch.cat([inputs["N"], torch.tensor([9])]),
                      batch_seg=torch.cat([inputs["batch_seg"], torch.full((9,), n_mol, dtype=inputs["batch_seg"].dtype)]), **pad)
        for batch in (dict(inputs), padded):
            plan = GraphPlan(batch, True)
            for name in ("t_c", "t_a"):
                ri = getattr(plan, name)
                perm, seg = ri.csr
                perm0, seg0 = RowIndex(ri.idx64, ri.n_rows).csr
                assert torch.equal(perm, perm0) and torch.equal(seg, seg0), (tag, name)


def test_padded_training_step_has_the_gradients_of_the_unpadded_batch(golden_model2):
    """PaddedTrainStep (training/ddp.py) on the CPU emulation, float64: loss and every parameter gradient of the padded
    batch equal those of the plain TrainStep on the unpadded batch — the dummy molecule is cut off before the loss."""
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
    g = golden_model2
    cfg, params, inputs = load_case(g, "t2s")
    idx = _idx(inputs)
    E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
    Et = torch.tensor(g["t2s.Et"]).double()[:, None]
    Ft = torch.tensor(g["t2s.Ft"]).double()
    grads = []
    with cpu_kernels.emulate():
        for padded in (False, True):
            model = build(cfg, params).train()
            if padded:
                ts = PaddedTrainStep(model, inputs["Z"], inputs["N"], E + 30, T + 120, max_in_degree=64, n_groups=2)
                for k in ("R",):
                    ts.inputs[k] = ts.inputs[k].double()
                ts.pad.inputs["R"] = ts.inputs["R"]
                ts.inputs["R"][ts.pad.A:] = dummy_positions(2, ts.inputs["R"], offset=40.0)
                ts.targets = {k: v.double() for k, v in ts.targets.items()}
                loss = ts.step(inputs["R"].double(), idx, Et, Ft, Z=inputs["Z"], step_optimizer=False)
            else:
                ts = TrainStep(model)
                loss = ts(dict(inputs, R=inputs["R"].double()), {"E": Et, "F": Ft}, step_optimizer=False)
            grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = grads
    np.testing.assert_allclose(l1, l0, rtol=1e-12)
    assert g0.keys() == g1.keys()
    for n in g0:
        np.testing.assert_allclose(g1[n].numpy(), g0[n].numpy(), rtol=1e-9, atol=1e-12 * float(g0[n].abs().max() + 1e-30), err_msg=n)


def test_variable_atom_counts_through_one_set_of_buffers(golden_model, golden_model2):
    """`a_cap`: batches whose molecules differ in size share the runner's static buffers — a 32-atom molecule, then a
    12-atom one (the 20 atoms in between turn into isolated filler atoms of the dummy molecule again), then the first
    again: energies and forces of the real atoms equal the unpadded runs (CPU emulation, float64)."""
    from gemnet_pytorch_amd.padded import PaddedGraphRunner
    cfg, params, small = load_case(golden_model, "t1")
    _, _, big = load_case(golden_model, "t4")
    with cpu_kernels.emulate():
        model = build(cfg, params).eval()
        ref = {}
        for name, inp in (("small", small), ("big", big)):
            E, F = model(dict(inp, R=inp["R"].double()))
            ref[name] = (E.detach().clone(), F.detach().clone())
        Eb, Tb = int(big["id_c"].shape[0]), int(big["id3_reduce_ca"].shape[0])
        runner = PaddedGraphRunner(model, big["Z"], big["N"], Eb + 40, Tb + 200, max_in_degree=64, n_groups=6, a_cap=40)
        runner.inputs["R"] = runner.inputs["R"].double()
        runner._R_fill = runner._R_fill.double()
        for name, inp in (("big", big), ("small", small), ("big", big)):
            runner._fill(inp["R"].double(), _idx(inp), Z=inp["Z"], N=inp["N"])
            padded = dict(runner.padded_inputs(), max_in_degree=64)
            A = int(inp["Z"].shape[0])
            assert int(padded["N"].sum()) == runner.A_tot and int(padded["N"][-1]) == runner.A_tot - A
            assert torch.equal(padded["batch_seg"][:A], inp["batch_seg"]) and bool((padded["batch_seg"][A:] == 1).all())
            E, F = model(padded)
            E0, F0 = ref[name]
            np.testing.assert_allclose(E[:1].detach().numpy(), E0.numpy(), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(F[:A].detach().numpy(), F0.numpy(), rtol=1e-10, atol=1e-12)
            assert torch.isfinite(F).all() and float(F[A:runner.a_cap].abs().max()) == 0.0     # isolated filler atoms
    import pytest
    with pytest.raises(ValueError):
        runner._fill(small["R"].double(), _idx(small))                  # a different atom count needs Z and N


def test_padded_training_step_with_changing_molecule_sizes(golden_model):
    """PaddedTrainStep with `a_cap`: a 32-atom batch, a 12-atom batch, the first again through one set of buffers — loss
    and parameter gradients of each equal the plain TrainStep on the unpadded batch (CPU emulation, float64)."""
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
    cfg, params, small = load_case(golden_model, "t1")
    _, _, big = load_case(golden_model, "t4")
    gen = torch.Generator().manual_seed(0)
    tgt = {id(b): (torch.randn(1, 1, generator=gen).double(), torch.randn(int(b["Z"].shape[0]), 3, generator=gen).double())
           for b in (small, big)}
    Eb, Tb = int(big["id_c"].shape[0]), int(big["id3_reduce_ca"].shape[0])
    with cpu_kernels.emulate():
        model_p = build(cfg, params).train()
        pts = PaddedTrainStep(model_p, big["Z"], big["N"], Eb + 40, Tb + 200, max_in_degree=64, n_groups=6, a_cap=40)
        pts.inputs["R"] = pts.pad.inputs["R"] = pts.pad.inputs["R"].double()
        pts.pad._R_fill = pts.pad._R_fill.double()
        for batch in (big, small, big):
            Et, Ft = tgt[id(batch)]
            lp = pts.step(batch["R"].double(), _idx(batch), Et, Ft, Z=batch["Z"], N=batch["N"], step_optimizer=False)
            gp = {n: p.grad.detach().clone() for n, p in model_p.named_parameters() if p.grad is not None}
            model_e = build(cfg, params).train()
            le = TrainStep(model_e)(dict(batch, R=batch["R"].double()), {"E": Et, "F": Ft}, step_optimizer=False)
            ge = {n: p.grad.detach().clone() for n, p in model_e.named_parameters() if p.grad is not None}
            np.testing.assert_allclose(float(lp), float(le), rtol=1e-12)
            assert gp.keys() == ge.keys()
            for n in ge:
                np.testing.assert_allclose(gp[n].numpy(), ge[n].numpy(), rtol=1e-9,
                                           atol=1e-12 * float(ge[n].abs().max() + 1e-30), err_msg=n)


# ---------------------------------------------------------------------------------------------- quadruplet models (round 5)
Q_KEYS = ("id4_int_a", "id4_int_b", "id4_reduce_intm_ca", "id4_reduce_intm_ab", "id4_expand_intm_db", "id4_expand_intm_ab",
          "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd")


def _idx_q(inputs):
    return {**_idx(inputs), **{k: inputs[k] for k in Q_KEYS}}


def _sizes_q(idx):
    return (int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0]), int(idx["id4_int_a"].shape[0]),
            int(idx["id4_expand_intm_db"].shape[0]), int(idx["id4_reduce_ca"].shape[0]))


def test_quadruplet_pad_rows_form_a_valid_graph(golden_model2):
    """GemNet-Q: groups of four dummy atoms, pad edges in units of six, pad interaction edges, pad intermediate triplets
    (sorted by interaction edge) and pad quadruplets (sorted by reduce edge) — every pad row carries valid indices, four
    distinct atoms per quadruplet, and the identities of the real arrays (data_container.py:331-397)."""
    cfg, params, inputs = load_case(golden_model2, "q2s")
    idx = _idx_q(inputs)
    A = int(inputs["Z"].shape[0])
    E, T, Eint, I, Q = _sizes_q(idx)
    for (de, dt, di, dm, dq), G in (((12, 6, 4, 5, 9), 1), ((48, 300, 9, 40, 700), 3), ((6, 0, 1, 1, 1), 2), ((0, 0, 0, 0, 0), 1),
                                    ((20, 2, 2, 7, 3), 2)):
        caps = (Eint + di, I + dm, Q + dq)
        o = pad_indices(idx, A, E + de, T + dt, G, quad_caps=caps)
        for k in idx:                                   # the real rows are untouched and come first
            assert torch.equal(o[k][:idx[k].shape[0]], idx[k].to(torch.int64)), k
        sw = o["id_swap"]
        assert torch.equal(sw[sw], torch.arange(E + de))
        assert torch.equal(o["id_c"][sw], o["id_a"]) and torch.equal(o["id_a"][sw], o["id_c"])
        r, x = o["id3_reduce_ca"], o["id3_expand_ba"]
        assert bool((r[1:] >= r[:-1]).all()) and torch.equal(o["id_a"][r], o["id_a"][x]) and bool((o["id_c"][r] != o["id_c"][x]).all())
        assert all(o[k].shape[0] == caps[0] for k in ("id4_int_a", "id4_int_b"))
        assert all(o[k].shape[0] == caps[1] for k in Q_KEYS[2:6]) and all(o[k].shape[0] == caps[2] for k in Q_KEYS[6:])
        ab = o["id4_expand_intm_ab"]
        assert bool((ab[1:] >= ab[:-1]).all())                                                    # sorted by interaction edge
        rq = o["id4_reduce_ca"]
        assert bool((rq[1:] >= rq[:-1]).all())                                                    # sorted by reduce edge
        assert torch.equal(o["id4_expand_db"], o["id4_expand_intm_db"][o["id4_expand_abd"]])      # the identities of Appendix B
        assert torch.equal(o["id4_reduce_ca"][:Q], o["id4_reduce_intm_ca"][o["id4_reduce_cab"]][:Q])
        # the four atoms of every quadruplet c -> a - b <- d: all distinct (real and pad)
        c, a = o["id_c"][o["id4_reduce_ca"]], o["id_a"][o["id4_reduce_ca"]]
        d, b = o["id_c"][o["id4_expand_db"]], o["id_a"][o["id4_expand_db"]]
        for u, v in ((c, a), (c, b), (c, d), (a, b), (a, d), (b, d)):
            assert bool((u != v).all())
        # the a - b <- d atoms of every intermediate triplet: distinct, the interaction edge's source is b
        ia, ib = o["id4_int_a"][ab], o["id4_int_b"][ab]
        dd, bb = o["id_c"][o["id4_expand_intm_db"]], o["id_a"][o["id4_expand_intm_db"]]
        assert bool((ia != ib).all()) and bool((dd != ib).all()) and bool((dd != ia)[I:].all())   # (real ones may have d = a)
        assert torch.equal(bb[:I], ib[:I])
        for k in Q_KEYS:                                 # pad rows live on pad edges / pad atoms / pad intermediate triplets
            n = idx[k].shape[0]
            lo = {"id4_int_a": A, "id4_int_b": A, "id4_reduce_cab": I, "id4_expand_abd": I, "id4_reduce_intm_ab": Eint,
                  "id4_expand_intm_ab": Eint}.get(k, E)
            assert o[k][n:].numel() == 0 or int(o[k][n:].min()) >= lo, k
        assert int(torch.cat([o["id4_int_a"], o["id4_int_b"]]).max()) < A + 4 * G


def _padded_quad_inputs(inputs, G, extra):
    A = int(inputs["Z"].shape[0])
    idx = _idx_q(inputs)
    E, T, Eint, I, Q = _sizes_q(idx)
    de, dt, di, dm, dq = extra
    pad = pad_indices(idx, A, E + de, T + dt, G, quad_caps=(Eint + di, I + dm, Q + dq))
    n_mol = int(inputs["N"].shape[0])
    R = torch.cat([inputs["R"].double(), dummy_positions(G, inputs["R"].double(), offset=50.0, quad=True)])
    return dict(Z=torch.cat([inputs["Z"], torch.ones(4 * G, dtype=inputs["Z"].dtype)]), R=R,
                N=torch.cat([inputs["N"], torch.tensor([4 * G])]),
                batch_seg=torch.cat([inputs["batch_seg"], torch.full((4 * G,), n_mol, dtype=inputs["batch_seg"].dtype)]),
                max_in_degree=64, **pad), A, n_mol


def test_padded_quadruplet_batch_reproduces_the_unpadded_molecules(golden_model2):
    cfg, params, inputs = load_case(golden_model2, "q2s")
    with cpu_kernels.emulate():
        model = build(cfg, params).eval()
        E0, F0 = model(dict(inputs, R=inputs["R"].double()))
        padded, A, n_mol = _padded_quad_inputs(inputs, 2, (36, 50, 6, 30, 400))
        E1, F1 = model(padded)
    assert E1.shape[0] == n_mol + 1 and F1.shape[0] == A + 8
    np.testing.assert_allclose(E1[:n_mol].detach().numpy(), E0.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(F1[:A].detach().numpy(), F0.detach().numpy(), rtol=1e-11, atol=1e-12)
    assert torch.isfinite(E1).all() and torch.isfinite(F1).all()


def test_padded_quadruplet_training_step_gives_the_unpadded_gradients(golden_model2):
    """Force training of a padded GemNet-Q batch (fused angle-form twins): the dummy molecule's energy and forces are cut off
    before the loss, so every parameter gradient equals the unpadded step's."""
    from oracle import gemnet_oracle as GO
    g = golden_model2
    cfg, params, inputs = load_case(g, "q2s")
    Et, Ft = torch.tensor(g["q2s.Et"]).double()[:, None], torch.tensor(g["q2s.Ft"]).double()
    grads = []
    with cpu_kernels.emulate():
        for pad in (False, True):
            model = build(cfg, params).train()
            if pad:
                batch, A, n_mol = _padded_quad_inputs(inputs, 2, (24, 20, 4, 12, 150))
            else:
                batch, A, n_mol = dict(inputs, R=inputs["R"].double()), int(inputs["Z"].shape[0]), int(inputs["N"].shape[0])
            E, F = model(batch)
            GO.training_loss(E[:n_mol, :1], F[:A], Et, Ft).backward()
            grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        scale = max(float(grads[0][n].abs().max()), 1e-30)
        assert float((grads[0][n] - grads[1][n]).abs().max()) <= 1e-9 * scale, n


def test_quadruplet_runner_fill_equals_pad_indices(golden_model2):
    from gemnet_pytorch_amd.padded import PaddedGraphRunner

    class _M:
        triplets_only, direct_forces = False, False
    cfg, params, inputs = load_case(golden_model2, "q2s")
    idx = _idx_q(inputs)
    sizes = _sizes_q(idx)
    assert PaddedGraphRunner.sizes_of(idx) == sizes
    e_cap, t_cap, caps = PaddedGraphRunner.suggest_capacities([sizes], margin=0.2)
    runner = PaddedGraphRunner(_M(), inputs["Z"], inputs["N"], e_cap, t_cap, max_in_degree=64, n_groups=3, quad_caps=caps)
    assert runner.fits(sizes) and not runner.fits((sizes[0],) + (t_cap + 2,) + sizes[2:])
    Q = sizes[4]
    fewer = dict(idx, **{k: idx[k][:Q - 40] for k in Q_KEYS[6:]})
    for batch in (idx, fewer, idx):
        runner._fill(inputs["R"].float(), batch, Z=inputs["Z"])
        ref = pad_indices(batch, runner.A, runner.e_cap, runner.t_cap, runner.G, quad_caps=runner.quad_caps)
        got = runner.padded_inputs()
        for k, v in ref.items():
            assert torch.equal(got[k].to(torch.int64), v), k


def test_dummy_molecule_geometry_scales_with_the_bond_length():
    """The dummy molecule's bonds sit at 0.9 x the embedding cutoff (padded.dummy_bond: the radial envelopes have almost closed
    there, so the pad rows' messages stay ~1e-3 of a real row's — docs/HISTORY.md section 14, "A silent NaN"); angles and the dihedral
    stay 90 degrees at any bond length, groups never overlap."""
    from gemnet_pytorch_amd.padded import dummy_bond

    class _M:
        class cbf_basis3:
            cutoff = 5.0
    assert abs(dummy_bond(_M()) - 4.5) < 1e-12 and dummy_bond(object()) == 1.0
    like = torch.zeros(1, dtype=torch.float64)
    for bond in (1.0, 4.5):
        for quad in (False, True):
            P = dummy_positions(3, like, offset=100.0, quad=quad, bond=bond).reshape(3, 4 if quad else 3, 3)
            a, b, c = P[:, 0], P[:, 1], P[:, 2]
            assert torch.allclose((b - a).norm(dim=1), torch.full((3,), bond, dtype=torch.float64))
            assert torch.allclose((c - a).norm(dim=1), torch.full((3,), bond, dtype=torch.float64))
            assert float(((b - a) * (c - a)).sum(dim=1).abs().max()) < 1e-9          # angle c-a-b = 90 degrees
            if quad:
                d = P[:, 3]
                assert torch.allclose((d - b).norm(dim=1), torch.full((3,), bond, dtype=torch.float64))
                assert float(((a - b) * (d - b)).sum(dim=1).abs().max()) < 1e-9      # angle a-b-d
                n1, n2 = torch.cross(c - a, b - a, dim=1), torch.cross(a - b, d - b, dim=1)
                assert float((n1 * n2).sum(dim=1).abs().max()) < 1e-9                 # dihedral c-a-b-d
            flat = P.reshape(-1, 3)
            dist = (flat[:, None] - flat[None]).norm(dim=2) + torch.eye(flat.shape[0], dtype=torch.float64) * 1e9
            assert float(dist.min()) >= 0.99 * bond
