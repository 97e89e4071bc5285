"""Batches whose array sizes change from call to call, replayed from ONE captured hipGraph.

The MD loop of the reference rebuilds the graph every step (ase_calculator.py:155-158) and the data provider hands out a
new batch every step (data_provider.py:159-165): the number of edges and triplets differs by a few per cent from call to
call, so a captured forward+force cannot be replayed and the eager path is bound by ~240 Python-side launches (8-12 ms for
2.7 ms of kernels).  Here every batch is PADDED to fixed capacities with a dummy molecule and the one graph captured for
those capacities is replayed:

  * `A_pad = 3 G` dummy atoms in G groups (a, b, c) with a valid geometry of their own (bonds of 0.9 x the cutoff, where
    the envelope has almost closed: `dummy_positions`; 90 degrees), far from
    every real atom and assigned to an extra molecule whose energy and forces are dropped;
  * pad edges come in quads  b->a, a->b, c->a, a->c  cycling over the groups (id_swap = the neighbour in the pair,
    id_undir continues the numbering), pad triplets are the two orderings (c->a, b->a), (b->a, c->a) of the quads' forward
    edges, as many as needed, sorted by reduce edge like the real ones — duplicates of an edge or of a triplet are just
    more rows of the dummy molecule's sums;
  * the index plan (CSR sorts, triplet groups) is built INSIDE the captured region from the static index buffers, so each
    replay rebuilds it on the device for the indices of that batch; the one host read-back it needed (largest in-degree)
    is replaced by the bound the caller knows (`max_in_degree`: atoms per molecule - 1).

Molecules are independent (block-diagonal index arrays), every kernel of the path treats a row on its own and sums a
segment in index order: the real molecules' energies and forces are those of the unpadded batch, bit for bit
(tests/test_gpu_padded.py).  With an atom capacity (`a_cap`) the molecules may differ in size from call to call as well: the
atoms between the batch and the capacity are isolated filler atoms of the dummy molecule.

Quadruplet models (GemNet-Q; round 5) — the model of the reference's only recorded run, the MD of ase_example.ipynb through
ase_calculator.py:148-170 — pad the same way with groups of FOUR dummy atoms (a, b, c, d: 1 A bonds a-b, a-c, b-d, all angles
and the dihedral c-a-b-d 90 degrees), pad edges cycling b->a, c->a, d->b (+ reverses) in UNITS of six, and three more
capacities: interaction edges (pad: b->a / a->b of a group), intermediate triplets (pad: (edge d->b, interaction edge b->a),
sorted by interaction edge like the real ones) and quadruplets (pad: reduce edge c->a of a unit, sorted; expand row = a pad
intermediate triplet).  A pad row only needs valid indices and a non-degenerate geometry — the real molecules never see it.

Callers: `runtime.DynamicForceField` (the MD loop), `training.ddp.PaddedTrainStep` and
`Trainer.enable_padded_graph` (the training step in the same form), `bench.py` (`extra.dynamic_shape`,
`extra.train_step_dynamic`).
"""
import torch


PAD_EDGE_KEYS = ("id_c", "id_a", "id_swap", "id_undir")
PAD_TRIP_KEYS = ("id3_reduce_ca", "id3_expand_ba")
PAD_INT_KEYS = ("id4_int_a", "id4_int_b")
PAD_INTM_KEYS = ("id4_reduce_intm_ca", "id4_reduce_intm_ab", "id4_expand_intm_db", "id4_expand_intm_ab")
PAD_QUAD_KEYS = ("id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd")


def _pad_edges(k, n_atoms, n_groups, quad=False):
    """Pad edge number k (0-based behind the real edges) -> (source atom, target atom).  Pairs (edge, reverse); the pairs
    alternate b->a | c->a and cycle over the groups of three dummy atoms.  `quad`: groups of FOUR atoms (a, b, c, d) and
    three kinds of pairs b->a | c->a | d->b — a UNIT of six pad edges per visit of a group."""
    pair, rev = k // 2, k % 2
    G = max(n_groups, 1)
    if not quad:
        typ, grp = pair % 2, (pair // 2) % G
        a = n_atoms + 3 * grp
        other = a + 1 + typ                           # b or c
        return torch.where(rev == 0, other, a), torch.where(rev == 0, a, other)
    typ, grp = pair % 3, (pair // 3) % G
    a = n_atoms + 4 * grp
    src = torch.where(typ == 2, a + 3, a + 1 + typ)   # b, c or d
    dst = torch.where(typ == 2, a + 1, a)             # a, a or b
    return torch.where(rev == 0, src, dst), torch.where(rev == 0, dst, src)


def _pad_triplets(j, E, ep, tp, quad=False):
    """Pad triplet number j of tp -> (reduce edge, expand edge): the forward edges (even k) of the complete quads, spread
    evenly and ALREADY sorted by reduce edge (f grows with j); the expand edge is the other forward edge of the quad —
    same target atom a, the other source.  `quad`: the forward edges b->a, c->a (offsets 0 and 2) of the complete units."""
    if not quad:
        n_fwd = 2 * (ep // 4)
        f = (j * n_fwd) // max(tp, 1)
        return E + 2 * f, E + 2 * (f ^ 1)
    s = (j * (2 * (ep // 6))) // max(tp, 1)
    u, w = s // 2, s % 2
    return E + 6 * u + 2 * w, E + 6 * u + 2 * (1 - w)


def _pad_int_edges(k, n_atoms, n_groups):
    """Pad interaction edge k -> (id4_int_b = source, id4_int_a = target): pairs b->a, a->b cycling over the groups."""
    p, rev = k // 2, k % 2
    a = n_atoms + 4 * (p % max(n_groups, 1))
    return torch.where(rev == 0, a + 1, a), torch.where(rev == 0, a, a + 1)


def _pad_intm(i, E, Eint, ep, eintp, ip, n_groups):
    """Pad intermediate triplet i of ip -> (embedding edge c->a, embedding edge d->b, interaction edge b->a): interaction edges
    spread evenly and sorted (id4_expand_intm_ab is a sorted list), the embedding edges those of a unit of the same group
    when one exists."""
    n_ab, n_units, G = (eintp + 1) // 2, max(ep // 6, 1), max(n_groups, 1)
    p = (i * n_ab) // max(ip, 1)
    g = p % G
    u = torch.where(g < n_units, g, p % n_units)
    return E + 6 * u + 2, E + 6 * u + 4, Eint + 2 * p


def _pad_quads(q, E, I, ep, ip, qp):
    """Pad quadruplet q of qp -> (reduce edge c->a, index of a pad intermediate triplet): reduce edges spread evenly over the
    units and sorted; the expand rows cycle over the pad intermediate triplets."""
    u = (q * max(ep // 6, 1)) // max(qp, 1)
    return E + 6 * u + 2, I + q % max(ip, 1)


def _check_quad_padding(ep, tp, eintp, ip, qp):
    if (tp or ip or qp) and ep < 6:
        raise ValueError("pad triplets / intermediate triplets / quadruplets need a complete unit of six pad edges")
    if ip and eintp < 1:
        raise ValueError("pad intermediate triplets need a pad interaction edge")
    if qp and ip < 1:
        raise ValueError("pad quadruplets need a pad intermediate triplet")


def pad_indices(idx, n_atoms, e_cap, t_cap, n_groups, dtype=torch.int64, quad_caps=None):
    """idx: the index dict of one batch (id_c, id_a, id_swap, id_undir, id3_reduce_ca, id3_expand_ba; any int dtype)
    -> dict of the same keys padded to (e_cap, t_cap) as described in the module docstring.  Pure index arithmetic
    (runs on any device, no host sync besides the shapes).  `quad_caps` = (eint_cap, i_cap, q_cap): a quadruplet batch (idx
    then also holds the eleven id4_* arrays) padded with groups of four dummy atoms."""
    quad = quad_caps is not None
    E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
    ep, tp = e_cap - E, t_cap - T
    if ep < 0 or tp < 0:
        raise ValueError(f"batch ({E} edges, {T} triplets) exceeds the capacities ({e_cap}, {t_cap})")
    if ep % 2 or E % 2 or tp % 2:
        raise ValueError("edge and triplet padding must be even (edges and triplets come in pairs)")
    if tp and ep < (6 if quad else 4):
        raise ValueError("pad triplets need a complete quad of pad edges (a unit of six for quadruplet batches)")
    dev = idx["id_c"].device
    k = torch.arange(ep, device=dev, dtype=torch.int64)
    src, dst = _pad_edges(k, n_atoms, n_groups, quad)
    out = {
        "id_c": torch.cat([idx["id_c"].to(dtype), src.to(dtype)]),
        "id_a": torch.cat([idx["id_a"].to(dtype), dst.to(dtype)]),
        "id_swap": torch.cat([idx["id_swap"].to(dtype), (E + (k ^ 1)).to(dtype)]),
        "id_undir": torch.cat([idx["id_undir"].to(dtype), (E // 2 + k // 2).to(dtype)]),
    }
    red, exp = _pad_triplets(torch.arange(tp, device=dev, dtype=torch.int64), E, ep, tp, quad)
    out["id3_reduce_ca"] = torch.cat([idx["id3_reduce_ca"].to(dtype), red.to(dtype)])
    out["id3_expand_ba"] = torch.cat([idx["id3_expand_ba"].to(dtype), exp.to(dtype)])
    if quad:
        eint_cap, i_cap, q_cap = quad_caps
        Eint, I, Q = int(idx["id4_int_a"].shape[0]), int(idx["id4_expand_intm_db"].shape[0]), int(idx["id4_reduce_ca"].shape[0])
        if int(idx["id4_reduce_intm_ca"].shape[0]) != I:
            raise ValueError("the two intermediate-triplet lists differ in length")
        eintp, ip, qp = eint_cap - Eint, i_cap - I, q_cap - Q
        if min(eintp, ip, qp) < 0:
            raise ValueError(f"batch ({Eint} interaction edges, {I} intermediate triplets, {Q} quadruplets) exceeds the "
                             f"capacities {tuple(quad_caps)}")
        _check_quad_padding(ep, tp, eintp, ip, qp)
        ar = lambda n: torch.arange(n, device=dev, dtype=torch.int64)      # noqa: E731
        ib, ia = _pad_int_edges(ar(eintp), n_atoms, n_groups)
        out["id4_int_b"] = torch.cat([idx["id4_int_b"].to(dtype), ib.to(dtype)])
        out["id4_int_a"] = torch.cat([idx["id4_int_a"].to(dtype), ia.to(dtype)])
        ca, db, ab = _pad_intm(ar(ip), E, Eint, ep, eintp, ip, n_groups)
        for key, v in (("id4_reduce_intm_ca", ca), ("id4_reduce_intm_ab", ab), ("id4_expand_intm_db", db),
                       ("id4_expand_intm_ab", ab)):
            out[key] = torch.cat([idx[key].to(dtype), v.to(dtype)])
        rq, im = _pad_quads(ar(qp), E, I, ep, ip, qp)
        out["id4_reduce_ca"] = torch.cat([idx["id4_reduce_ca"].to(dtype), rq.to(dtype)])
        out["id4_expand_abd"] = torch.cat([idx["id4_expand_abd"].to(dtype), im.to(dtype)])
        out["id4_reduce_cab"] = torch.cat([idx["id4_reduce_cab"].to(dtype), im.to(dtype)])
        out["id4_expand_db"] = torch.cat([idx["id4_expand_db"].to(dtype), db[im - I].to(dtype) if qp else db[:0].to(dtype)])
    return out


def dummy_positions(n_groups, like, offset=1.0e3, quad=False, bond=1.0):
    """(3 G, 3) positions: group g = atoms a, b, c with |ab| = |ac| = `bond` and a right angle, 10 bond lengths between
    groups, `offset` away from the origin (real molecules of a batch sit near it).  `quad`: (4 G, 3) — a fourth atom d bonded
    to b, out of the plane: the angles c-a-b, a-b-d and the dihedral c-a-b-d are all 90 degrees.
    The runner uses bond = 0.9 x the embedding cutoff: the envelope of every radial basis is 0.04 there, so the pad rows'
    messages are ~1e-3 of a real row's.  With 1 A bonds the few pad edges that carry ALL pad triplets (24-45 identical
    ones each, summed coherently where a real edge sums ~18 different ones) reached 7e4 in the bilinear layer's output at the
    32 x 32 batch — beyond the fp16 planes of the default Dense arithmetic: harmless in inference (the dummy molecule's rows
    are dropped) but in force TRAINING their zero cotangents times inf made every weight gradient NaN (found by the range
    flag of round 5; round 4's padded training step had it silently)."""
    g = torch.arange(n_groups, device=like.device, dtype=like.dtype)
    base = torch.stack([offset + 10.0 * bond * g, torch.full_like(g, offset), torch.full_like(g, offset)], dim=1)
    rows = [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]] + ([[1.0, 0.0, 1.0]] if quad else [])
    d = bond * torch.tensor(rows, device=like.device, dtype=like.dtype)
    return (base[:, None, :] + d[None, :, :]).reshape(-1, 3)


def dummy_bond(model):
    """Bond length of the dummy molecule for `model`: 0.9 x its embedding cutoff (1 A when the model does not tell)."""
    try:
        return 0.9 * float(model.cbf_basis3.cutoff)
    except (AttributeError, TypeError, ValueError):
        return 1.0


class PaddedGraphRunner:
    """runner = PaddedGraphRunner(model, Z, N, e_cap, t_cap);  E, F = runner(R, idx[, Z])   for every batch of this layout.

    Z (A,), N (n_mol,): atomic numbers and molecule sizes of a first batch; e_cap / t_cap: capacities (rounded up to even
    here); `max_in_degree`: bound of the incoming edges of one atom (default max(N) - 1).
    `a_cap` (optional): capacity of ATOMS — batches whose molecules differ in size from call to call (the same number of
    molecules; a loader's batches): pass N (and Z) with every call; the atoms between the batch and `a_cap` are isolated
    filler atoms of the dummy molecule.  Without it the layout N is fixed and only Z / R / the index arrays change.
    A batch that does not fit raises `ValueError` — size the capacities from the first batches with a margin
    (`suggest_capacities`)."""

    def __init__(self, model, Z, N, e_cap, t_cap, max_in_degree=None, n_groups=None, a_cap=None, index_dtype=torch.int32,
                 quad_caps=None):
        """`quad_caps` = (eint_cap, i_cap, q_cap): capacities of interaction edges, intermediate triplets and quadruplets —
        required for quadruplet models (`suggest_capacities` sizes all five from a few batches)."""
        self.quad = not model.triplets_only
        if self.quad and quad_caps is None:
            raise ValueError("quadruplet models need quad_caps = (eint_cap, i_cap, q_cap)")
        self.GS = 4 if self.quad else 3              # atoms per dummy group
        self.model = model
        dev = Z.device
        self.A, self.n_mol = int(Z.shape[0]), int(N.shape[0])          # A: atoms of the CURRENT batch
        self.variable_atoms = a_cap is not None
        self.a_cap = int(a_cap) if a_cap is not None else self.A
        if self.a_cap < self.A:
            raise ValueError("a_cap below the first batch")
        self.e_cap, self.t_cap = (int(e_cap) + 1) // 2 * 2, (int(t_cap) + 1) // 2 * 2
        self.deg = int(max_in_degree) if max_in_degree is not None else int(N.max().item()) - 1
        # every group's atom a takes pad edges of both kinds: at most 2 * ceil(pad quads / G) incoming
        self.G = int(n_groups) if n_groups is not None else max(1, -(-self.e_cap // (16 * max(self.deg, 2))))
        Ap = self.GS * self.G
        self.A_tot = self.a_cap + Ap
        i64 = torch.int64
        n_fill = self.a_cap - self.A
        self.inputs = {
            "Z": torch.cat([Z.to(i64), torch.ones(n_fill + Ap, dtype=i64, device=dev)]),
            "N": torch.cat([N.to(i64), torch.tensor([n_fill + Ap], dtype=i64, device=dev)]),
            "R": torch.zeros(self.A_tot, 3, device=dev, dtype=self._float_dtype(model)),
            "batch_seg": torch.cat([torch.repeat_interleave(torch.arange(self.n_mol, device=dev), N.to(i64)),
                                    torch.full((n_fill + Ap,), self.n_mol, dtype=i64, device=dev)]),
            "max_in_degree": None,
        }
        # filler atoms: isolated, 10 A apart on a line of their own; the dummy groups behind them (fixed positions: only
        # the rows of the current batch's atoms are rewritten per call)
        fill = torch.arange(self.a_cap, device=dev, dtype=self.inputs["R"].dtype)
        self._R_fill = torch.stack([-1.0e3 - 10.0 * fill, torch.full_like(fill, -1.0e3), torch.full_like(fill, -1.0e3)], dim=1)
        self.inputs["R"][:self.a_cap] = self._R_fill
        self.inputs["R"][self.a_cap:] = dummy_positions(self.G, self.inputs["R"], quad=self.quad, bond=dummy_bond(model))
        # The edge / triplet arrays live in `index_dtype` (int32: what the kernels read — the plan built inside the replayed
        # graph then has no conversion launches, and the device index builder hands its int32 arrays over as they are;
        # batches given as int64 are converted by the copy into the buffers).  Values stay far below 2^31 (checked).
        self.quad_caps = tuple(int(c) for c in quad_caps) if self.quad else None
        if max((self.e_cap, self.t_cap, self.A_tot) + (self.quad_caps or ())) >= 2 ** 31:
            index_dtype = i64
        self.index_dtype = index_dtype
        for k in PAD_EDGE_KEYS:
            self.inputs[k] = torch.zeros(self.e_cap, dtype=index_dtype, device=dev)
        for k in PAD_TRIP_KEYS:
            self.inputs[k] = torch.zeros(self.t_cap, dtype=index_dtype, device=dev)
        if self.quad:
            eint_cap, i_cap, q_cap = self.quad_caps
            for k in PAD_INT_KEYS:
                self.inputs[k] = torch.zeros(eint_cap, dtype=index_dtype, device=dev)
            for k in PAD_INTM_KEYS:
                self.inputs[k] = torch.zeros(i_cap, dtype=index_dtype, device=dev)
            for k in PAD_QUAD_KEYS:
                self.inputs[k] = torch.zeros(q_cap, dtype=index_dtype, device=dev)
            self._arange_q = torch.arange(max(eint_cap, i_cap, q_cap), device=dev, dtype=i64)
        k = torch.arange(self.e_cap, device=dev, dtype=i64)
        self._pat_src, self._pat_dst = (t.to(index_dtype) for t in _pad_edges(k, self.a_cap, self.G, self.quad))
        self._pat_swap, self._pat_pair = (k ^ 1).to(index_dtype), (k // 2).to(index_dtype)
        self._arange_t = torch.arange(self.t_cap, device=dev, dtype=i64)
        self.graph = None
        self.out = None
        self.flag = None
        self.builder = None                 # attach_builder: the index build runs inside the replayed graph
        self._filled = False
        if dev.type == "cuda":
            from .runtime import RangeFlag
            self.flag = RangeFlag(dev)      # device-side range check of every replay (fp16-plane arithmetic), polled lazily

    @staticmethod
    def _float_dtype(model):
        """fp32 on the device; the float64 CPU emulation of the tests keeps its own precision."""
        try:
            return next(model.parameters()).dtype
        except (AttributeError, StopIteration, TypeError):
            return torch.float32

    def padded_inputs(self):
        """The static input buffers as one padded batch (a view of the runner's state: for tests)."""
        return {k: v for k, v in self.inputs.items() if v is not None}

    @staticmethod
    def suggest_capacities(sizes, margin=0.06):
        """sizes: [(E, T), ...] of a few batches -> (e_cap, t_cap).  Triplets get `margin` head room, edges a little more:
        every pad triplet needs a pad edge to reduce into, and a pad edge with hundreds of triplets is a long tail for the
        one wave that owns it (a real edge has ~18).  Quadruplet batches: sizes [(E, T, Eint, I, Q), ...] ->
        (e_cap, t_cap, (eint_cap, i_cap, q_cap))."""
        e = max(s[0] for s in sizes)
        t = max(s[1] for s in sizes)
        if len(sizes[0]) == 2:
            return int(e * (1 + 1.5 * margin)) // 4 * 4 + 8, int(t * (1 + margin)) // 2 * 2 + 2
        ei, im, q = (max(s[k] for s in sizes) for k in (2, 3, 4))
        return (int(e * (1 + 1.5 * margin)) // 12 * 12 + 24, int(t * (1 + margin)) // 2 * 2 + 2,
                (int(ei * (1 + 1.5 * margin)) + 4, int(im * (1 + margin)) + 4, int(q * (1 + margin)) + 4))

    @staticmethod
    def sizes_of(idx):
        """(E, T[, Eint, I, Q]) of an index dict."""
        s = (int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0]))
        if "id4_reduce_ca" in idx:
            s += (int(idx["id4_int_a"].shape[0]), int(idx["id4_expand_intm_db"].shape[0]), int(idx["id4_reduce_ca"].shape[0]))
        return s

    def fits(self, sizes):
        """Does a batch of these sizes (sizes_of) fit the capacities with valid padding?"""
        E, T = sizes[:2]
        ep, tp = self.e_cap - E, self.t_cap - T
        ok = ep >= 0 and tp >= 0 and not (tp and ep < (6 if self.quad else 4)) and self.pad_in_degree(ep) <= self.pad_degree_bound()
        if ok and self.quad:
            eintp, ip, qp = (c - n for c, n in zip(self.quad_caps, sizes[2:5]))
            ok = min(eintp, ip, qp) >= 0 and not ((ip or qp) and ep < 6) and not (ip and eintp < 1) and not (qp and ip < 1)
        return ok

    def _fill(self, R, idx, Z=None, N=None):
        """Write one batch into the static buffers: the real rows as they are, the pad rows from the patterns computed
        once for the whole capacity (an edge's pattern only depends on its distance from the first pad edge)."""
        A = int(R.shape[0])
        if A != self.A:
            if not self.variable_atoms or N is None or Z is None:
                raise ValueError("the number of atoms changed: build the runner with a_cap and pass Z and N with every call")
            if A > self.a_cap or int(N.shape[0]) != self.n_mol:
                raise ValueError(f"batch of {A} atoms / {int(N.shape[0])} molecules exceeds a_cap = {self.a_cap} or changes the "
                                 f"number of molecules ({self.n_mol})")
        if N is not None:
            buf = self.inputs
            buf["N"][:self.n_mol].copy_(N)
            buf["N"][self.n_mol:].fill_(self.A_tot - A)
            buf["batch_seg"][:A].copy_(torch.repeat_interleave(torch.arange(self.n_mol, device=N.device), N.to(torch.int64),
                                                                output_size=A))
            buf["batch_seg"][A:].fill_(self.n_mol)
            if A < self.A:      # atoms that were real in the previous batch become filler again
                buf["R"][A:self.A].copy_(self._R_fill[A:self.A])
                buf["Z"][A:self.A].fill_(1)
            self.A = A
        if Z is not None:
            self.inputs["Z"][:self.A].copy_(Z)
        E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])
        ep, tp = self.e_cap - E, self.t_cap - T
        if ep < 0 or tp < 0:
            raise ValueError(f"batch ({E} edges, {T} triplets) exceeds the capacities ({self.e_cap}, {self.t_cap})")
        if ep % 2 or tp % 2 or (tp and ep < (6 if self.quad else 4)):
            raise ValueError("edge and triplet padding must be even, and pad triplets need a complete quad of pad edges "
                             "(a unit of six for quadruplet models)")
        if self.pad_in_degree(ep) > self.pad_degree_bound():
            raise ValueError(f"{ep} pad edges over {self.G} dummy groups exceed the in-degree bound {self.pad_degree_bound()}")
        buf = self.inputs
        for k in PAD_EDGE_KEYS + PAD_TRIP_KEYS:
            buf[k][:idx[k].shape[0]].copy_(idx[k])
        if ep:
            buf["id_c"][E:].copy_(self._pat_src[:ep])
            buf["id_a"][E:].copy_(self._pat_dst[:ep])
            torch.add(self._pat_swap[:ep], E, out=buf["id_swap"][E:])
            torch.add(self._pat_pair[:ep], E // 2, out=buf["id_undir"][E:])
        if tp:
            red, exp = _pad_triplets(self._arange_t[:tp], E, ep, tp, self.quad)
            buf["id3_reduce_ca"][T:].copy_(red)
            buf["id3_expand_ba"][T:].copy_(exp)
        if self.quad:
            self._fill_quad(idx, E, ep, tp)
        buf["R"][:self.A].copy_(R)
        self._filled = True

    def _fill_quad(self, idx, E, ep, tp):
        """Interaction edges, intermediate triplets and quadruplets of one batch + their pad rows (module docstring)."""
        buf = self.inputs
        Eint, I, Q = int(idx["id4_int_a"].shape[0]), int(idx["id4_expand_intm_db"].shape[0]), int(idx["id4_reduce_ca"].shape[0])
        if int(idx["id4_reduce_intm_ca"].shape[0]) != I:
            raise ValueError("the two intermediate-triplet lists differ in length")
        eint_cap, i_cap, q_cap = self.quad_caps
        eintp, ip, qp = eint_cap - Eint, i_cap - I, q_cap - Q
        if min(eintp, ip, qp) < 0:
            raise ValueError(f"batch ({Eint} interaction edges, {I} intermediate triplets, {Q} quadruplets) exceeds the "
                             f"capacities {self.quad_caps}")
        _check_quad_padding(ep, tp, eintp, ip, qp)
        for k in PAD_INT_KEYS + PAD_INTM_KEYS + PAD_QUAD_KEYS:
            buf[k][:idx[k].shape[0]].copy_(idx[k])
        ar = self._arange_q
        if eintp:
            ib, ia = _pad_int_edges(ar[:eintp], self.a_cap, self.G)
            buf["id4_int_b"][Eint:].copy_(ib)
            buf["id4_int_a"][Eint:].copy_(ia)
        if ip:
            ca, db, ab = _pad_intm(ar[:ip], E, Eint, ep, eintp, ip, self.G)
            buf["id4_reduce_intm_ca"][I:].copy_(ca)
            buf["id4_expand_intm_db"][I:].copy_(db)
            buf["id4_reduce_intm_ab"][I:].copy_(ab)
            buf["id4_expand_intm_ab"][I:].copy_(ab)
        if qp:
            rq, im = _pad_quads(ar[:qp], E, I, ep, ip, qp)
            buf["id4_reduce_ca"][Q:].copy_(rq)
            buf["id4_expand_abd"][Q:].copy_(im)
            buf["id4_reduce_cab"][Q:].copy_(im)
            buf["id4_expand_db"][Q:].copy_(db[im - I])

    def pad_degree_bound(self):
        return max(self.deg, 2)

    def pad_in_degree(self, ep):
        """Largest in-degree of a dummy atom with `ep` pad edges.  Triplet layout: both kinds of pairs (b->a, c->a) land on a
        group's atom a: the pairs of the group.  Quadruplet layout: a unit of six pad edges puts two on a (b->a, c->a) and two
        on b (a->b, d->b): twice the units of the group."""
        if self.quad:
            return 2 * -(-(-(-ep // 6)) // self.G)
        return -(-(ep // 2) // self.G)

    def __call__(self, R, idx, Z=None, N=None):
        """R (A, 3) float32 on the device, idx: the index dict of this batch, Z: its atomic numbers when they change from
        batch to batch, N: its molecule sizes when those change too (runner built with `a_cap`) -> (E (n_mol, targets),
        F (A, [targets,] 3)); the results live in the graph's static output buffers until the next call."""
        if self.flag is not None and self.flag.tripped():
            # an EARLIER replay returned non-finite energies / forces (seen without synchronising: one or two calls late;
            # a caller that waits for its results anyway checks `flag.tripped()` itself and calls `recover()` — md.py)
            self.recover()
        self._fill(R, idx, Z, N)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        E, F = self.out
        return E.detach()[:self.n_mol], F.detach()[:self.A]

    # ---- the index build INSIDE the replayed graph (triplets-only models, fixed molecule layout) --------------------------
    def attach_builder(self, builder):
        """From the next capture on the graph starts with the index build itself (index_device.DeviceGraphBuilder's layout,
        gn_index_gpu_padded_t: counts stay on the device, the arrays are committed into the static buffers with their pad
        rows by one kernel): a step is  positions in -> one replay -> (E, F) out  with no read-back and no padding launches
        on the host — the MD step of ase_calculator.py:148-170 as ONE graph.  The buffers must hold a valid batch already
        (one `_fill`): a step that does not fit the capacities keeps the previous arrays, poisons its outputs with NaN and
        reports through `index_error()`."""
        if self.variable_atoms:
            raise NotImplementedError("the in-graph index build needs a fixed molecule layout (no a_cap)")
        if builder.triplets_only == self.quad or builder.A != self.A or self.index_dtype != torch.int32:
            raise ValueError("builder and runner describe different systems")
        if not self._filled:
            raise RuntimeError("attach_builder: fill the buffers with a first batch (runner._fill / runner(R, idx)) before")
        dev = self.inputs["R"].device
        self.builder = builder
        n_stage = 4 * self.e_cap + 2 * (self.quad_caps[0] if self.quad else self.t_cap)
        # (zeros: a build whose interaction-edge count alone exceeds its capacity skips the edge kernel, and the count kernels
        # behind it must not index with uninitialised staging on the very first build)
        self._staging = torch.zeros(n_stage, dtype=torch.int32, device=dev)
        self._idx_state = torch.zeros(8, dtype=torch.int32, device=dev)
        self._idx_host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.graph = None
        self.out = None

    def _index_in_graph(self):
        from . import kernels as K
        if self.quad:
            K.index_padded_q(self.builder, self.inputs["R"][:self.A], (self.e_cap, self.t_cap) + self.quad_caps, self.a_cap,
                             self.G, self.pad_degree_bound(), self._staging, self.inputs, self._idx_state)
        else:
            K.index_padded_t(self.builder, self.inputs["R"][:self.A], self.e_cap, self.t_cap, self.a_cap, self.G,
                             self.pad_degree_bound(), self._staging, self.inputs, self._idx_state)

    def index_error(self):
        """Error bits (include/gemnet_hip.h, gn_index_gpu_padded_t) of the in-graph index builds COMPLETED so far — sticky;
        exact after anything that waited for the last replay."""
        return int(self._idx_host[0]) if self.builder is not None else 0

    def index_sizes(self):
        """(E, T[, Eint, I, Q]) of the last completed in-graph index build."""
        h = self._idx_host
        return (int(h[1]), int(h[2])) + ((int(h[4]), int(h[5]), int(h[6])) if self.quad else ())

    def reset_index_state(self):
        self._idx_state.zero_()
        self._idx_host.zero_()

    def run_positions(self, R, Z=None):
        """One step with the index build inside the graph: R (A, 3) float32 on the device -> (E, F) as `__call__`."""
        if self.builder is None:
            raise RuntimeError("run_positions needs attach_builder")
        if R.dtype != self.inputs["R"].dtype or tuple(R.shape) != (self.A, 3):
            # the neighbour list is built from the graph's own (fp32) position buffer: positions in another precision would
            # give other neighbours near the cutoff than the caller's own index build
            raise TypeError(f"run_positions: positions must be {self.inputs['R'].dtype} of shape ({self.A}, 3); got "
                            f"{R.dtype} {tuple(R.shape)}")
        if self.flag is not None and self.flag.tripped():
            self.recover()
        if self.index_error():
            raise ValueError(f"an earlier step did not fit the capacities ({self.e_cap}, {self.t_cap}, {self.quad_caps}): index error bits "
                             f"{self.index_error()}, sizes {self.index_sizes()} — its outputs were NaN; build a larger runner")
        if Z is not None:
            self.inputs["Z"][:self.A].copy_(Z)
        self.inputs["R"][:self.A].copy_(R)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        E, F = self.out
        return E.detach()[:self.n_mol], F.detach()[:self.A]

    def recover(self):
        """The range flag tripped: move the model off the fp16 planes (runtime.fall_back_to_bf16_planes: warns) and drop the
        graph — the next call captures anew in the bf16-plane arithmetic."""
        from .runtime import fall_back_to_bf16_planes
        torch.cuda.synchronize()
        self.flag.trips += 1
        self.flag.reset()
        if fall_back_to_bf16_planes(self.model, "a replayed padded batch (PaddedGraphRunner)", positions=self.inputs["R"]):
            self.graph = None
            self.out = None

    def build_and_run(self, builder, R, Z=None, positions_ready=False, N=None):
        """Index build (index_device.DeviceGraphBuilder) + padded replay with the build on a stream of its own: its size
        read-back then waits for the build kernels only, not for the previous replay still running on the calling stream,
        and the host can enqueue the next step while the GPU works.  `positions_ready`: R does not depend on work pending on
        the calling stream (a new batch of a data provider); the default (False: an MD step's new positions) orders the
        build behind the calling stream first."""
        main = torch.cuda.current_stream(R.device)
        if getattr(self, "_bstream", None) is None:
            self._bstream = torch.cuda.Stream(device=R.device)
        bs = self._bstream
        if not positions_ready:
            bs.wait_stream(main)
        with torch.cuda.stream(bs):
            idx = builder(R, dtype=self.index_dtype)
        main.wait_stream(bs)
        for t in idx.values():
            t.record_stream(main)
        return self(R, idx, Z, N)

    def _capture(self):
        inputs = dict(self.inputs, max_in_degree=self.pad_degree_bound(), _guard_rows=(self.n_mol, self.a_cap))
        if self.flag is not None:
            inputs["_range_flag"] = self.flag
        inputs.pop("_plan", None)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the default stream (autograd stream bookkeeping, lazy caches)
            for _ in range(2):
                if self.builder is not None:
                    self._index_in_graph()
                self.model(dict(inputs))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = dict(inputs)                  # a fresh dict: the plan is built inside the capture, from the static buffers
            if self.builder is not None:
                self._index_in_graph()
            if getattr(self, "check", False):   # happens-before check of the capture (hbcheck.py); recorder left in self.hb
                from . import hbcheck
                with hbcheck.record() as self.hb:
                    self.out = self.model(cap)
            else:
                self.out = self.model(cap)
            if self.builder is not None:
                from . import kernels as K
                for t in self.out:
                    K.index_poison(t, self._idx_state)
                self._idx_host.copy_(self._idx_state, non_blocking=True)
        self._cap_inputs = cap                  # keeps the plan's tensors (graph memory) referenced
        self.graph = g
