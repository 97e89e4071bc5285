"""Device-side index plans: int32 copies of the reference's index arrays plus the CSR groupings
(offsets + permutation) that make every gather's adjoint an atomic-free segmented sum.

The reference hands `GemNet.forward` a dict of int64 tensors (data_container.py:33-56,505-516)
whose triplet/quadruplet arrays are sorted by reduce edge (:324-328,:369-375); `Kidx3/Kidx4`
(position inside the segment) exist only to build zero-padded tensors and are not needed here.
SURVEY.md Appendix D lists the transpose groupings the backward passes need.
"""
import os

import torch


def _seg_offsets_sorted(sorted_idx: torch.Tensor, n_rows: int) -> torch.Tensor:
    """offsets[r] = first position with value >= r  (no host sync)."""
    bounds = torch.arange(n_rows + 1, device=sorted_idx.device, dtype=sorted_idx.dtype)
    return torch.searchsorted(sorted_idx.contiguous(), bounds).to(torch.int32)


def _late_wait(obj):
    """Structures that `GraphPlan.warm(late_stream=...)` builds on a stream of their own carry the event recorded behind
    their construction: a reader on another stream waits for it first (inside a capture: one more graph edge)."""
    ev = getattr(obj, "_late", None)
    if ev is not None:
        stream, event = ev
        cur = torch.cuda.current_stream()
        # (the event belongs to the capture that built the plan; outside of it there is nothing to order against)
        if cur != stream and torch.cuda.is_current_stream_capturing():
            cur.wait_event(event)


class RowIndex:
    """A row-index array `idx` (T,) into a matrix with `n_rows` rows.

    idx32        int32 copy used by the gather kernels
    perm/seg_off CSR by destination row for the adjoint (perm is None when idx is sorted)
    inverse      for permutations (id_swap): adjoint is a gather with the inverse permutation
    """

    def __init__(self, idx: torch.Tensor, n_rows: int, is_sorted: bool = False, inverse=None, csr_builder=None):
        self.idx64 = idx
        self.idx32 = idx.to(torch.int32).contiguous()
        self.n_rows = int(n_rows)
        self.size = int(idx.shape[0])
        self.is_sorted = is_sorted
        self.inverse = inverse
        self._csr = None
        self._csr_builder = csr_builder   # () -> (perm, seg_off): a cheaper construction than the sort, same result

    @property
    def csr(self):
        _late_wait(self)
        if self._csr is None:
            if self._csr_builder is not None:
                self._csr = self._csr_builder()
            elif self.idx32.is_cuda and _native_csr():
                # one native launch pair per grouping (csrc/csr.hip: radix sort over the significant key bits, int32 throughout)
                from . import kernels as _K
                if self.is_sorted:
                    self._csr = (None, _K.seg_offsets(self.idx32, self.n_rows))
                else:
                    self._csr = _K.csr_build(self.idx32, self.n_rows)
            elif self.is_sorted:
                self._csr = (None, _seg_offsets_sorted(self.idx64, self.n_rows))
            else:
                # 32-bit keys: half the bytes through every pass of the device sort (the plan is rebuilt per batch on the
                # dynamic-shape paths, inside the replayed graph of padded.py: four sorts of T keys were 0.5 ms of it)
                perm = torch.argsort(self.idx32, stable=True)
                seg = _seg_offsets_sorted(self.idx32[perm], self.n_rows)
                self._csr = (perm.to(torch.int32).contiguous(), seg)
        return self._csr


def expanded_csr(row_of_edge: RowIndex, seg_off_of_edge: torch.Tensor, n_items: int):
    """CSR (perm, seg_off) by row of the items t (triplets) whose row is `row_of_edge[edge(t)]`, when the items are sorted
    by their edge (`seg_off_of_edge`: item range of every edge): the rows' edge lists (the CSR of `row_of_edge`, a sort of
    E keys) expanded into item ranges — the same permutation as the stable sort of the T item keys (edges of a row in
    increasing order, the items of an edge in increasing order), without sorting T keys."""
    perm_e, seg_e = row_of_edge.csr
    if seg_e.is_cuda and _native_csr() and seg_e.dtype == torch.int32 and seg_off_of_edge.dtype == torch.int32:
        from . import kernels as _K
        if _K.USE_NATIVE_EXPANDED:
            return _K.expanded_csr(perm_e, seg_e, seg_off_of_edge, n_items)
    so = seg_off_of_edge.to(torch.int64)
    cnt = so[1:] - so[:-1]
    order = perm_e.to(torch.int64) if perm_e is not None else torch.arange(cnt.shape[0], device=cnt.device)
    c = cnt[order]
    off = torch.zeros(c.shape[0] + 1, dtype=torch.int64, device=c.device)
    torch.cumsum(c, 0, out=off[1:])
    j = torch.arange(n_items, device=c.device, dtype=torch.int64)
    k = torch.searchsorted(off[1:].contiguous(), j, right=True)      # slot of the edge that owns item j
    perm = so[order[k]] + (j - off[k])
    return perm.to(torch.int32).contiguous(), off[seg_e.to(torch.int64)].to(torch.int32).contiguous()


def _native_csr():
    from . import kernels as _K
    return _K.USE_NATIVE_CSR


def _atom_blocks_on():
    from . import kernels as _K
    return _K.USE_ATOM_BLOCKS


class SegmentPlan:
    """Triplets (or quadruplets) t with reduce row r(t) (sorted) and expand row g(t)."""

    def __init__(self, reduce_idx: torch.Tensor, expand_idx: torch.Tensor, n_reduce: int, n_expand: int):
        self.reduce = RowIndex(reduce_idx, n_reduce, is_sorted=True)
        self.expand = RowIndex(expand_idx, n_expand)
        self.size = int(reduce_idx.shape[0])
        self.n_reduce = int(n_reduce)
        self.n_expand = int(n_expand)

    @property
    def expand_pos(self):
        """Inverse of the expand CSR's permutation (int32): position of item t in the order of its expand row."""
        if getattr(self, "_expand_pos", None) is None:
            perm, _ = self.expand.csr
            pos = torch.empty_like(perm)
            pos[perm.long()] = torch.arange(perm.shape[0], device=perm.device, dtype=perm.dtype)
            self._expand_pos = pos
        return self._expand_pos

    @property
    def seg_off(self):
        return self.reduce.csr[1]

    def set_atom_blocks(self, reduce_atom: "RowIndex", expand_atom: torch.Tensor, n_atoms: int, max_rows=None):
        """Declare the quadruplet structure of GemNet-Q (data_container.py:331-397): r(t) — a reduce edge c -> a — and g(t)
        — an intermediate triplet a <- b <- d — of every entry end in the same target atom, and the expand rows are sorted by
        that atom.  `reduce_atom`: target atom of every reduce row (a RowIndex: its CSR groups the edges by atom);
        `expand_atom`: target atom of every expand row (non-decreasing).  Enables the fused per-atom x-adjoint
        (gn_bil_expand_atoms_ang_f32: no per-quadruplet rows in memory)."""
        self._ab_src = (reduce_atom, expand_atom, int(n_atoms))
        self._ab_max_rows = None if max_rows is None else int(max_rows)     # static bound of the reduce edges per atom (padded.py)
        self._atom_blocks = None

    @property
    def atom_blocks(self):
        """(a_perm int32 | None, a_seg int32 (A+1), j_off int32 (A+1), max_J) or None.  Built once per batch; the maximum is
        read back from the device (a host sync: `GraphPlan.warm` does it outside of any capture)."""
        src = getattr(self, "_ab_src", None)
        if src is None:
            return None
        if self._atom_blocks is None:
            reduce_atom, expand_atom, A = src
            perm, seg = reduce_atom.csr
            cnt = torch.bincount(expand_atom, minlength=A)
            j_off = torch.zeros(A + 1, dtype=torch.int64, device=cnt.device)
            torch.cumsum(cnt, 0, out=j_off[1:])
            max_J = int(cnt.max().item()) if cnt.numel() else 0
            self._atom_blocks = (perm, seg.to(torch.int32).contiguous(), j_off.to(torch.int32).contiguous(), max_J)
        return self._atom_blocks

    ROW_TILE = int(os.environ.get("GEMNET_ROW_TILE", "64"))      # expand rows per wave of gn_bil_expand_rows_ang_f32 (32 | 64)

    @property
    def row_grid(self):
        """(a_perm | None, a_seg, j_off, qmap, g_off, task_atom, task_row0, n_tasks) for gn_bil_expand_rows_ang_f32 (the x-adjoint
        of the quadruplet layer without per-quadruplet rows in memory), or None.  qmap: for atom a, starting at g_off[a], the
        dense grid [reduce edges into a][expand rows of a] of quadruplet numbers, -1 where the pair has none.  Built once per
        batch with device-side tensor ops; the two sizes are read back (host syncs: `GraphPlan.warm` does this outside of any
        capture — inside a capture the property answers None and the two-pass form runs)."""
        src = getattr(self, "_ab_src", None)
        if src is None:
            return None
        if getattr(self, "_row_grid", None) is None:
            static = getattr(self, "_ab_max_rows", None)
            if static is None and self.reduce.idx32.is_cuda and torch.cuda.is_current_stream_capturing():
                return None
            reduce_atom, expand_atom, A = src
            dev = expand_atom.device
            perm, seg = reduce_atom.csr                       # edges grouped by target atom
            seg64 = seg.to(torch.int64)
            nE = seg64[1:] - seg64[:-1]
            # (no bincount: its output size is data-dependent — a host sync, impossible inside a capture)
            nJ = torch.zeros(A, dtype=torch.int64, device=dev).index_add_(
                0, expand_atom.to(torch.int64), torch.ones(expand_atom.shape[0], dtype=torch.int64, device=dev))
            j_off = torch.zeros(A + 1, dtype=torch.int64, device=dev)
            torch.cumsum(nJ, 0, out=j_off[1:])
            g_off = torch.zeros(A + 1, dtype=torch.int64, device=dev)
            torch.cumsum(nE * nJ, 0, out=g_off[1:])
            n_edges = int(reduce_atom.idx32.shape[0])
            pos = torch.arange(n_edges, device=dev, dtype=torch.int64)
            atom_of_edge = reduce_atom.idx32.to(torch.int64)
            if perm is None:
                inv = pos
            else:
                inv = torch.empty(n_edges, dtype=torch.int64, device=dev)
                inv[perm.to(torch.int64)] = pos
            e_rank = inv - seg64[atom_of_edge]                # position of an edge in its atom's list
            r = self.reduce.idx32.to(torch.int64)
            a_q = atom_of_edge[r]
            flat = g_off[a_q] + e_rank[r] * nJ[a_q] + (self.expand.idx32.to(torch.int64) - j_off[a_q])
            nt = (nJ + self.ROW_TILE - 1) // self.ROW_TILE
            t_off = torch.zeros(A + 1, dtype=torch.int64, device=dev)
            torch.cumsum(nt, 0, out=t_off[1:])
            if static is None:
                g_total, n_tasks = (int(v) for v in torch.stack([g_off[-1], t_off[-1]]).tolist())
            else:
                # static capacities from the sizes of the (capacity-sized) arrays and the in-degree bound: nothing is read back, so
                # the construction can sit inside a captured graph (padded.py); tasks beyond the real count carry atom -1
                g_total = static * int(self.n_expand)
                n_tasks = int(self.n_expand) // self.ROW_TILE + A + 1
            if g_total >= 2 ** 31:
                self._row_grid = ()
                return None
            qmap = torch.full((max(g_total, 1),), -1, dtype=torch.int32, device=dev)
            qmap[flat] = torch.arange(self.size, device=dev, dtype=torch.int32)
            t = torch.arange(n_tasks, device=dev, dtype=torch.int64)
            task_atom = torch.searchsorted(t_off[1:].contiguous(), t, right=True)
            real = task_atom < A
            task_atom = torch.where(real, task_atom, torch.zeros_like(task_atom))
            task_row0 = (t - t_off[task_atom]) * self.ROW_TILE
            task_atom = torch.where(real, task_atom, torch.full_like(task_atom, -1))
            i32 = lambda x: x.to(torch.int32).contiguous()   # noqa: E731
            self._row_grid = (perm, i32(seg), i32(j_off), qmap, i32(g_off), i32(task_atom), i32(task_row0), n_tasks)
        return self._row_grid or None

    def set_row_groups(self, row_group: torch.Tensor, n_groups: int, max_rows=None):
        """Declare that r(t) and g(t) of every entry fall in the same group of rows (`row_group[row]`), as the
        triplets c->a<-b do with the target atom a of both edges (data_container.py:262-300).
        `max_rows`: a known upper bound of the rows per group (the largest in-degree of an atom) — without it the exact
        maximum is read back from the device (a host sync: not possible inside a hipGraph capture)."""
        assert self.n_reduce == self.n_expand == int(row_group.shape[0])
        self._row_group, self._n_groups, self._groups = row_group, int(n_groups), None
        self._max_rows = None if max_rows is None else int(max_rows)

    @property
    def groups(self):
        """(grp_rows, grp_off, grp_kseg, rposT, max_rows) for gn_bil_reduce_t_grouped_f32, or None."""
        if getattr(self, "_row_group", None) is None:
            return None
        _late_wait(self)
        if self._groups is None:
            key = self._row_group
            if key.is_cuda and _native_csr():
                from . import kernels as _K
                rows32, off = _K.csr_build(key.to(torch.int32), self._n_groups)
                rows = rows32.to(torch.int64)
            else:
                rows = torch.argsort(key, stable=True)
                off = _seg_offsets_sorted(key[rows], self._n_groups)
            rank = torch.empty_like(rows)
            rank[rows] = torch.arange(rows.shape[0], device=rows.device) - off.to(torch.int64)[key[rows]]
            permT, segT = self.expand.csr
            rposT = rank[self.reduce.idx64[permT.to(torch.int64)]].to(torch.int32).contiguous()
            kseg = torch.stack([segT[:-1][rows], segT[1:][rows]], dim=1).to(torch.int32).contiguous()
            if getattr(self, "_max_rows", None) is not None:
                max_rows = self._max_rows
            else:
                max_rows = int((off[1:] - off[:-1]).max().item()) if rows.shape[0] else 0
            self._groups = (rows.to(torch.int32).contiguous(), off, kseg, rposT, max_rows)
        return self._groups


class GraphPlan:
    """All index plans of one batch.  Built once per batch (`GraphPlan.from_inputs`), cached in the
    inputs dict under the key "_plan" so repeated forwards (MD, benchmarks, graph replay) reuse it."""

    def __init__(self, inputs: dict, triplets_only: bool):
        Z = inputs["Z"]
        dev = Z.device
        self.n_atoms = int(Z.shape[0])
        id_a, id_c = inputs["id_a"], inputs["id_c"]
        self.n_edges = int(id_a.shape[0])
        if "N" in inputs:
            self.n_mol = int(inputs["N"].shape[0])
        else:  # reference semantics (gemnet.py:578) — costs a host sync
            self.n_mol = int(inputs["batch_seg"].max().item()) + 1 if self.n_atoms else 0
        self.id_a = RowIndex(id_a, self.n_atoms)
        self.id_c = RowIndex(id_c, self.n_atoms)
        swap = inputs["id_swap"]
        self.id_swap = RowIndex(swap, self.n_edges)
        self.id_swap.inverse = self.id_swap  # id_swap is an involution (data_container.py:303-308)
        self.batch_seg = RowIndex(inputs["batch_seg"], self.n_mol, is_sorted=True)
        self.trip = SegmentPlan(inputs["id3_reduce_ca"], inputs["id3_expand_ba"], self.n_edges, self.n_edges)
        # reduce c->a and expand b->a share the target atom; "max_in_degree": optional static bound (padded.py)
        self.trip.set_row_groups(id_a, self.n_atoms, max_rows=inputs.get("max_in_degree"))
        # atom triples of each triplet for the angle c<-a->b (gemnet.py:442-444)
        r, x = inputs["id3_reduce_ca"], inputs["id3_expand_ba"]
        # t_c, t_a follow the (sorted) reduce edge: their CSR by atom is the atoms' edge lists expanded into triplet ranges
        T3 = int(r.shape[0])
        self.t_c = RowIndex(id_c[r], self.n_atoms, csr_builder=lambda: expanded_csr(self.id_c, self.trip.seg_off, T3))
        self.t_a = RowIndex(id_a[r], self.n_atoms, csr_builder=lambda: expanded_csr(self.id_a, self.trip.seg_off, T3))
        self.t_b = RowIndex(id_c[x], self.n_atoms)
        self.z_rows = RowIndex(Z - 1, 93)
        self.id_undir = RowIndex(inputs["id_undir"], self.n_edges // 2)
        if "N" in inputs:
            self.atoms_per_mol = inputs["N"].to(torch.float32)
        else:
            self.atoms_per_mol = torch.bincount(inputs["batch_seg"], minlength=self.n_mol).to(torch.float32)
        self.triplets_only = triplets_only
        if not triplets_only:
            i_a, i_b = inputs["id4_int_a"], inputs["id4_int_b"]
            self.n_int = int(i_a.shape[0])
            self.int_a, self.int_b = RowIndex(i_a, self.n_atoms), RowIndex(i_b, self.n_atoms)
            red_ca, exp_db = inputs["id4_reduce_intm_ca"], inputs["id4_expand_intm_db"]
            red_ab, exp_ab = inputs["id4_reduce_intm_ab"], inputs["id4_expand_intm_ab"]
            self.n_intm = int(exp_db.shape[0])
            self.intm_db = RowIndex(exp_db, self.n_edges)
            self.intm_ab = RowIndex(exp_ab, self.n_int, is_sorted=True)
            self.quad = SegmentPlan(inputs["id4_reduce_ca"], inputs["id4_expand_abd"], self.n_edges, self.n_intm)
            # reduce edge c -> a and intermediate triplet a <- b <- d share the target atom a; the latter are sorted by it
            # (only the fused per-atom x-adjoint reads it — kernels.USE_ATOM_BLOCKS, off by default: no gather / bincount /
            #  host read-back per batch for a structure nobody consumes)
            from . import kernels as _K
            if _K.USE_ATOM_BLOCKS or _K.USE_ROW_GRID:
                self.quad.set_atom_blocks(self.id_a, i_a[exp_ab], self.n_atoms, max_rows=inputs.get("max_in_degree"))
            A = self.n_atoms
            self.quad_geom = {
                # a - b <- d per intermediate triplet (gemnet.py:385-388)
                "a_of_exp": RowIndex(i_a[exp_ab], A), "b_of_exp": RowIndex(i_b[exp_ab], A),
                "d_of_exp": RowIndex(id_c[exp_db], A),
                # c -> a <- b per intermediate triplet (gemnet.py:399-402)
                "c_of_red": RowIndex(id_c[red_ca], A), "a_of_red": RowIndex(id_a[red_ca], A),
                "b_of_red": RowIndex(i_b[red_ab], A),
                "reduce_cab": RowIndex(inputs["id4_reduce_cab"], self.n_intm),
            }
            # the four atoms of every quadruplet c -> a - b <- d (data_container.py:393-397)
            q_ca, q_db = inputs["id4_reduce_ca"], inputs["id4_expand_db"]
            self.q_c, self.q_a = RowIndex(id_c[q_ca], A), RowIndex(id_a[q_ca], A)
            self.q_d, self.q_b = RowIndex(id_c[q_db], A), RowIndex(id_a[q_db], A)
        self.device = dev

    @staticmethod
    def from_inputs(inputs: dict, triplets_only: bool) -> "GraphPlan":
        plan = inputs.get("_plan")
        if plan is None or plan.triplets_only != triplets_only or plan.device != inputs["Z"].device:
            plan = GraphPlan(inputs, triplets_only)
            inputs["_plan"] = plan
        return plan

    def row_indices(self):
        """The row indices whose CSR some kernel of the path reads (id_swap's adjoint is a gather with its inverse;
        id_undir only with coupled direct forces: built on first use there)."""
        out = [self.id_a, self.id_c, self.batch_seg, self.trip.reduce, self.trip.expand,
               self.t_c, self.t_a, self.t_b, self.z_rows]
        if not self.triplets_only:
            out += [self.int_a, self.int_b, self.intm_db, self.intm_ab, self.quad.reduce, self.quad.expand]
            # (the CSRs of the four atoms of every quadruplet — q_c, q_a, q_b, q_d: four sorts of Q keys, 9 M at B = 32 — are NOT
            #  built here: the force assembly sums per reduce edge / intermediate triplet first (ops._quad_adjoint with the
            #  plan: the only form the model calls); without a plan they are built on first use)
            # likewise the c -> a <- b structures of the intermediate triplets and `reduce_cab` (a sort of Q keys): only the
            # composite closure's calculate_angles reads them, on the main stream — built on first use there
            out += [self.quad_geom[k] for k in ("a_of_exp", "b_of_exp", "d_of_exp")]
        return out

    def late_indices(self):
        """The structures only adjoint kernels read (CSR of the gathers' transposes, the triplet groups of the x-adjoint):
        the two sorts of T keys among them."""
        return [self.id_c, self.trip.expand, self.t_c, self.t_a, self.t_b, self.z_rows]

    def warm(self, late_stream=None):
        """Materialise every lazily-built CSR: before a hipGraph capture, and before a forward that forks onto a side
        stream — a structure first built (sorted) on one stream and read by a kernel of the other is a race.
        `late_stream` (triplets-only plans): the structures of `late_indices` and the triplet groups are built there,
        beside the forward pass that does not read them; their readers wait for `late_event` (`_late_wait`), and
        `join_late` orders the calling stream behind the construction (a capture must end with every stream joined)."""
        if not getattr(self, "_warmed", False):
            late = self.late_indices() if (late_stream is not None and self.triplets_only) else []
            for ri in self.row_indices():
                if not any(ri is l for l in late):
                    ri.csr
            if late:
                main = torch.cuda.current_stream()
                late_stream.wait_stream(main)
                with torch.cuda.stream(late_stream):
                    for ri in late:
                        ri.csr
                    self.trip.groups
                    ev = torch.cuda.Event()
                    ev.record(late_stream)
                for obj in late + [self.trip]:
                    obj._late = (late_stream, ev)
                self._late_event = ev
            else:
                self.trip.groups
            if not self.triplets_only and _atom_blocks_on():
                self.quad.atom_blocks
            if not self.triplets_only:
                from . import kernels as _K
                if _K.USE_ROW_GRID:
                    self.quad.row_grid        # (two size read-backs: here, outside of any capture)
            self._warmed = True
        return self

    def join_late(self):
        ev = getattr(self, "_late_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def to(self, *args, **kwargs):
        """Trainer.dict2device (trainer.py:313-318) calls .to(device) on every dict value."""
        return self
