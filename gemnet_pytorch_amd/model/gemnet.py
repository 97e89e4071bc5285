"""`GemNet` — drop-in for the reference's `gemnet.model.gemnet.GemNet` (gemnet/model/gemnet.py:21-790)
with the InteractionBlock hot path running on hand-written gfx950 HIP kernels.

Kept verbatim from the reference: constructor signature (:82-113), `forward(inputs) -> (E, F)`
(:453-615; the force is differentiated w.r.t. a private leaf sharing `inputs["R"]`'s storage instead of toggling
`inputs["R"].requires_grad` :494,:613 — the caller's tensor is left untouched), `predict` (:780-784),
`load_weights`/`save_weights` (:786-790), attribute names the trainer reaches into
(`mlp_rbf3/.mlp_cbf3/.mlp_rbf_h/.mlp_rbf4/.mlp_cbf4/.mlp_sbf4/.mlp_rbf_out`, `num_blocks`,
`triplets_only`, `direct_forces`, trainer.py:263-278,417) and the state_dict key set.

Different by design: no zero-padded (E,Kmax,.) tensors, no `Kidx` host syncs, no sympy at
construction (closed-form basis kernels), no torch_scatter (CSR segmented sums), the concat-Dense
split into atom-side and edge-side GEMMs.  fp32 on a HIP device only; CPU tensors raise.

Force graph policy: the reference always builds the force with `create_graph=True`.  Here the
second-order graph is built when it can be used — `self.training and torch.is_grad_enabled()` —
or when `self.force_graph` is set to True/False explicitly.
"""
import contextlib
import os

import torch

from .. import ops
from ..graph import GraphPlan
from .layers import (AtomEmbedding, BesselBasisLayer, Dense, EdgeEmbedding,
                     EfficientInteractionDownProjection, InteractionBlock,
                     InteractionBlockTripletsOnly, OutputBlock, SphericalBasisLayer, TensorBasisLayer)
from .scaling import AutomaticFit


_nullcontext = contextlib.nullcontext
# A/B switch: one running gradient for rbf_out (five consumers on the side stream) instead of four engine-side adds.
# Off: same-box A/B on MI355X (profiles/r3_ab.txt) 2.777 / 2.747 ms with it, 2.729 / 2.685 ms without — the chained
# in-place sums order the output blocks' adjoints behind each other, the engine's adds did not.
_RBF_OUT_ACC = os.environ.get("GEMNET_RBF_OUT_ACC", "0") == "1"
_H3_GUARD = os.environ.get("GEMNET_H3_GUARD", "1") == "1"     # see GemNet.forward
# Side-stream placement switches.  All three were forced off in round 3 because hipGraph replays stopped matching the eager
# run with them; round 4 located the cause below the library — packed-FP32 instructions of the fused aggregation adjoint
# returning wrong lanes when its waves share CUs with the chain kernels of another graph branch (csrc/aggregate.hip,
# tools/exp/graph_corun.py, docs/HISTORY.md section 11) — and removed it, so they are on again.  `=0` restores the in-line forms.
_TRAIN_OVERLAP = os.environ.get("GEMNET_TRAIN_OVERLAP", "1") == "1"  # force training: output blocks on the side stream
_Q_OVERLAP = os.environ.get("GEMNET_Q_OVERLAP", "1") == "1"          # quadruplet models: output blocks on the side stream
_RBF_OUT_SIDE = os.environ.get("GEMNET_RBF_OUT_SIDE", "1") == "1"    # output-block radial projection in the forked head
_PLAN_LATE = os.environ.get("GEMNET_PLAN_LATE", "1") == "1"          # adjoint-only index structures on a stream of their own (in captures)


def K_chain_mode():
    from .. import kernels
    return kernels.current_mode()


class GemNet(torch.nn.Module):
    def __init__(
        self,
        num_spherical: int,
        num_radial: int,
        num_blocks: int,
        emb_size_atom: int,
        emb_size_edge: int,
        emb_size_trip: int,
        emb_size_quad: int,
        emb_size_rbf: int,
        emb_size_cbf: int,
        emb_size_sbf: int,
        emb_size_bil_quad: int,
        emb_size_bil_trip: int,
        num_before_skip: int,
        num_after_skip: int,
        num_concat: int,
        num_atom: int,
        triplets_only: bool,
        num_targets: int = 1,
        direct_forces: bool = False,
        cutoff: float = 5.0,
        int_cutoff: float = 10.0,
        envelope_exponent: int = 5,
        extensive=True,
        forces_coupled: bool = False,
        output_init="HeOrthogonal",
        activation: str = "swish",
        scale_file=None,
        name="gemnet",
        **kwargs,
    ):
        super().__init__()
        assert num_blocks > 0
        self.num_targets = num_targets
        self.num_blocks = num_blocks
        self.extensive = extensive
        self.forces_coupled = forces_coupled
        self.direct_forces = direct_forces
        self.triplets_only = triplets_only
        self.num_spherical = num_spherical
        self.force_graph = None  # None: auto (training & grad enabled); True/False: forced
        # arithmetic of the LDS-resident Dense stacks: None = the package default ("h3": fp32 operands as two fp16 planes,
        # three products), "split6" = three bf16 planes / six products (fp32 exponent range), "f32" = the f32-input MFMA — all
        # three at fp32 accuracy (force MAE 1e-6 .. 4e-6 eV/A against float64).  Single-plane bf16 / three-product modes exist
        # in the chain KERNEL (kernels.CHAIN_MODES, tests) but are not a model option: measured on the configs[4] shard they are
        # slower than the default (116 vs 109 ms) at 4e-2 eV/A (docs/HISTORY.md section 14)
        self.matmul_precision = None
        self.overlap_output_blocks = True
        self._side = None
        self._cot = None      # cached force cotangents (see _cotangent)
        self._wcache = {}  # derived (transposed / contiguous) copies of frozen weights, see ops.weight_cache
        self._packs = ops.PackRegistry()   # split-bf16 planes of the trainable weights, repacked once per training step

        AutomaticFit.reset()

        self.rbf_basis = BesselBasisLayer(num_radial, cutoff=cutoff, envelope_exponent=envelope_exponent)
        if not triplets_only:
            self.cbf_basis = SphericalBasisLayer(num_spherical, num_radial, cutoff=int_cutoff,
                                                 envelope_exponent=envelope_exponent, efficient=False)
            self.sbf_basis = TensorBasisLayer(num_spherical, num_radial, cutoff=cutoff,
                                              envelope_exponent=envelope_exponent, efficient=True)
        self.cbf_basis3 = SphericalBasisLayer(num_spherical, num_radial, cutoff=cutoff,
                                              envelope_exponent=envelope_exponent, efficient=True)

        # shared down projections (gemnet.py:158-204)
        if not triplets_only:
            self.mlp_rbf4 = Dense(num_radial, emb_size_rbf, activation=None, bias=False)
            self.mlp_cbf4 = Dense(num_radial * num_spherical, emb_size_cbf, activation=None, bias=False)
            self.mlp_sbf4 = EfficientInteractionDownProjection(num_spherical ** 2, num_radial, emb_size_sbf)
        self.mlp_rbf3 = Dense(num_radial, emb_size_rbf, activation=None, bias=False)
        self.mlp_cbf3 = EfficientInteractionDownProjection(num_spherical, num_radial, emb_size_cbf)
        self.mlp_rbf_h = Dense(num_radial, emb_size_rbf, activation=None, bias=False)
        self.mlp_rbf_out = Dense(num_radial, emb_size_rbf, activation=None, bias=False)

        self.atom_emb = AtomEmbedding(emb_size_atom)
        self.edge_emb = EdgeEmbedding(emb_size_atom, num_radial, emb_size_edge, activation=activation)

        block_cls = InteractionBlockTripletsOnly if triplets_only else InteractionBlock
        int_blocks = []
        for i in range(num_blocks):
            kw = dict(emb_size_atom=emb_size_atom, emb_size_edge=emb_size_edge, emb_size_trip=emb_size_trip,
                      emb_size_quad=emb_size_quad, emb_size_rbf=emb_size_rbf, emb_size_cbf=emb_size_cbf,
                      emb_size_bil_trip=emb_size_bil_trip, num_before_skip=num_before_skip,
                      num_after_skip=num_after_skip, num_concat=num_concat, num_atom=num_atom,
                      activation=activation, scale_file=scale_file, name=f"IntBlock_{i+1}")
            if not triplets_only:
                kw.update(emb_size_sbf=emb_size_sbf, emb_size_bil_quad=emb_size_bil_quad)
            int_blocks.append(block_cls(**kw))
        out_blocks = [
            OutputBlock(emb_size_atom=emb_size_atom, emb_size_edge=emb_size_edge, emb_size_rbf=emb_size_rbf,
                        nHidden=num_atom, num_targets=num_targets, activation=activation,
                        output_init=output_init, direct_forces=direct_forces, scale_file=scale_file,
                        name=f"OutBlock_{i}")
            for i in range(num_blocks + 1)
        ]
        self.out_blocks = torch.nn.ModuleList(out_blocks)
        self.int_blocks = torch.nn.ModuleList(int_blocks)

    # -------------------------------------------------------------------------- geometry (P9)
    @staticmethod
    def calculate_interatomic_vectors(R, id_s, id_t):
        """id_s / id_t: RowIndex of the source / target atoms (gemnet.py:261-286)."""
        V_st = ops.gather_rows(R, id_t) - ops.gather_rows(R, id_s)
        D_st = torch.sqrt(torch.sum(V_st ** 2, dim=1))
        return D_st, V_st / D_st[..., None]

    @staticmethod
    def calculate_neighbor_angles(R_ac, R_ab):
        """atan2(max(|u x v|, 1e-9), u.v)  (gemnet.py:288-311)."""
        x = torch.sum(R_ac * R_ab, dim=1)
        y = torch.linalg.cross(R_ac, R_ab, dim=-1).norm(dim=-1)
        y = torch.clamp(y, min=1e-9)
        return torch.atan2(y, x)

    @staticmethod
    def vector_rejection(R_ab, P_n):
        a_x_b = torch.sum(R_ab * P_n, dim=-1)
        b_x_b = torch.sum(P_n * P_n, dim=-1)
        return R_ab - (a_x_b / b_x_b)[:, None] * P_n

    @staticmethod
    def calculate_angles3(R, plan):
        """(gemnet.py:420-451) angle c<-a->b of every triplet."""
        Ra = ops.gather_rows(R, plan.t_a)
        R_ac = ops.gather_rows(R, plan.t_c) - Ra
        R_ab = ops.gather_rows(R, plan.t_b) - Ra
        return GemNet.calculate_neighbor_angles(R_ac, R_ab)

    @staticmethod
    def calculate_angles(R, plan):
        """Quadruplet angles (gemnet.py:334-418): Phi_cab (Q,), Phi_abd (I,), Theta_cabd (Q,)."""
        q = plan.quad_geom
        Ra = ops.gather_rows(R, q["a_of_exp"])
        Rb = ops.gather_rows(R, q["b_of_exp"])
        Rd = ops.gather_rows(R, q["d_of_exp"])
        R_ba, R_bd = Ra - Rb, Rd - Rb
        angle_abd = GemNet.calculate_neighbor_angles(R_ba, R_bd)
        R_bd_proj = ops.gather_rows(GemNet.vector_rejection(R_bd, R_ba), plan.quad.expand)

        Rc = ops.gather_rows(R, q["c_of_red"])
        Ra2 = ops.gather_rows(R, q["a_of_red"])
        Rb2 = ops.gather_rows(R, q["b_of_red"])
        R_ac, R_ab = Rc - Ra2, Rb2 - Ra2
        angle_cab = ops.gather_rows(GemNet.calculate_neighbor_angles(R_ab, R_ac)[:, None], q["reduce_cab"])[:, 0]
        R_ac_proj = ops.gather_rows(GemNet.vector_rejection(R_ac, R_ab), q["reduce_cab"])
        angle_cabd = GemNet.calculate_neighbor_angles(R_ac_proj, R_bd_proj)
        return angle_cab, angle_abd, angle_cabd

    # ---------------------------------------------------------------------------------- forward
    def _energy(self, R, plan):
        T = self.triplets_only
        b3 = self.cbf_basis3
        # Output blocks on a side stream (all model kinds, inference and force training; the round-3 restrictions are gone,
        # see the switches at the top of the file).
        overlap = self.overlap_output_blocks and (not ops.train2_enabled() or _TRAIN_OVERLAP)
        side = self._side_stream(R.device) if overlap and R.is_cuda and (T or _Q_OVERLAP) else None
        # The head of the forward is a string of small launches (110 us at B = 32); only distances -> edge embedding ->
        # rbf3 are needed by the first kernel of block 0.  With a side stream the rest forks off: the triplet angles,
        # the atom embedding and its two concat-Dense terms need positions / atomic numbers only and run beside the edge
        # geometry kernel; the circular-basis and the atom-update / output radial projections run beside the edge
        # embedding.  Autograd replays a node on its forward stream, so the tail of the backward overlaps the same way.
        # Triplets-only models only for now: the fork is measured (+2.5-3.8 %) and verified (graph replay == eager, bit
        # for bit) on GemNet-T; the quadruplet models spend 0.1 of 13.6 ms in this head and keep the sequential form.
        fork = side is not None and ops.is_fused() and T
        main = torch.cuda.current_stream(R.device) if fork else None
        h = terms = rbf_W1_3 = rbf_h = rbf_out = None
        if ops.is_fused():
            if fork:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    sph3 = ops.share_gradient(ops.trip_basis(R, plan.t_c, plan.t_a, plan.t_b, self.num_spherical))
                    h = self.atom_emb(plan.z_rows)
                    terms = self.edge_emb.atom_terms(h)
                    ev_a = torch.cuda.Event()
                    ev_a.record(side)
            # one launch: distances + Bessel rbf + spherical-Bessel radial basis; one launch: angles + Y_l0
            D_ca, V_ca, rbf, rad3 = ops.edge_basis(R, self.rbf_basis.frequencies, plan.id_c, plan.id_a,
                                                   b3.z_ln, b3.n_ln, b3.cutoff, b3.p,
                                                   want_V=self.direct_forces)
            if fork:
                ev_1 = torch.cuda.Event()
                ev_1.record(main)
                side.wait_event(ev_1)
                with torch.cuda.stream(side):
                    rbf.record_stream(side)
                    rad3.record_stream(side)
                    rbf_W1_3 = self.mlp_cbf3(rad3)
                    rbf_h = self.mlp_rbf_h(rbf)
                    if _RBF_OUT_SIDE:
                        rbf_out = self.mlp_rbf_out(rbf)
                    ev_b = torch.cuda.Event()
                    ev_b.record(side)
                # (all consumers of the output blocks' radial projection run on the side stream: so does the projection)
                if not _RBF_OUT_SIDE:
                    rbf_out = self.mlp_rbf_out(rbf)
                if _RBF_OUT_ACC:
                    rbf_out = ops.accumulate_gradient(rbf_out, stream=side)
                main.wait_event(ev_a)
                for t in (sph3, h) + tuple(terms):
                    t.record_stream(main)
            else:
                sph3 = ops.share_gradient(ops.trip_basis(R, plan.t_c, plan.t_a, plan.t_b, self.num_spherical))
        elif ops.train2_enabled() and not self.direct_forces:
            # force training: distances and triplet angles as twice-differentiable kernels (ops_train._Dist2 / _Angle2),
            # the bases on their closed derivative kernels
            from .. import ops_train
            D_ca, V_ca = ops_train.distances(R, plan.id_c, plan.id_a), None
            rbf = self.rbf_basis(D_ca)
            rad3, sph3 = b3(D_ca, ops_train.triplet_angles(R, plan.t_c, plan.t_a, plan.t_b))
        else:
            D_ca, V_ca = self.calculate_interatomic_vectors(R, plan.id_c, plan.id_a)
            rbf = self.rbf_basis(D_ca)
            rad3, sph3 = b3(D_ca, self.calculate_angles3(R, plan))
        cbf4_done = False       # (the fused branch below may project the circular basis through mlp_cbf4 itself)
        if not T and ops.is_fused():
            b4, qg = self.cbf_basis, plan.quad_geom
            # interaction-edge radial basis (cutoff = int_cutoff), a-b<-d angles and the quadruplet
            # harmonics, each one launch, straight from the positions
            _, _, _, rad4 = ops.edge_basis(R, None, plan.int_b, plan.int_a, b4.z_ln, b4.n_ln, b4.cutoff, b4.p,
                                           want_rbf=False)
            y_abd = ops.trip_basis(R, qg["a_of_exp"], qg["b_of_exp"], qg["d_of_exp"], self.num_spherical)
            S, NR = self.num_spherical, b4.num_radial
            cbf4 = None
            if self.mlp_cbf4.bias is None and not self.mlp_cbf4.act:
                # gather + Hadamard with Y_l0 + mlp_cbf4 in one pass (csrc/cbf.hip); the projected basis is marked below
                cbf4 = ops.cbf_project(rad4, plan.intm_ab, y_abd, self.mlp_cbf4.weight)
            cbf4_done = cbf4 is not None
            if not cbf4_done:
                cbf4 = (ops.gather_rows(rad4, plan.intm_ab) * y_abd[:, :, None]).reshape(-1, S * NR)
            # (angle form for the published quadruplet widths: the bilinear kernels rebuild Y_lm from 16 B per quadruplet)
            ang = self.int_blocks[0].quad_interaction.mlp_sbf.weight.shape[:2] == (32, 32)
            sbf4 = (rad3, ops.share_gradient(ops.quad_basis(R, plan.q_c, plan.q_a, plan.q_b, plan.q_d, S, plan=plan,
                                                            angle_form=ang)))
        elif not T:
            qi = self.int_blocks[0].quad_interaction.mlp_sbf.weight
            if (ops.train2_enabled() and not self.direct_forces and self.num_spherical == 7
                    and ops.quad_train2_enabled(qi.shape[0], qi.shape[1], self.num_spherical ** 2)):
                # force training of the quadruplet interaction on fused kernels: the a - b <- d angles and the two quadruplet
                # angles as twice-differentiable kernels, the tensor basis in ANGLE form (16 B per quadruplet; the bilinear
                # twins rebuild Y_lm and its tangent in-kernel, ops_train._BilinearAng2) — no (Q, 49) array in any of the four
                # sweeps, no ATen launch over (Q, 3) temporaries
                from .. import ops_train
                qg = plan.quad_geom
                D_ab = ops_train.distances(R, plan.int_b, plan.int_a)
                Phi_abd = ops_train.triplet_angles(R, qg["a_of_exp"], qg["b_of_exp"], qg["d_of_exp"])
                cbf4 = self.cbf_basis(D_ab, Phi_abd, plan.intm_ab)
                sbf4 = (rad3, ops_train.quad_angles(R, plan.q_c, plan.q_a, plan.q_b, plan.q_d, plan))
            else:
                if ops.train2_enabled() and not self.direct_forces:
                    from .. import ops_train
                    D_ab = ops_train.distances(R, plan.int_b, plan.int_a)
                else:
                    D_ab, _ = self.calculate_interatomic_vectors(R, plan.int_b, plan.int_a)
                Phi_cab, Phi_abd, Theta_cabd = self.calculate_angles(R, plan)
                cbf4 = self.cbf_basis(D_ab, Phi_abd, plan.intm_ab)           # (I, S*R)
                # the tensor basis shares cutoff and radial tables with cbf_basis3: reuse rad3
                sbf4 = (rad3, ops.ylm(Phi_cab, Theta_cabd, self.num_spherical))  # ((E,S,R), (Q,S^2))

        if h is None:
            h = self.atom_emb(plan.z_rows)
        rbf = ops.accumulate_gradient(rbf)
        m = self.edge_emb(h, rbf, plan.id_c, plan.id_a, terms=terms)

        if not T:
            rbf4 = ops.accumulate_gradient(self.mlp_rbf4(rbf))
            if not cbf4_done:
                cbf4 = self.mlp_cbf4(cbf4)
            cbf4 = ops.accumulate_gradient(cbf4)     # (one consumer per block: no engine-side (I,16) adds)
            sbf4 = (ops.accumulate_gradient(self.mlp_sbf4(sbf4[0])), sbf4[1])
        else:
            rbf4 = cbf4 = sbf4 = None
        # radial projections shared by all blocks: their gradients are summed inside the consumers' backward kernels
        rbf3 = ops.accumulate_gradient(self.mlp_rbf3(rbf))
        if fork:
            main.wait_event(ev_b)
            for t in (rbf_W1_3, rbf_h):    # produced on the side stream, consumed (and summed: `stream=`) on the main one
                t.record_stream(main)
            cbf3 = (ops.accumulate_gradient(rbf_W1_3, stream=main), sph3)
            rbf_h = ops.accumulate_gradient(rbf_h, stream=main)
        else:
            cbf3 = (ops.accumulate_gradient(self.mlp_cbf3(rad3)), sph3)
            rbf_h = ops.accumulate_gradient(self.mlp_rbf_h(rbf))
            rbf_out = self.mlp_rbf_out(rbf)
            if _RBF_OUT_ACC:
                rbf_out = ops.accumulate_gradient(rbf_out, stream=side)

        # OutputBlock i only feeds the final energy sum: it runs on a side stream, concurrently with
        # InteractionBlock i+1 (and, since autograd replays a node on its forward stream, so does its
        # backward).  Its kernels are atom-side (A = 1024 rows: 32 workgroups) and latency-bound, so
        # overlapping them is free.  Captured hipGraphs keep the fork/join as graph edges.
        outs = []

        def ready():
            """Event on the main stream: (h, m) of this point exist.  None without a side stream."""
            if side is None:
                return None
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return ev

        def out_block(i, h, m, ev):
            # the energies are summed along the chain of output blocks (epilogue of each energy head); the direct
            # force terms, when present, are summed below
            E_sum = outs[-1][0] if outs else None
            if side is None:
                outs.append(self.out_blocks[i](h, m, rbf_out, plan.id_a, E_sum=E_sum))
                return
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for t in (h, m, rbf_out):
                    t.record_stream(side)
                outs.append(self.out_blocks[i](h, m, rbf_out, plan.id_a, E_sum=E_sum))

        # OutputBlock i is ISSUED after InteractionBlock i (it only waits for the event recorded before it): in a
        # captured hipGraph the first-captured successor of a fork keeps the queue, and with the output block issued
        # first the main chain changed queues at every block (10-17 us idle per hop, tools/timeline.py); the later
        # issue also gives the output block the higher autograd sequence number, so its backward is enqueued (on the
        # side stream) before the backward of the interaction block that needs its contribution to dE/dm.
        # (Issuing the output block of the LAST interaction block early — so that its backward is not enqueued between
        # the backward of output block nb, which the main chain waits for, and the main chain itself — measured 4 %
        # slower on the same box.)
        # Same-box A/B of the issue lag (output block i after interaction block i + lag - 1): lag 0 (before) 11.20 k,
        # 1 (here) 11.36-11.45 k, 2 11.16 k, 3 11.00 k, all at the end 10.84 k molecules/s.
        for i in range(self.num_blocks):
            ev = ready()
            h_i, m_i = h, m
            h, m = self.int_blocks[i](h=h, m=m, rbf4=rbf4, cbf4=cbf4, sbf4=sbf4, rbf3=rbf3, cbf3=cbf3,
                                      rbf_h=rbf_h, plan=plan)
            out_block(i, h_i, m_i, ev)
        out_block(self.num_blocks, h, m, ready())
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        E_a, F_ca = outs[-1][0], outs[0][1]
        for _, F in outs[1:]:
            F_ca = F_ca + F

        E_mol = ops.segsum_rows(E_a, plan.batch_seg)                      # (nMolecules, num_targets)
        if not self.extensive:
            E_mol = E_mol / plan.atoms_per_mol.clamp(min=1)[:, None]
        return E_mol, F_ca, V_ca

    def forward(self, inputs):
        """E, F as the reference's `GemNet.forward` (gemnet.py:453-615).
        Range guard of the default Dense arithmetic: the "h3" forward programs keep activations in two fp16 planes, so a
        value beyond 65 504 becomes inf — fitted scale factors keep activations O(1), a model with
        unfitted ones (the starting state of fit_scaling.py, foreign checkpoints) need not.  A non-finite result of an eager
        forward in that arithmetic switches THIS model to the bf16-plane form ("split6": fp32 exponent range), says so, and
        repeats the pass.  (One host read-back per eager forward.  A hipGraph cannot read back: a runner hands its
        `runtime.RangeFlag` in as `inputs["_range_flag"]`, the captured pass ends with the device-side check of the same rows,
        and the runner polls the flag — PaddedGraphRunner, DynamicForceField, ForceGraphs, TrainStep.)"""
        inputs = self.with_indices(inputs)
        with ops.exclusive():
            return self._forward_guarded(inputs)

    def with_indices(self, inputs):
        """`inputs` with the index arrays: as given when it has them, else (a `DataContainer(indices="device")` batch: Z, R, N
        on the device) a new dict with the arrays built on the GPU (index_device.ensure_indices).  Callers that CAPTURE a step
        call this first: the build reads sizes back, which a stream capture does not allow."""
        if "id_c" in inputs:
            return inputs
        from ..index_device import ensure_indices
        self._check_inputs(inputs["R"])
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("inputs without index arrays inside a stream capture: call model.with_indices(inputs) first")
        cutoff, int_cutoff = self.cbf_basis3.cutoff, getattr(getattr(self, "cbf_basis", None), "cutoff", 10.0)
        if "cutoffs" in inputs:     # DataContainer(indices="device"): the container's own cutoffs define the graph
            c = inputs["cutoffs"].detach().cpu().tolist()
            cutoff, int_cutoff = float(c[0]), float(c[1])
        return ensure_indices(inputs, cutoff, int_cutoff, self.triplets_only)

    def _forward_guarded(self, inputs):
        out = self._forward(inputs)
        R = inputs["R"]
        h3 = R.is_cuda and (self.matmul_precision or K_chain_mode()) == "h3"
        # (a padded batch — padded.py — names its real rows: the dummy molecule behind them is not the model's concern)
        rows = inputs.get("_guard_rows")
        flag = inputs.get("_range_flag")
        if flag is not None and R.is_cuda:
            # a runner's device-side range check (runtime.RangeFlag): two tiny launches + a copy of one word to pinned host
            # memory at the end of the pass — what a REPLAYED graph has instead of the read-back below
            E, F = out
            flag.watch(E.detach()[:rows[0]] if rows is not None else E.detach(),
                       F.detach()[:rows[1]] if rows is not None else F.detach())
        if _H3_GUARD and h3 and not torch.cuda.is_current_stream_capturing():
            seen = out if rows is None else (out[0][:rows[0]], out[1][:rows[1]])
            if not bool(torch.stack([torch.isfinite(t).all() for t in seen]).all()):
                import warnings
                warnings.warn("gemnet_pytorch_amd: non-finite energies / forces from the fp16-plane Dense arithmetic ('h3': "
                              "activations beyond 65504 overflow; are the scale factors fitted?) — this model now uses "
                              "matmul_precision = 'split6' (bf16 planes, fp32 range); the pass is repeated", RuntimeWarning)
                self.matmul_precision = "split6"
                self._wcache = {}
                if getattr(self, "_packs", None) is not None:
                    self._packs.clear()
                if flag is not None:
                    torch.cuda.current_stream().synchronize()
                    flag.reset()       # the eager pass reported it itself
                out = self._forward(inputs)
        return out

    PRECISIONS = (None, "h3", "split6", "f32")

    def _forward(self, inputs):
        R = inputs["R"]
        self._check_inputs(R)
        if self.matmul_precision not in self.PRECISIONS and not getattr(self, "_experimental_precision", False):
            raise ValueError(f"matmul_precision must be one of {self.PRECISIONS}; got {self.matmul_precision!r} (reduced-precision "
                             "operand modes are kernel-level experiments, not a model option: slower AND 4e-2 eV/A off on "
                             "BASELINE configs[4], docs/HISTORY.md section 14)")
        plan = GraphPlan.from_inputs(inputs, self.triplets_only)
        late = None
        pos_graph = False
        if R.is_cuda and self.overlap_output_blocks:
            # the output blocks run on a side stream: every lazily-built index structure they share with the main
            # stream (CSR sorts, triplet groups) must exist before the fork, not be built by whichever stream gets
            # there first (an un-warmed GemNet-Q batch aborted with a memory fault in one of three runs).
            # A plan that is built INSIDE a capture (padded.py: rebuilt by every replay) puts the structures that only the
            # adjoint kernels read — two sorts of T keys among them — on a stream of their own, beside the forward pass.
            if (_PLAN_LATE and self.triplets_only and not self.direct_forces and not getattr(plan, "_warmed", False)
                    and torch.cuda.is_current_stream_capturing()):
                late = self._side_stream(R.device, "plan")
            plan.warm(late_stream=late)
        if not self.direct_forces:
            # The reference flips `inputs["R"].requires_grad` on the caller's tensor (gemnet.py:494,:613).  Here the
            # force is differentiated w.r.t. a FRESH leaf that shares R's storage: autograd keeps a leaf's gradient
            # accumulator — tied to the stream of its first backward — on the tensor object, and positions that once
            # went through a forward on the default stream made later hipGraph captures of the same batch fail
            # (the engine synchronised the capturing stream with the default stream).
            # A caller whose R already takes part in an autograd graph (a non-leaf, or a leaf that requires grad:
            # Hessians, position-dependent losses) keeps its tensor, as in the reference.
            if R.requires_grad:
                pos_graph = True      # the caller may differentiate through the force w.r.t. its positions (ops.position_graph)
            else:
                R = R.detach().requires_grad_(True)
        # second-order graph only when it can be used (see module docstring)
        graph = self.force_graph
        if graph is None:
            graph = self.training and torch.is_grad_enabled() and not self.direct_forces
        # fused single-launch layers whenever no double backward will run through them; the
        # scale-factor fitting mode needs the unfused layer order to observe variances
        fused = not graph and not AutomaticFit.fitting_mode

        # force-by-autograd without a second-order graph: the graph of E is consumed right here, so
        # parameter gradients can never be requested -> weights are constants (enables ops.stack)
        const_w = fused and not self.direct_forces
        # force training: the Dense stacks as twice-differentiable single-launch Functions (ops_train.py), the rest on
        # the composite closure; the split-operand chain kernel only (its f32 sibling has no second-order source terms)
        mode = self.matmul_precision or K_chain_mode()
        # (one target: with several, each target's force pass would need its own record of the S2 / S3 sweeps per stack and
        #  one source term per target in S4 — those models train on the composite closure)
        t2 = bool(graph) and ops.USE_TRAIN2 and not AutomaticFit.fitting_mode and mode != "f32" and self.num_targets == 1
        with ops.weight_cache(self._wcache), ops.fused_first_order(fused), ops.param_grads(not const_w), \
                ops.train2(t2, self._packs if (t2 and R.is_cuda) else None), \
                ops.chain_mode(self.matmul_precision), ops.position_graph(pos_graph), \
                torch.enable_grad() if not self.direct_forces else _nullcontext():
            E_mol, F_ca, V_ca = self._energy(R, plan)

            if self.direct_forces:
                if self.forces_coupled:  # enforce |F_ac| = |F_ca| (gemnet.py:588-592)
                    F_ca = ops.segsum_rows(F_ca, plan.id_undir) * 0.5
                    F_ca = ops.gather_rows(F_ca, plan.id_undir)
                F_ji = F_ca[:, :, None] * V_ca[:, None, :]
                F_j = ops.segsum_rows(F_ji, plan.id_a)                         # (nAtoms, num_targets, 3)
            else:
                with ops.param_grads(False):  # only dE/dR is needed here
                    # F = -d(sum_mol E)/dR: the cotangent -1 (one column per target) goes in directly — no reduction,
                    # no ones_like fill and no negation launch
                    if self.num_targets > 1:
                        F_j = torch.stack(
                            [torch.autograd.grad(E_mol, R, grad_outputs=self._cotangent(E_mol, i), create_graph=graph,
                                                 retain_graph=True)[0] for i in range(self.num_targets)], dim=1)
                    else:
                        F_j = torch.autograd.grad(E_mol, R, grad_outputs=self._cotangent(E_mol, 0),
                                                  create_graph=graph)[0]
        if late is not None:
            plan.join_late()
        return E_mol, F_j

    def _cotangent(self, E_mol, target):
        """Constant -1 in column `target` (zeros elsewhere) shaped like E_mol, cached per (shape, device)."""
        key = (tuple(E_mol.shape), E_mol.device, E_mol.dtype, target)
        c = self._cot.get(key) if self._cot is not None else None
        if c is None:
            c = torch.zeros(E_mol.shape, device=E_mol.device, dtype=E_mol.dtype)
            c[:, target] = -1.0
            if E_mol.is_cuda and torch.cuda.is_current_stream_capturing():
                return c      # memory of a capture's private pool must not outlive it in a cache
            # never evicted: a hipGraph captured after this call bakes the address in and reads it at every replay
            # (one (n_molecules, n_targets) tensor per distinct batch size: bytes)
            if self._cot is None:
                self._cot = {}
            self._cot[key] = c
        return c

    def _side_stream(self, device, role="out"):
        """One side stream per calling stream (several molecule shards may run this module concurrently) and role
        ("out": the output blocks; "plan": the adjoint-only index structures of a plan built inside a capture)."""
        if self._side is None:
            self._side = {}
        cur = torch.cuda.current_stream(device)
        key = (device.index, cur.cuda_stream) if role == "out" else (device.index, cur.cuda_stream, role)
        st = self._side.get(key)
        if st is None:
            st = torch.cuda.Stream(device=device)
            # torch hands out streams round-robin from a pool of 32: late in a long-lived process the new "side" stream
            # can BE the calling stream (e.g. the capture stream of torch.cuda.graph).  Everything would still be correct
            # (one stream, serial), but the output blocks would then count as same-stream consumers of the gradient sinks
            # and change the summation order — graph replay and eager run would stop being bit-identical.
            taken = {cur.cuda_stream} | {v.cuda_stream for k, v in self._side.items() if k[:2] == key[:2]}
            for _ in range(4):
                if st.cuda_stream not in taken:
                    break
                st = torch.cuda.Stream(device=device)
            self._side[key] = st
        return st

    def train(self, mode=True):
        if bool(mode) != self.training:
            self._wcache = {}   # weights change while training: derived forms cached for inference are stale afterwards
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._wcache = {}  # .to()/.float()/.cuda() replace the parameters' storage
        self._packs.clear()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._wcache = {}
        self._packs.clear()
        return super().load_state_dict(*args, **kwargs)

    def __deepcopy__(self, memo):
        side, self._side = self._side, None  # HIP stream handles are not copyable
        cache, self._wcache = self._wcache, {}  # keyed by this instance's weight addresses
        packs, self._packs = self._packs, ops.PackRegistry()
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            import copy
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
            return new
        finally:
            self._side = side
            self._wcache = cache
            self._packs = packs

    @staticmethod
    def _check_inputs(R):
        if not R.is_cuda:
            raise RuntimeError("gemnet_pytorch_amd.GemNet runs on a HIP device only (no CPU fallback); "
                               "move the model and the batch to 'cuda'.")
        if R.dtype != torch.float32:
            raise TypeError("gemnet_pytorch_amd.GemNet computes in fp32; got R of dtype %s" % R.dtype)

    # ----------------------------------------------------------------------------------- misc
    def predict(self, inputs):
        """(E, F) detached on the host, as the reference (gemnet.py:780-784).  Inputs from `md.DeviceMolecule.get()` (the MD
        loop of ase_calculator.py:148-170) carry no index arrays: they are served by the device index builder + one
        replayed hipGraph (md.predict_molecule)."""
        from ..md import MoleculeInputs, predict_molecule
        if isinstance(inputs, MoleculeInputs):
            return predict_molecule(self, inputs, to_host=True)
        E, F = self(inputs)
        return E.detach().cpu(), F.detach().cpu()

    def load_weights(self, path):
        self.load_state_dict(torch.load(path))

    def save_weights(self, path):
        torch.save(self.state_dict(), path)

    def tf_variable_names(self):
        """parameter name -> variable of the TensorFlow GemNet checkpoint (the copy list of gemnet.py:633-778 as rules;
        pinned against the reference in tests/golden/tf_names.json): '.' -> '/', Dense `weight` -> `kernel`, the
        ResidualLayer members `dense_mlp.<k>` -> `dense_mlp/layer_with_weights-<k>`, the embedding table without leaf."""
        out = {}
        for name, _ in self.named_parameters():
            parts = name.split(".")
            if name == "atom_emb.embeddings.weight":
                parts = parts[:-1]
            elif parts[-1] == "weight":
                parts[-1] = "kernel"
            parts = [f"layer_with_weights-{q}" if i > 0 and parts[i - 1] == "dense_mlp" else q
                     for i, q in enumerate(parts)]
            out[name] = "/".join(parts) + "/.ATTRIBUTES/VARIABLE_VALUE"
        return out

    def load_tfmodel(self, path):
        """Import the weights of a TensorFlow GemNet checkpoint (gemnet.py:617-778): 2-D kernels are transposed
        (TF Dense is (in, out)), the 3-D bilinear kernels and all other variables copied as they are.  `path` is a TF
        checkpoint prefix (needs `tensorflow`, as in the reference) or a `.npz` holding the same variables under the
        same names (`np.savez(f, **{n: reader.get_tensor(n) for n in names})` on a machine that has TensorFlow).
        Unlike the reference — which raises AttributeError on `out_forces/bias` of its bias-free Dense — direct-force
        models load as well."""
        import numpy as np
        if str(path).endswith(".npz"):
            arrays = np.load(path)
            get = lambda n: arrays[n]   # noqa: E731
        else:
            try:
                import tensorflow as tf
            except ImportError as e:
                raise ImportError(
                    "GemNet.load_tfmodel needs TensorFlow to read a TF checkpoint (as the reference does); without it, "
                    "export the checkpoint's variables to a .npz with the same names and pass that file") from e
            get = tf.train.load_checkpoint(path).get_tensor
        with torch.no_grad():
            for name, p in self.named_parameters():
                tf_name = self.tf_variable_names_cached()[name]
                W = torch.as_tensor(np.asarray(get(tf_name)))
                if tf_name.endswith("kernel/.ATTRIBUTES/VARIABLE_VALUE") and W.dim() == 2:
                    W = W.t()
                if tuple(W.shape) != tuple(p.shape):
                    raise ValueError(f"checkpoint variable {tf_name}: shape {tuple(W.shape)} does not fit "
                                     f"parameter {name} {tuple(p.shape)}")
                p.copy_(W.to(p.dtype))
        self._wcache.clear()

    def tf_variable_names_cached(self):
        names = getattr(self, "_tf_names", None)
        if names is None:
            names = self._tf_names = self.tf_variable_names()
        return names
