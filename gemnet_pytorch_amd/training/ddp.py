"""Data-parallel training step: one process per GPU, molecule shards, ONE RCCL all-reduce per step.

The reference is single-process; this is the build's addition named by the
north star.  Molecules are independent (all index arrays are block-diagonal per molecule), so each
rank owns a shard of the global batch with rank-local atom/edge offsets and no halo.  The loss of
the reference trainer (gemnet/training/trainer.py:284-292,338-343: (1-rho) * MAE(E) + rho *
mean_atoms ||F - F_t||_2) is a mean over molecules plus a mean over atoms, so every rank scales its
local sums by the GLOBAL counts and the gradients are SUMMED:

    loss_r = (1-rho) * sum_mol |E - E_t| / B_global + rho * sum_atom ||F - F_t|| / A_global
    grad   = all_reduce_sum(grad_r)          == gradient of the single-process loss on the union

The whole model (1.9 M parameters, 7.6 MB fp32) is ONE contiguous gradient buffer -> one
collective over xGMI, issued after `loss.backward()` and before the shared-gradient rescale
(trainer.py:250-278) and the global-norm clip (trainer.py:353-356) so that both act on global
gradients exactly like the single-process trainer.
"""
import numpy as np
import os

import torch
import torch.distributed as dist


def partition_molecules(costs, world_size):
    """Greedy longest-processing-time partition of molecule ids into `world_size` shards balanced by
    `costs` (number of triplets for GemNet-T, quadruplets for GemNet-Q: fan-out varies ~ n*deg^3)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    load = np.zeros(world_size)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += costs[i]
    return [sorted(s) for s in shards]


class FlatGradBuffer:
    """All trainable parameters' .grad as views into one contiguous fp32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, device=p0.device, dtype=p0.dtype)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, always=False):
        """Sum over the ranks; `always`: issue the collective even in a one-rank group (exercises RCCL on one GPU)."""
        if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size() > 1):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


def scale_shared_grads(model):
    """Counterpart of Trainer.scale_shared_grads (trainer.py:250-278)."""
    shared = [model.mlp_rbf3, model.mlp_cbf3, model.mlp_rbf_h]
    if not model.triplets_only:
        shared += [model.mlp_rbf4, model.mlp_cbf4, model.mlp_sbf4]
    with torch.no_grad():
        for layer in shared:
            if layer.weight.grad is not None:
                layer.weight.grad.div_(model.num_blocks)
        if model.mlp_rbf_out.weight.grad is not None:
            model.mlp_rbf_out.weight.grad.div_(model.num_blocks + 1)


def make_optimizer(model, learning_rate=1e-3, weight_decay=2e-6):
    """AdamW on the weights, Adam (no decay) on atom_emb / frequencies / bias (trainer.py:115-160),
    amsgrad, eps=1e-7 as in the reference."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if any(s in name for s in ("atom_emb", "frequencies", "bias")) else decay).append(p)
    return torch.optim.AdamW(
        [dict(params=decay, weight_decay=weight_decay), dict(params=no_decay, weight_decay=0.0)],
        lr=learning_rate, eps=1e-7, amsgrad=True)


# The loss (trainer.py:330-343) and its cotangents as ONE launch on the device; GEMNET_FUSED_LOSS=0: the ATen composite.
USE_FUSED_LOSS = os.environ.get("GEMNET_FUSED_LOSS", "1") == "1"


class _ForceLoss(torch.autograd.Function):
    """loss = w_e sum|E - Et| + w_f [* w_f_dev] sum_a mask_a |F_a - Ft_a|_2 (kernels.force_loss); the weights carry rho and the
    global molecule / atom counts.  First-order backward only: the training step differentiates the loss once (its second
    order is the force's, inside the model)."""

    @staticmethod
    def forward(ctx, E, F, Et, Ft, w_e, w_f, mask, w_f_dev):
        from .. import kernels as K
        loss, gE, gF = K.force_loss(E, Et, F, Ft, w_e, w_f, mask=mask, w_f_dev=w_f_dev)
        ctx.save_for_backward(gE, gF)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gE, gF = ctx.saved_tensors
        return gE * g, gF * g, None, None, None, None, None, None


class TrainStep:
    """fwd + force + loss + backward + (all-reduce) + shared-grad rescale + clip + optimizer step."""

    def __init__(self, model, world_size=1, rho_force=0.999, grad_clip_max=10.0, optimizer=None,
                 global_counts=None, fused_optimizer=False):
        self.model = model
        self.world_size = world_size
        self.rho = rho_force
        self.clip = grad_clip_max
        self.fused = None
        if fused_optimizer:
            # rescale + clip + AdamW + EMA in two launches over the flat buffer (training/fused_optim.py)
            from .fused_optim import FusedAdamWEMA
            self.fused = FusedAdamWEMA(model, grad_clip_max=grad_clip_max)
            self.buf = FlatGradBuffer.__new__(FlatGradBuffer)
            self.buf.params, self.buf.flat = self.fused.params, self.fused.flat_g
            self.opt = None
        else:
            self.buf = FlatGradBuffer(model.parameters())
            self.opt = optimizer if optimizer is not None else make_optimizer(model)
        self.global_counts = global_counts  # (B_global, A_global) if known statically
        self.last_loss = None
        self.always_reduce = False   # issue the gradient all-reduce even when the process group has one rank
        self._pinned_counts = None   # ((B_local, A_local), (B_global, A_global)) of the captured static batch
        self._use_pinned = False     # True only inside capture(): the graph replays with the counts it was captured with
        self.wgrad = None
        self.flag = None
        if self.buf.params[0].is_cuda:
            from .wgrad_queue import WeightGradQueue
            self.wgrad = WeightGradQueue()  # all weight-gradient GEMMs of the final backward as one grouped launch
            # device-side range check of the step (runtime.RangeFlag): the forward's energies / forces inside the captured
            # graph, the gradient norm inside the fused optimizer (which skips a non-finite step); polled at the next call
            from ..runtime import RangeFlag
            self.flag = RangeFlag(self.buf.params[0].device)

    def _counts(self, n_mol, n_atoms, device):
        """(B_global, A_global).  Batches of a real loader vary in atom count and the last one is partial
        (drop_last=False), so the counts are exchanged EVERY step unless the caller fixed them (`global_counts=`).
        The counts pinned by `capture()` are used ONLY while that batch is being captured (the graph must not hold a
        collective): whether a rank skips the exchange has to be the same decision on every rank, and "my local shape
        equals the captured one" is not — one rank could match while another holds a partial last batch, and its
        2-element all-reduce would then pair with the first rank's gradient all-reduce."""
        if self.global_counts is not None:
            return self.global_counts
        if self._use_pinned and self._pinned_counts is not None:
            assert self._pinned_counts[0] == (n_mol, n_atoms), "captured step replayed on a different batch"
            return self._pinned_counts[1]
        if self.world_size > 1:
            t = torch.tensor([n_mol, n_atoms], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return float(t[0]), float(t[1])
        return float(n_mol), float(n_atoms)

    def loss(self, E, F, targets):
        B, A = self._counts(E.shape[0], F.shape[0], E.device)
        if USE_FUSED_LOSS and E.is_cuda and E.dtype == torch.float32:
            # one launch for the loss and its cotangents (csrc/optim.hip: force_loss_kernel) instead of 16 + 16 ATen nodes
            return _ForceLoss.apply(E, F, targets["E"], targets["F"], (1 - self.rho) / (B * E.shape[1]), self.rho / A, None, None)
        e_term = (E - targets["E"]).abs().sum() / (B * E.shape[1])
        f_term = torch.norm(F - targets["F"], p=2, dim=1).sum() / A
        return (1 - self.rho) * e_term + self.rho * f_term

    @staticmethod
    def _local_counts(inputs):
        """(molecules, atoms) of this rank's batch that enter the loss normalisation."""
        return int(inputs["N"].shape[0]), int(inputs["Z"].shape[0])

    def _outputs(self, inputs):
        """(E (n_mol, targets), F (n_atoms, 3)) of the batch — a hook for subclasses that run a padded batch."""
        if self.flag is None:
            E, F = self.model(inputs)
        else:
            # the flag rides in the dict for the duration of the call only: the caller's dict is left without the non-tensor
            # entry (Trainer.dict2device, torch.save and a later eval forward of the same dict see what they handed in)
            inputs["_range_flag"] = self.flag
            try:
                E, F = self.model(inputs)
            finally:
                inputs.pop("_range_flag", None)
        if F.dim() == 3:
            F = F[:, 0]
        return E, F

    def _forward_backward(self, inputs, targets):
        E, F = self._outputs(inputs)
        loss = self.loss(E, F, targets)
        self.buf.zero()
        # restrict the double backward to the parameters: no gradient w.r.t. the positions
        from .. import ops
        if self.wgrad is None:
            with ops.exclusive():
                torch.autograd.backward(loss, inputs=self.buf.params)
        else:
            with ops.exclusive(), ops.wgrad_queue(self.wgrad), ops.position_second_order_grads(False):
                torch.autograd.backward(loss, inputs=self.buf.params)
            self.wgrad.flush()
        return loss.detach()

    def capture(self, inputs, targets, check=False, _reuse_counts=False):
        """Capture forward + force + loss + double backward (thousands of small launches) into one
        hipGraph for this (static-shape) batch; the collective, clipping and optimizer stay eager.
        The graph reads the parameters in place, so optimizer updates are seen by every replay.
        `check`: record the capture with the happens-before checker (hbcheck.py; the recorder is left in `self.hb`)."""
        self.model.train()
        if "id_c" not in inputs:
            raise ValueError("capture() needs a batch with its index arrays: inputs = model.with_indices(inputs)")
        local = self._local_counts(inputs)
        if not (_reuse_counts and self._pinned_counts is not None and self._pinned_counts[0] == local):
            # (a recapture after a range fall-back keeps the counts of the same static batch: no collective on that path)
            self._pinned_counts = None
            self._use_pinned = False
            self._pinned_counts = (local, self._counts(*local, inputs["Z"].device))   # no collective inside the graph
        self._use_pinned = True
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._forward_backward(inputs, targets)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                if check:
                    from .. import hbcheck
                    with hbcheck.record(keep=check == "keep") as self.hb:
                        self._graph_loss = self._forward_backward(inputs, targets)
                else:
                    self._graph_loss = self._forward_backward(inputs, targets)
        finally:
            self._use_pinned = False
        self._graph_key = id(inputs)
        self._cap_args = (inputs, targets)
        return self

    def _range_check(self, agreed=False):
        """Poll the range flag.  Tripped: the fused optimizer has skipped the non-finite step(s) on the device; warn, move the
        model off the fp16 planes and capture the step anew.  One rank: what the completed steps left in the pinned word (no
        synchronisation).  Several ranks: the OR of all ranks' words as of the step before the previous one
        (RangeFlag.snapshot / poll_lagged) — every rank takes this branch at the SAME call, so the recapture (and whatever
        collectives the steps around it issue) pair up, and no rank runs a step in another arithmetic than its peers.
        `agreed`: the caller has already made the decision collective and written it into every rank's pinned word
        (Trainer._train_on_batch_padded synchronises and exchanges the flag BEFORE its optimizers see the gradients)."""
        if self.flag is None:
            return False
        if self.world_size > 1 and not agreed:
            word = self.flag.poll_lagged()
            if not (word & 0xff):
                return False
            torch.cuda.synchronize()
            # the word as it stands now: every rank has enqueued the same steps, each step OR-reduced it over the ranks, and the
            # flags are sticky / the skip count monotone — a superset of what the lagged poll saw, identical on every rank
            word = int(self.flag.word.item())
            bits, skipped = word & 0xff, word >> 8
        else:
            if not self.flag.tripped():
                return False
            torch.cuda.synchronize()
            bits, skipped = self.flag.tripped(), self.flag.skipped()
        from ..runtime import fall_back_to_bf16_planes
        self.flag.trips += 1
        self.flag.reset()
        if self.fused is not None and bits & self.flag.GRAD:
            # exactly the steps the device skipped (counted next to the flag, csrc/optim.hip): Adam's bias-correction counter
            self.fused.steps = max(0, self.fused.steps - max(1, skipped))
        cap = getattr(self, "_cap_args", None)
        if fall_back_to_bf16_planes(self.model, "a training step (%s)" % ("gradient norm" if bits & self.flag.GRAD else
                                                                          "energies / forces"),
                                    positions=cap[0]["R"] if cap is not None else None):
            if getattr(self, "_graph", None) is not None:
                self._recapture()
            return True
        return False

    def _recapture(self):
        inputs, targets = self._cap_args
        inputs.pop("_plan", None)
        self._graph = None
        self.capture(inputs, targets, _reuse_counts=True)

    def _eager(self, inputs, targets):
        """Eager forward/backward on a private stream: autograd ties every parameter's AccumulateGrad node to the
        stream of its first backward, and nodes created on the DEFAULT stream make a later hipGraph capture of the
        step fail inside hipStreamEndCapture (the engine then synchronises the capturing stream with the default
        stream).  The caller's stream waits for the side stream, so the semantics are unchanged."""
        dev_is_cuda = self.buf.params[0].is_cuda
        if not dev_is_cuda or torch.cuda.is_current_stream_capturing():
            return self._forward_backward(inputs, targets)
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream(device=self.buf.params[0].device)
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            loss = self._forward_backward(inputs, targets)
        cur.wait_stream(self._stream)
        return loss

    def __call__(self, inputs, targets, step_optimizer=True):
        self.model.train()
        self._range_check()
        if getattr(self, "_graph", None) is not None and self._graph_key == id(inputs):
            self._graph.replay()
            loss = self._graph_loss
        else:
            loss = self._eager(inputs, targets)
        self.buf.all_reduce(always=self.always_reduce)
        if self.fused is not None:
            if step_optimizer:
                self.fused.step(flag=self.flag)
        else:
            scale_shared_grads(self.model)
            norm = torch.nn.utils.clip_grad_norm_(self.buf.params, max_norm=self.clip)
            if step_optimizer:
                # an eager optimizer has no device-side skip: the gradient norm is read here (one synchronisation of this —
                # already host-bound — path) so that a non-finite step never reaches the parameters; the flag's GRAD bit makes
                # the next call fall back exactly as after a skipped fused step.  (All ranks hold the same reduced gradient.)
                if self.flag is not None and not bool(torch.isfinite(norm)):
                    self.flag.word.bitwise_or_(torch.tensor(self.flag.GRAD, dtype=torch.int32, device=self.flag.word.device))
                    self.flag.mirror()
                else:
                    self.opt.step()
        if self.flag is not None and self.world_size > 1:
            self.flag.snapshot()         # every rank, every step, at this fixed point: the ranks decide together
        self.last_loss = loss
        return self.last_loss


class PaddedTrainStep(TrainStep):
    """The training step for batches whose array sizes change every step — what a real training loop sees
    (data_provider.py:159-165) — replayed from ONE captured hipGraph: every batch is padded to fixed capacities with the
    dummy molecule of `padded.py`, whose energy and forces never reach the loss (its rows then receive zero cotangents:
    no contribution to any parameter gradient).  The number of molecules per batch is fixed; with `a_cap` the molecules
    may differ in size from batch to batch (pass Z and N with every step), without it the layout N is fixed.

        ts = PaddedTrainStep(model, Z, N, e_cap, t_cap, a_cap=..., fused_optimizer=True)
        loss = ts.step(R, idx, E_target, F_target, Z=Z_batch, N=N_batch)

    The loss is the one of `TrainStep.loss` written with a mask over the atom capacity (a slice would bake the first
    batch's atom count into the graph); the atom count of the step (all ranks) is a device scalar filled before the
    replay.  On a CPU model (host emulation of the launchers: tests) the padded batch runs eagerly."""

    def __init__(self, model, Z, N, e_cap, t_cap, max_in_degree=None, n_groups=None, a_cap=None, **kw):
        super().__init__(model, **kw)
        from ..padded import PaddedGraphRunner
        self.pad = PaddedGraphRunner(model, Z, N, e_cap, t_cap, max_in_degree=max_in_degree, n_groups=n_groups, a_cap=a_cap)
        dev = Z.device
        self.inputs = dict(self.pad.inputs, max_in_degree=self.pad.pad_degree_bound())
        dt = next(model.parameters()).dtype
        self.targets = {"E": torch.zeros(self.pad.n_mol, model.num_targets, device=dev, dtype=dt),
                        "F": torch.zeros(self.pad.a_cap, 3, device=dev, dtype=dt)}
        self.mask = torch.zeros(self.pad.a_cap, device=dev, dtype=dt)        # 1 for the atoms of the current batch
        self.inv_atoms = torch.ones((), device=dev, dtype=dt)                # 1 / (atoms of the step, all ranks)
        self._captured = False

    def _local_counts(self, inputs):
        return self.pad.n_mol, self.pad.a_cap

    def _recapture(self):
        self._captured = False      # (the capture run rebuilds the index plan inside the graph: see _outputs)
        try:
            super()._recapture()
        finally:
            self._captured = True

    def _outputs(self, inputs):
        if not self._captured:
            # the index plan must be built INSIDE the captured region (from the static buffers, so that every replay
            # rebuilds it for the batch just written): no plan cached by an earlier eager call may survive into it.
            # The plan of the capture run itself stays in the dict — its tensors are the graph's memory.
            inputs.pop("_plan", None)
        # range guard of the fp16-plane arithmetic: the real molecules' rows (up to the atom CAPACITY: a later batch may hold
        # more atoms than the one the graph is captured on; filler atoms are isolated and finite) — not the dummy molecule's
        inputs["_guard_rows"] = (self.pad.n_mol, self.pad.a_cap)
        E, F = super()._outputs(inputs)
        return E[:self.pad.n_mol], F[:self.pad.a_cap]

    def loss(self, E, F, targets):
        B = float(self.pad.n_mol * max(self.world_size, 1))
        if USE_FUSED_LOSS and E.is_cuda and E.dtype == torch.float32:
            return _ForceLoss.apply(E, F, targets["E"], targets["F"], (1 - self.rho) / (B * E.shape[1]), self.rho, self.mask,
                                    self.inv_atoms)
        e_term = (E - targets["E"]).abs().sum() / (B * E.shape[1])
        m = self.mask
        # masked rows get a constant difference: the norm of an exact zero has no gradient (0 * nan), and they have no share
        d = torch.where(m[:, None] > 0, F - targets["F"], torch.ones_like(F))
        f_term = (torch.norm(d, p=2, dim=1) * m).sum() * self.inv_atoms
        return (1 - self.rho) * e_term + self.rho * f_term

    def step(self, R, idx, E_target, F_target, Z=None, N=None, step_optimizer=True):
        self.pad._fill(R, idx, Z, N)
        A = self.pad.A
        self.targets["E"].copy_(E_target.reshape(self.targets["E"].shape))
        self.targets["F"][:A].copy_(F_target)
        self.targets["F"][A:].zero_()
        self.mask[:A].fill_(1.0)
        self.mask[A:].zero_()
        if self.world_size > 1:
            cnt = torch.full((), float(A), device=self.mask.device, dtype=torch.float64)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)          # device-side: no host sync, before the replay
            self.inv_atoms.copy_(1.0 / cnt)
        else:
            self.inv_atoms.fill_(1.0 / A)
        if R.is_cuda and not self._captured:
            self.capture(self.inputs, self.targets)
            self._captured = True
        return self(self.inputs, self.targets, step_optimizer=step_optimizer)
