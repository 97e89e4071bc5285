"""`Trainer` — counterpart of the reference's gemnet/training/trainer.py:9-520 with the call surface used by
train.ipynb / train_seml.py / fit_scaling.py (H2 of SURVEY.md §8):

    train_on_batch / test_on_batch / eval_on_batch / predict_on_batch / predict / dict2device,
    save_variable_backups / load_averaged_variables / restore_variable_backups / decay_maybe,
    schedulers[i].get_last_lr(), optimizers, tracked_metrics, state_dict / load_state_dict.

Same numerics as the reference: loss = (1-rho) * MAE(E) + rho * {MAE | mean L2}(F) (trainer.py:280-343) or
the Gaussian NLL pair for mean-variance estimation; AdamW (weights) + Adam (atom_emb / frequencies / bias),
amsgrad, eps 1e-7 (:115-160); shared-gradient rescale (:250-278); global-norm or adaptive gradient clipping
(:218-248,349-356); linear-warmup exponential decay + reduce-on-plateau; EMA of the parameters.

Native differences: all gradients live in ONE flat buffer (`FlatGradBuffer`), so in a multi-process run
(one process per GPU, `torch.distributed` initialised) the step issues exactly one RCCL all-reduce of the
gradients, with the loss terms normalised by the GLOBAL molecule / atom counts (see training/ddp.py) and the
returned / logged loss all-reduced to the global value (every rank must therefore run the same number of
`train_on_batch` / `test_on_batch` calls — shard the validation set evenly or pad it); `loss.backward()` is
restricted to the parameters (no gradient w.r.t. positions); `load_state_dict` works (upstream's iterates a
bound method, trainer.py:508).
"""
import torch
import torch.distributed as dist

from .. import ops
from .ddp import FlatGradBuffer, scale_shared_grads
from .ema_decay import ExponentialMovingAverage
from .schedules import LinearWarmupExponentialDecay, ReduceLROnPlateau


class MultiWrapper:
    """Treat several optimizers / schedules as one (zero_grad, step, state_dict, indexing)."""

    def __init__(self, *ops):
        self.wrapped = ops

    def __getitem__(self, idx):
        return self.wrapped[idx]

    def zero_grad(self):
        for op in self.wrapped:
            op.zero_grad()

    def step(self):
        for op in self.wrapped:
            op.step()

    def state_dict(self):
        return {i: op.state_dict() for i, op in enumerate(self.wrapped)}

    def load_state_dict(self, state_dict):
        for i, op in enumerate(self.wrapped):
            op.load_state_dict(state_dict[i])


class Trainer:
    _SUBSTATE = ("schedulers", "optimizers", "plateau_callback", "exp_decay")
    # per-process execution state: never part of a checkpoint (the padded step holds the model, lambdas and a hipGraph)
    _RUNTIME_ONLY = ("model", "params_except_last", "_grads", "_wgrad", "_pstep", "_padded_caps")

    def __init__(self, model, learning_rate: float = 1e-3, decay_steps: int = 100000, decay_rate: float = 0.96,
                 warmup_steps: int = 0, weight_decay: float = 0.001, staircase: bool = False,
                 grad_clip_max: float = 1000, decay_patience: int = 10, decay_factor: float = 0.5,
                 decay_cooldown: int = 10, ema_decay: float = 0.999, rho_force: float = 0.99, loss: str = "mae",
                 mve: bool = False, agc=False):
        assert 0 <= rho_force <= 1
        self.model = model
        self.ema_decay = ema_decay
        self.grad_clip_max = grad_clip_max
        self.rho_force = float(rho_force)
        self.mve = mve
        self.loss = loss
        self.agc = agc
        self.tracked_metrics = (["loss", "energy_mae", "energy_nll", "energy_var", "force_mae", "force_rmse",
                                 "force_nll", "force_var"] if mve
                                else ["loss", "energy_mae", "force_mae", "force_rmse"])
        self._grads = None
        self._wgrad = None
        self._padded_caps = None     # enable_padded_graph(): capacities of the captured training step
        self._pstep = None
        self.reset_optimizer(learning_rate, weight_decay, warmup_steps, decay_steps, decay_rate, staircase,
                             decay_patience, decay_factor, decay_cooldown)

    # ------------------------------------------------------------------------------------ optimiser
    def reset_optimizer(self, learning_rate, weight_decay, warmup_steps, decay_steps, decay_rate, staircase,
                        decay_patience, decay_factor, decay_cooldown):
        adam_kw = dict(lr=learning_rate, betas=(0.9, 0.999), eps=1e-07, amsgrad=True)
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        if weight_decay > 0:
            plain = lambda n: any(s in n for s in ("atom_emb", "frequencies", "bias"))
            opts = [torch.optim.AdamW([p for n, p in named if not plain(n)], weight_decay=weight_decay, **adam_kw),
                    torch.optim.Adam([p for n, p in named if plain(n)], **adam_kw)]
        else:
            opts = [torch.optim.Adam(self.model.parameters(), **adam_kw)]
        self.optimizers = MultiWrapper(*opts)
        self.schedulers = MultiWrapper(*[LinearWarmupExponentialDecay(o, warmup_steps, decay_steps, decay_rate,
                                                                      staircase) for o in opts])
        self.plateau_callback = ReduceLROnPlateau(optimizer=self.optimizers, scheduler=self.schedulers,
                                                  factor=decay_factor, patience=decay_patience,
                                                  cooldown=decay_cooldown, verbose=True)
        if self.agc:  # (the reference's selection, trainer.py:193-198: the output heads)
            self.params_except_last = [p for n, p in named if "out_energy" in n or "out_forces" in n]
        self.exp_decay = ExponentialMovingAverage([p for _, p in named], self.ema_decay)

    def save_variable_backups(self):
        self.exp_decay.store()

    def load_averaged_variables(self):
        self.exp_decay.copy_to()

    def restore_variable_backups(self):
        self.exp_decay.restore()

    def decay_maybe(self, val_loss):
        self.plateau_callback.step(val_loss)

    # -------------------------------------------------------------------------------------- gradients
    @staticmethod
    def _unitwise_norm(x, norm_type=2.0):
        if x.ndim <= 1:
            return x.norm(norm_type)
        return x.norm(norm_type, dim=tuple(range(1, x.ndim)), keepdim=True)

    @staticmethod
    def _adaptive_gradient_clipping(parameters, clip_factor=0.05, eps=1e-3, norm_type=2.0):
        """Unit-wise adaptive gradient clipping (Brock et al., 2021): |g_unit| <= clip_factor * max(|w_unit|, eps)."""
        with torch.no_grad():
            for p in ([parameters] if isinstance(parameters, torch.Tensor) else parameters):
                if p.grad is None:
                    continue
                max_norm = Trainer._unitwise_norm(p, norm_type).clamp_(min=eps).mul_(clip_factor)
                grad_norm = Trainer._unitwise_norm(p.grad, norm_type)
                clipped = p.grad * (max_norm / grad_norm.clamp(min=1e-6))
                p.grad.copy_(torch.where(grad_norm < max_norm, p.grad, clipped))

    def scale_shared_grads(self):
        scale_shared_grads(self.model)

    # ----------------------------------------------------------------------------------------- losses
    @staticmethod
    def get_mae(targets, pred):
        return torch.nn.functional.l1_loss(pred, targets, reduction="mean")

    @staticmethod
    def get_rmse(targets, pred):
        return torch.mean(torch.norm(pred - targets, p=2, dim=1))

    @staticmethod
    def get_nll(targets, mean_pred, var_pred):
        return torch.nn.functional.gaussian_nll_loss(mean_pred, targets, var_pred, reduction="mean")

    def predict(self, inputs):
        energy, forces = self.model(inputs)
        if self.mve:
            return (energy[:, :1], torch.nn.functional.softplus(energy[:, 1:]),
                    forces[:, 0, :], torch.nn.functional.softplus(forces[:, 1, :]))
        if forces.dim() == 3:
            forces = forces[:, 0]
        return energy, None, forces, None

    @staticmethod
    def dict2device(data, device=None):
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        for key in data:
            data[key] = data[key].to(device)
        return data

    def predict_on_batch(self, dataset_iter):
        inputs, _ = next(dataset_iter)
        return self.predict(self.dict2device(inputs))

    def _world(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _global_weights(self, n_mol, n_atoms, device, dtype):
        """(B_r / B, A_r / A): this rank's share of the global molecule / atom counts (one small all-reduce)."""
        counts = torch.tensor([n_mol, n_atoms], dtype=torch.float64, device=device)
        local = counts.clone()
        dist.all_reduce(counts)
        return (local / counts).to(dtype)

    def _objective(self, targets, mean_energy, var_energy, mean_forces, var_forces):
        """-> (objective to differentiate on this rank, value to report, dict of metric tensors).

        Single process: both are the reference's loss (trainer.py:284-343).  Multi-process (one rank per GPU): every
        term is a mean over the rank's molecules / atoms, so the rank's objective weights it by its share of the
        GLOBAL counts (B_r/B, A_r/A) and the SUM of the ranks' gradients is the gradient of the loss on the union
        batch — for the MAE/RMSE terms and for the Gaussian NLL terms alike.  The REPORTED value is the all-reduced
        sum of those weighted objectives, i.e. the global loss, identical on every rank (so `decay_maybe(val_loss)`
        takes the same decision everywhere)."""
        out = {}
        if self.mve:
            out["energy_nll"] = e_term = self.get_nll(targets["E"], mean_energy, var_energy)
            out["force_nll"] = f_term = self.get_nll(targets["F"], mean_forces, var_forces)
        else:
            out["energy_mae"] = e_term = self.get_mae(targets["E"], mean_energy)
            f_term = self.get_mae(targets["F"], mean_forces) if self.loss == "mae" \
                else self.get_rmse(targets["F"], mean_forces)
            out["force_mae" if self.loss == "mae" else "force_rmse"] = f_term
        if self._world() == 1:
            loss = e_term * (1 - self.rho_force) + self.rho_force * f_term
            return loss, loss.detach(), out
        w = self._global_weights(mean_energy.shape[0], mean_forces.shape[0], mean_energy.device, e_term.dtype)
        loss = e_term * w[0] * (1 - self.rho_force) + self.rho_force * f_term * w[1]
        report = loss.detach().clone()
        dist.all_reduce(report)
        return loss, report, out

    # ------------------------------------------------------------------------------------------ steps
    def enable_padded_graph(self, a_cap, e_cap, t_cap, max_in_degree, n_groups=None, n_mol=None):
        """Run `train_on_batch` from ONE captured hipGraph although every batch has its own array sizes: each batch is
        padded to these capacities (atoms, edges, triplets) with the dummy molecule of `padded.py` and the captured
        forward + force + loss + backward is replayed; all-reduce, shared-gradient rescale, clipping, the optimizers,
        schedulers, EMA and the metrics run as before.  Same loss values as the plain step (the objective is written
        with a mask over the atom capacity); triplets-only models with forces by autograd, MAE / RMSE objectives (not the
        mean-variance NLL).  Batches with the loader's number of molecules that fit the capacities take the graph; any other
        batch (the partial last one) takes the plain step.  `max_in_degree`: largest molecule size - 1; `n_mol`: the loader's
        batch size (default: that of the first batch that fits)."""
        if self.mve:
            raise NotImplementedError("padded graph: MAE / RMSE objectives")
        self._padded_caps = dict(a_cap=int(a_cap), e_cap=int(e_cap), t_cap=int(t_cap), max_in_degree=int(max_in_degree),
                                 n_groups=n_groups, n_mol=None if n_mol is None else int(n_mol))
        self._pstep = None

    def _padded_step_for(self, inputs):
        """The padded step if this batch can take it, else None."""
        caps = self._padded_caps
        if caps is None or not self.model.triplets_only or self.model.direct_forces:
            return None
        n_mol, A = int(inputs["N"].shape[0]), int(inputs["Z"].shape[0])
        E, T = int(inputs["id_c"].shape[0]), int(inputs["id3_reduce_ca"].shape[0])
        if self._pstep is None:
            # the first batch that takes the graph fixes the number of molecules: with `n_mol` given, a partial batch
            # that happens to come first does not (every later full batch would fall back to the plain step)
            ok = A <= caps["a_cap"] and (caps.get("n_mol") is None or n_mol == caps["n_mol"])
            if not self._all_ranks(ok):
                return None
            # the captured step writes the gradients into ITS flat buffer: from here on that buffer is the Trainer's (a plain
            # step taken earlier — a first batch that did not fit — used one of its own; `.grad` is re-pointed)
            self._pstep = _TrainerPaddedStep(self, inputs["Z"], inputs["N"],
                                             **{k: v for k, v in caps.items() if k != "n_mol"})
            self._grads = self._pstep.buf
        ps = self._pstep
        pad = ps.pad
        fits = (n_mol == pad.n_mol and A <= pad.a_cap and E + 4 <= pad.e_cap and T <= pad.t_cap
                and -(-((pad.e_cap - E) // 2) // pad.G) <= pad.pad_degree_bound())
        return ps if self._all_ranks(fits) else None

    def _all_ranks(self, flag):
        """`flag` on every rank?  The padded and the plain step issue different collectives (the plain objective
        exchanges counts), so the choice between them has to be the same everywhere."""
        if self._world() == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=self._grads.params[0].device if self._grads is not None
                         else next(self.model.parameters()).device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0)

    def _train_on_batch_padded(self, ps, inputs, targets, metrics):
        loss = ps.run(inputs, targets)
        if ps.flag is not None:
            # the replayed graph cannot read its energies / forces back: its device-side range check (runtime.RangeFlag) is
            # read here, BEFORE the optimizers see the gradients — an overflow of the fp16-plane arithmetic moves the model to
            # the bf16 planes (with a warning), the step is captured anew and repeated (all ranks decide together)
            torch.cuda.current_stream().synchronize()
            if not self._all_ranks(not ps.flag.tripped()):
                ps.flag.host.fill_(ps.flag.tripped() | ps.flag.OUTPUT)      # (a rank that did not trip follows the others)
                if ps._range_check(agreed=True):
                    loss = ps.run(inputs, targets)
        report = loss.detach().clone()
        if self._world() > 1:
            dist.all_reduce(report)
        self._grads.all_reduce()
        self.scale_shared_grads()
        if self.agc:
            self._adaptive_gradient_clipping(self.params_except_last, clip_factor=self.grad_clip_max)
        else:
            torch.nn.utils.clip_grad_norm_(self._grads.params, max_norm=self.grad_clip_max)
        self.optimizers.step()
        self.schedulers.step()
        self.exp_decay.update()
        with torch.no_grad():
            A = int(inputs["Z"].shape[0])
            E_pred, F_pred = ps.last_out[0].detach(), ps.last_out[1].detach()[:A]
            e_term, f_term = ps.parts
            parts = {"energy_mae": e_term, ("force_mae" if self.loss == "mae" else "force_rmse"): f_term}
            self._update_metrics(metrics, report, targets, E_pred, None, F_pred, None, parts)
        return report

    def train_on_batch(self, dataset_iter, metrics):
        self.model.train()
        inputs, targets = next(dataset_iter)
        inputs, targets = self.dict2device(inputs), self.dict2device(targets)
        inputs = self.model.with_indices(inputs)      # (DataContainer(indices="device"): the index arrays are built here, on the GPU)
        ps = self._padded_step_for(inputs)
        if ps is not None:
            return self._train_on_batch_padded(ps, inputs, targets, metrics)
        mean_energy, var_energy, mean_forces, var_forces = self.predict(inputs)
        loss, report, parts = self._objective(targets, mean_energy, var_energy, mean_forces, var_forces)

        if self._grads is None:
            self._grads = FlatGradBuffer(self.model.parameters())
            self._wgrad = None
            if self._grads.params[0].is_cuda:
                from .wgrad_queue import WeightGradQueue
                self._wgrad = WeightGradQueue()   # all weight-gradient products of the step as one grouped launch
        self._grads.zero()
        # gradients w.r.t. the parameters only: the second-order POSITION terms of the fused geometry ops are not needed
        if self._wgrad is None:
            with ops.exclusive(), ops.position_second_order_grads(False):
                torch.autograd.backward(loss, inputs=self._grads.params)
        else:
            with ops.exclusive(), ops.wgrad_queue(self._wgrad), ops.position_second_order_grads(False):
                torch.autograd.backward(loss, inputs=self._grads.params)
            self._wgrad.flush()
        self._grads.all_reduce()  # ONE collective (no-op in a single process)
        self.scale_shared_grads()
        if self.agc:
            self._adaptive_gradient_clipping(self.params_except_last, clip_factor=self.grad_clip_max)
        else:
            torch.nn.utils.clip_grad_norm_(self._grads.params, max_norm=self.grad_clip_max)
        self.optimizers.step()
        self.schedulers.step()
        self.exp_decay.update()

        with torch.no_grad():
            self._update_metrics(metrics, report, targets, mean_energy, var_energy, mean_forces, var_forces, parts)
        return report

    def _update_metrics(self, metrics, loss, targets, mean_energy, var_energy, mean_forces, var_forces, parts):
        energy_mae = parts.get("energy_mae")
        if energy_mae is None:
            energy_mae = self.get_mae(targets["E"], mean_energy)
        force_mae = parts.get("force_mae")
        if force_mae is None:
            force_mae = self.get_mae(targets["F"], mean_forces)
        force_rmse = parts.get("force_rmse")
        if force_rmse is None:
            force_rmse = self.get_rmse(targets["F"], mean_forces)
        if self.mve:
            metrics.update_state(nsamples=mean_energy.shape[0], loss=loss, energy_mae=energy_mae,
                                 energy_nll=parts["energy_nll"], energy_var=var_energy)
            metrics.update_state(nsamples=mean_forces.shape[0], force_mae=force_mae, force_rmse=force_rmse,
                                 force_nll=parts["force_nll"], force_var=var_forces)
        else:
            metrics.update_state(nsamples=mean_energy.shape[0], loss=loss, energy_mae=energy_mae)
            metrics.update_state(nsamples=mean_forces.shape[0], force_mae=force_mae, force_rmse=force_rmse)

    def test_on_batch(self, dataset_iter, metrics):
        self.model.eval()
        inputs, targets = next(dataset_iter)
        inputs, targets = self.dict2device(inputs), self.dict2device(targets)
        if self.model.direct_forces:
            with torch.no_grad():
                outs = self.predict(inputs)
        else:
            outs = self.predict(inputs)  # forces need dE/dR (first-order, fused path in eval mode)
        mean_energy, var_energy, mean_forces, var_forces = (None if o is None else o.detach() for o in outs)
        with torch.no_grad():
            _, report, parts = self._objective(targets, mean_energy, var_energy, mean_forces, var_forces)
            self._update_metrics(metrics, report, targets, mean_energy, var_energy, mean_forces, var_forces, parts)
        return report

    def eval_on_batch(self, dataset_iter):
        self.model.eval()
        inputs, targets = next(dataset_iter)
        inputs, targets = self.dict2device(inputs), self.dict2device(targets)
        energy, _, forces, _ = self.predict(inputs)
        return (energy.detach(), forces.detach()), targets

    # ------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        skip = self._RUNTIME_ONLY + self._SUBSTATE
        state = {k: v for k, v in self.__dict__.items() if k not in skip}
        state.update({attr: getattr(self, attr).state_dict() for attr in self._SUBSTATE})
        return state

    def load_state_dict(self, state_dict):
        for k, v in state_dict.items():
            if k in self._SUBSTATE:
                getattr(self, k).load_state_dict(v)
            elif k not in self._RUNTIME_ONLY:   # a checkpoint written before these were excluded may carry them
                setattr(self, k, v)


def _padded_step_base():
    from .ddp import PaddedTrainStep
    return PaddedTrainStep


class _TrainerPaddedStep(_padded_step_base()):
    """`PaddedTrainStep` carrying the Trainer's objective (trainer.py:284-343 of the reference: (1 - rho) MAE(E) + rho
    {MAE | mean L2}(F), every term a mean over this rank's molecules / atoms times its share of the global counts) in
    masked form; only the captured part (forward, force, loss, backward) is used — the Trainer runs the rest of its step."""

    def __init__(self, trainer, Z, N, a_cap, e_cap, t_cap, max_in_degree, n_groups=None):
        super().__init__(trainer.model, Z, N, e_cap, t_cap, max_in_degree=max_in_degree, n_groups=n_groups, a_cap=a_cap,
                         world_size=trainer._world(), rho_force=trainer.rho_force, optimizer=trainer.optimizers)
        self.trainer = trainer
        # `loss` below carries its own per-step shares: the base class's count exchange (a collective + host sync inside
        # `capture`, which ranks may reach on different steps) must never run
        self.global_counts = (1, 1)
        dev, dt = self.mask.device, self.mask.dtype
        self.share = torch.ones(2, device=dev, dtype=dt)          # (B_r / B, A_r / A) of this rank in this step
        self.inv_local = torch.ones((), device=dev, dtype=dt)     # 1 / A_r
        self.parts = self.last_out = None

    def _outputs(self, inputs):
        self.last_out = super()._outputs(inputs)
        return self.last_out

    def loss(self, E, F, targets):
        tr = self.trainer
        m = self.mask
        e_term = (E - targets["E"]).abs().mean()
        diff = F - targets["F"]
        if tr.loss == "mae":
            f_term = (diff.abs() * m[:, None]).sum() * (self.inv_local / 3.0)
        else:
            d = torch.where(m[:, None] > 0, diff, torch.ones_like(diff))
            f_term = (torch.norm(d, p=2, dim=1) * m).sum() * self.inv_local
        self.parts = (e_term.detach(), f_term.detach())
        return e_term * self.share[0] * (1 - tr.rho_force) + tr.rho_force * f_term * self.share[1]

    def run(self, inputs, targets):
        """Pad the batch into the static buffers and run / replay forward + force + loss + backward -> loss (this rank's
        weighted objective; gradients in the flat buffer)."""
        idx = {k: inputs[k] for k in ("id_c", "id_a", "id_swap", "id_undir", "id3_reduce_ca", "id3_expand_ba")}
        self.pad._fill(inputs["R"], idx, inputs["Z"], inputs["N"])
        A = self.pad.A
        self.targets["E"].copy_(targets["E"].reshape(self.targets["E"].shape))
        self.targets["F"][:A].copy_(targets["F"])
        self.targets["F"][A:].zero_()
        self.mask[:A].fill_(1.0)
        self.mask[A:].zero_()
        self.inv_local.fill_(1.0 / A)
        if self.world_size > 1:
            cnt = torch.tensor([self.pad.n_mol, A], dtype=torch.float64, device=self.mask.device)
            tot = cnt.clone()
            dist.all_reduce(tot)
            self.share.copy_(cnt / tot)
        self.model.train()
        if inputs["R"].is_cuda:
            if not self._captured:
                self.capture(self.inputs, self.targets)
                self._captured = True
            self._graph.replay()
            return self._graph_loss
        return self._eager(self.inputs, self.targets)
