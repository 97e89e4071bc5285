"""Static-shape inference runtime: hipGraphs of forward+force, replayed concurrently on dedicated HIP streams.

At the headline batch (32 molecules x 32 atoms: 18 k edges) every kernel of the path is latency-bound — a
128x128 layer over 18 k rows is ~4 us of MFMA work behind ~4 us of launch/drain — so a single in-order
stream leaves the chip partly idle between kernels.  Molecules are independent (block-diagonal index
arrays), so the batch is cut into sub-batches, each captured ONCE into its own hipGraph
(`torch.cuda.CUDAGraph`, including the autograd pass that yields F = -dE/dR) on its own stream; a step
launches all graphs back-to-back and the hardware interleaves their kernels.  No nested capture, no
collective, no host synchronisation inside a step.

The graphs read the model parameters and the `R` tensors of the sub-batches in place: `set_positions`
copies new coordinates into the captured buffers (MD with a fixed neighbour list); a changed graph
(new neighbour list) needs a new runner.
"""
import os
import warnings

import torch


class RangeFlag:
    """The range check of REPLAYED graphs (include/gemnet_hip.h, gn_nonfinite_flag_f32).

    The default Dense arithmetic ("h3") keeps activations in two fp16 planes: beyond 65 504 they become inf, which propagates
    to the energies and forces; the reference's fp32 (base_layers.py:44-48) has no such cliff.  An eager forward reads its
    outputs back (GemNet.forward); a captured hipGraph — the MD loop, padded batches, the training step: the paths built for
    long unattended runs — cannot.  A graph captured with a RangeFlag ends with one tiny launch per output that ORs a bit
    into ONE device word when the output holds an inf / NaN, and with a copy of that word into pinned host memory; the host
    reads the pinned copy without synchronising (`tripped()`: what the COMPLETED replays left there — exact right after
    anything that waited for the replay, e.g. the `.cpu()` of `GemNet.predict`).  The word is sticky until `reset()`.
    Bits: OUTPUT = non-finite energies / forces of a forward; GRAD = a non-finite gradient norm (the fused optimizer skipped
    that step, csrc/optim.hip); bits 8+ of the word count the steps the fused optimizer skipped since the last reset.
    With several ranks (`snapshot` / `poll_lagged`) the word is OR-reduced over the ranks at a fixed point of every step and
    read from the mirror slot of the step BEFORE the previous one, behind that step's event: every rank then sees the same
    word at the same call, whatever its host is ahead of its GPU by."""
    OUTPUT, GRAD = 1, 2

    def __init__(self, device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory() if torch.device(device).type == "cuda" \
            else torch.zeros(1, dtype=torch.int32)
        self.trips = 0
        self._slots = None          # multi-rank polling: two pinned mirror slots + their events, by step parity
        self._step = 0

    def watch(self, *tensors, bit=OUTPUT):
        """Enqueue (or capture) the check of `tensors` (contiguous fp32) and the mirror copy on the current stream."""
        from . import kernels as K
        for t in tensors:
            K.nonfinite_flag(t, self.word, bit)
        self.mirror()

    def mirror(self):
        self.host.copy_(self.word, non_blocking=True)

    def tripped(self):
        return int(self.host[0]) & 0xff

    def skipped(self):
        """Steps the fused optimizer skipped on the device since the last reset (as of the last completed mirror copy)."""
        return int(self.host[0]) >> 8

    def reset(self):
        self.word.zero_()
        self.host.zero_()
        if self._slots is not None:
            for h, _ in self._slots:
                h.zero_()

    # ---- several ranks: one decision for all of them -------------------------------------------------------------------
    def snapshot(self, group=None):
        """End of a step (every rank, same point): OR the word over the ranks on the device (the collective is enqueued like the
        gradient all-reduce: no host synchronisation) and mirror it into the pinned slot of this step's parity, behind an event."""
        import torch.distributed as dist
        if self._slots is None:
            cuda = self.word.is_cuda
            self._slots = [((torch.zeros(1, dtype=torch.int32).pin_memory() if cuda else torch.zeros(1, dtype=torch.int32)),
                            torch.cuda.Event() if cuda else None) for _ in range(2)]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.word, op=dist.ReduceOp.BOR, group=group)
        host, ev = self._slots[self._step & 1]
        host.copy_(self.word, non_blocking=True)
        if ev is not None:
            ev.record()
        self.host.copy_(self.word, non_blocking=True)
        self._step += 1

    def poll_lagged(self):
        """Start of a step: the word as the step before the previous one left it (waits for THAT step's event — free when the
        host runs at most one step ahead of the device).  The same value on every rank at the same call."""
        if self._slots is None or self._step < 2:
            return 0
        host, ev = self._slots[self._step & 1]          # parity of step - 2
        if ev is not None:
            ev.synchronize()
        return int(host[0])


def fall_back_to_bf16_planes(model, what, positions=None):
    """A replayed graph of `model` produced non-finite values in the fp16-plane arithmetic: warn, move THIS model to the
    bf16-plane form ("split6": fp32 exponent range) and drop everything derived from its weights in the old format.  The
    caller captures anew.  False when the model was not on the fp16 planes (the values are non-finite for another reason:
    nothing to fall back to)."""
    from . import kernels as K
    mode = getattr(model, "matmul_precision", None) or K.DEFAULT_CHAIN_MODE
    if positions is not None:
        bad = [p for p in (positions if isinstance(positions, (list, tuple)) else [positions])
               if torch.is_tensor(p) and not bool(torch.isfinite(p).all())]
        if bad:
            # NaN / inf handed in by the caller is not an overflow of the arithmetic: the model stays where it is
            warnings.warn(f"gemnet_pytorch_amd: non-finite values in {what}: the POSITIONS handed in are not finite "
                          "(the arithmetic is left unchanged)", RuntimeWarning)
            return False
    if mode != "h3":
        warnings.warn(f"gemnet_pytorch_amd: non-finite values in {what} (arithmetic {mode!r}: no fp16 range limit involved)",
                      RuntimeWarning)
        return False
    warnings.warn(f"gemnet_pytorch_amd: non-finite values in {what} — the fp16-plane Dense arithmetic ('h3') overflows beyond "
                  "65504 (are the scale factors fitted?); this model now uses matmul_precision = 'split6' (bf16 planes, fp32 "
                  "range) and the graph is captured again", RuntimeWarning)
    model.matmul_precision = "split6"
    model._wcache = {}
    if getattr(model, "_packs", None) is not None:
        model._packs.clear()
    return True


class ForceGraphs:
    def __init__(self, model, batches, warmup=2):
        """`batches`: list of input dicts (reference keys, already on the HIP device), one per sub-batch."""
        if not batches:
            raise ValueError("at least one sub-batch")
        self.model = model
        # Private position buffers: autograd ties a leaf's gradient accumulator to the stream of its first backward
        # and keeps it on the tensor; positions that already went through an eager forward on the DEFAULT stream make
        # the capture below crash inside hipStreamEndCapture (the engine synchronises the capturing stream with the
        # default stream).  Fresh leaves see their first backward on this runner's own streams.
        self.batches = [dict(b, R=b["R"].detach().clone()) for b in batches]
        for b in self.batches:
            b.pop("_plan", None)
        dev = self.batches[0]["R"].device
        if dev.type != "cuda":
            raise RuntimeError("ForceGraphs needs a HIP device (no CPU fallback)")
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.batches]
        self.graphs = []
        self.outputs = []
        self.flag = RangeFlag(dev)
        self._capture(warmup)

    def _capture(self, warmup=2):
        model, dev = self.model, self.batches[0]["R"].device
        self.graphs, self.outputs = [], []
        model.eval()
        cur = torch.cuda.current_stream(dev)
        for inputs, st in zip(self.batches, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(warmup):
                    model(inputs)
            cur.wait_stream(st)
        torch.cuda.synchronize(dev)
        for inputs, st in zip(self.batches, self.streams):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                out = model(dict(inputs, _range_flag=self.flag))
            self.graphs.append(g)
            self.outputs.append(out)
        torch.cuda.synchronize(dev)

    def set_positions(self, i, R):
        self.batches[i]["R"].detach().copy_(R)

    def replay(self):
        """Enqueue one step (all sub-batches); the calling stream waits for all of them.  A non-finite result of an EARLIER
        replay (RangeFlag, polled without synchronising) makes the model fall back to the bf16 planes and the graphs are
        captured again before this step runs."""
        if self.flag.tripped():
            self.flag.trips += 1
            torch.cuda.synchronize()
            self.flag.reset()
            if fall_back_to_bf16_planes(self.model, "a replayed forward+force graph (ForceGraphs)",
                                        positions=[b["R"] for b in self.batches]):
                self._capture()
        cur = torch.cuda.current_stream()
        for g, st in zip(self.graphs, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.replay()
        for st in self.streams:
            cur.wait_stream(st)
        return self.outputs

    __call__ = replay

    def energies_forces(self):
        """Concatenated (E (nMol, T), F (nAtoms, ...)) of the last replay, sub-batch order."""
        return (torch.cat([o[0] for o in self.outputs]), torch.cat([o[1] for o in self.outputs]))


# The index build inside the replayed graph (padded.PaddedGraphRunner.attach_builder: GemNet-T and GemNet-Q).
IN_GRAPH_INDEX = os.environ.get("GEMNET_INDEX_IN_GRAPH", "1") == "1"


class DynamicForceField:
    """Energies and forces of ONE system for positions that change from call to call — the loop of the reference's ASE
    calculator (ase_calculator.py:148-170: new neighbour list every step) — at hipGraph speed:

        ff = DynamicForceField(model, Z, N_host, cutoff=5.0, int_cutoff=10.0)
        E, F = ff(R)            # R (A, 3) float32 on the device

    Every call builds the index arrays on the device (index_device.DeviceGraphBuilder, on a stream of its own), pads them to
    the current capacities and replays the one captured graph (padded.PaddedGraphRunner: bit-identical to the eager run
    on the unpadded arrays).  The capacities start `margin` above the first call's sizes; a call that outgrows them
    captures a new graph with `margin` head room again (counted in `recaptures`).  Triplets-only AND quadruplet models (the
    latter since round 5: padded.py pads interaction edges, intermediate triplets and quadruplets as well — GemNet-Q is the
    model of the reference's MD example, ase_example.ipynb cell 13)."""

    def __init__(self, model, Z, N_host, cutoff, int_cutoff, margin=0.08, max_in_degree=None):
        from .index_device import DeviceGraphBuilder
        import numpy as np
        self.model, self.Z = model, Z
        self.N_host = np.asarray(N_host, dtype=np.int64).reshape(-1)
        self.N = torch.as_tensor(self.N_host, device=Z.device)
        self.builder = DeviceGraphBuilder(self.N_host, cutoff, int_cutoff, model.triplets_only, device=Z.device)
        self.margin = float(margin)
        self.deg = int(max_in_degree) if max_in_degree is not None else int(self.N_host.max()) - 1
        self.runner = None
        self.recaptures = 0
        self._bstream = None

    def _build(self, R):
        main = torch.cuda.current_stream(R.device)
        if self._bstream is None:
            self._bstream = torch.cuda.Stream(device=R.device)
        self._bstream.wait_stream(main)          # the new positions come from work on the calling stream (the integrator)
        with torch.cuda.stream(self._bstream):
            idx = self.builder(R, dtype=torch.int32)     # as built: the padded runner keeps its index buffers in int32
        main.wait_stream(self._bstream)
        for t in idx.values():
            t.record_stream(main)
        return idx

    def __call__(self, R, exact=True):
        """`exact` (the graph builds its own neighbour list): wait for the step and look at the
        device-side report of its index build — a system that outgrew the capacities is re-sized and the step repeated, as
        on the host-sized path.  With exact=False nothing waits: such a step returns NaN energies / forces and the NEXT call
        re-sizes (for callers that keep everything on the device and check `index_failed()` themselves)."""
        from .padded import PaddedGraphRunner
        r = self.runner
        if r is not None and r.builder is not None:
            if not r.index_error():
                out = r.run_positions(R)            # the whole step — index build included — is one replay
                if not exact:
                    return out
                torch.cuda.current_stream(R.device).synchronize()
                if not r.index_error():
                    return out
            r.reset_index_state()                   # this / an earlier step outgrew the capacities: size them anew below
        idx = self._build(R)
        sizes = PaddedGraphRunner.sizes_of(idx)
        r = self.runner
        # pad triplets need a complete quad (unit) of pad edges; the pad edges must fit the dummy groups' in-degree bound
        if r is None or not r.fits(sizes):
            E, T = sizes[:2]
            m = self.margin
            e_cap = int(E * (1 + 1.5 * m)) // 12 * 12 + 24
            t_cap = int(T * (1 + m)) // 2 * 2 + 2
            # dummy groups for the largest padding this runner may see (a later call with fewer edges pads more)
            groups = max(1, -(-int(e_cap * min(1.0, 4 * m)) // (2 * max(self.deg, 2))))
            quad_caps = None
            if not self.model.triplets_only:
                Eint, I, Q = sizes[2:5]
                quad_caps = (int(Eint * (1 + 1.5 * m)) + 4, int(I * (1 + m)) + 4, int(Q * (1 + m)) + 4)
            self.runner = PaddedGraphRunner(self.model, self.Z, self.N, e_cap, t_cap, max_in_degree=self.deg, n_groups=groups,
                                            quad_caps=quad_caps)
            self.recaptures += self.runner is not r and r is not None
        if IN_GRAPH_INDEX and R.dtype == torch.float32:
            # from here on the index build is part of the graph (padded.attach_builder): this call's
            # arrays validate the buffers, every later call is  positions in -> one replay -> results out
            self.runner._fill(R, idx)
            if self.runner.builder is None:
                self.runner.attach_builder(self.builder)
            return self.runner.run_positions(R)
        return self.runner(R, idx)

    def index_failed(self):
        """Did a completed step's in-graph index build outgrow the capacities (its outputs are NaN)?  Exact after the caller
        waited for that step."""
        return self.runner is not None and self.runner.builder is not None and bool(self.runner.index_error())

    def range_tripped(self):
        """RangeFlag of the current runner (exact after the caller waited for the last replay)."""
        return self.runner is not None and self.runner.flag.tripped()

    def recover(self, R):
        """After `range_tripped()`: fall back to the bf16 planes, capture anew, repeat the step."""
        self.runner.recover()
        return self(R)
