#!/usr/bin/env python
"""bench.py — molecules/s (forward + force) of GemNet-T on COLL-shaped batches, N x MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one pass of the hot path over one batch:
GemNet-T full (4 blocks, emb 128; pretrained/GemNet-T/model_kwargs.json of the reference) on a
synthetic batch of 32 molecules x 32 atoms per GPU (BASELINE.json configs[1]), forward + force
(F = -dE/dR), fp32, inputs resident in HBM.  Weak scaling: every rank owns its own batch of
`--batch` molecules; no data-path collective is needed for forward+force (molecules are independent).

  --mode train   times the full training step instead (fwd + force + loss.backward + RCCL gradient
                 all-reduce + AdamW), reported under the same JSON keys with metric suffix.

Besides the headline, the same JSON line carries `extra` (never part of `value`):
  extra.train_step                 the full training step (with N > 1 GPUs: including the RCCL gradient all-reduce)
  extra.interaction_block_fwd_bwd  ONE InteractionBlock forward+backward, isolated: step/s and roofline fractions
  extra.gemnet_q                   BASELINE configs[2]: GemNet-Q forward+force with its own roofline, and extra.gemnet_q.train_step:
                                   its training step (eager) with the roofline of the dominant library family
  extra.dynamic_shape              a new batch every step: device index build + plan + eager forward+force, and the same
                                   loop padded to fixed capacities and replayed from ONE hipGraph (padded.py)
  extra.train_step_dynamic         the training step on a new batch every step: eager and padded-capacity hipGraph
  extra.config4_shard              BASELINE configs[4], one GPU's shard: GemNet-Q, 64 molecules x 64 atoms, default and
                                   peak memory, roofline of the dominant family (--no-config4 skips)

Extra objects on the JSON line:
  roofline      dominant kernel family of the step, measured live with HIP events around every launch
                in one instrumented (eager) pass over the same batch
  cpu_baseline  the oracle (plain-PyTorch CPU restatement, kind "port") timed on the host cores,
                rank 0 / N=1 only, on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GEMNET_T = dict(num_spherical=7, num_radial=6, num_blocks=4, emb_size_atom=128, emb_size_edge=128,
                emb_size_trip=64, emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32,
                emb_size_bil_trip=64, emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1,
                num_concat=1, num_atom=2, triplets_only=True, num_targets=1, direct_forces=False,
                cutoff=5.0, int_cutoff=10.0, envelope_exponent=5, extensive=True, forces_coupled=False,
                output_init="HeOrthogonal", activation="swish")
SCALE_FILE = os.path.join(ROOT, "gemnet_pytorch_amd", "scaling_factors.json")
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def shard_ids(world, rank, batch, n_atoms, cutoff=5.0):
    """Molecule numbers of this rank's shard: the global batch of world x batch synthetic molecules partitioned by
    `partition_molecules` (longest-processing-time on the per-molecule triplet count), as the data-parallel trainer
    does it.  One rank: the first `batch` molecules."""
    if world == 1:
        return list(range(batch))
    from gemnet_pytorch_amd.synthetic import make_molecule, triplet_count
    from gemnet_pytorch_amd.training.ddp import partition_molecules
    costs = [triplet_count(make_molecule(n_atoms, 1000 * 2 + i)["R"], cutoff) for i in range(world * batch)]
    return partition_molecules(costs, world)[rank]


def make_batch(cfg, n_mol, n_atoms, first, device, ids=None):
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.data_container import DataContainer
    if ids is not None:
        n_mol = len(ids)
    ds = make_dataset(n_mol, n_atoms, config=2, first=first, ids=ids)
    dc = DataContainer.from_arrays(ds, cfg["cutoff"], cfg["int_cutoff"], triplets_only=cfg["triplets_only"])
    batch = dc[list(range(n_mol))]
    targets = {k: batch.pop(k).to(device) for k in ("E", "F")}
    inputs = {k: v.to(device) for k, v in batch.items()}
    return inputs, targets


class LaunchTimer:
    """Records every launch of one step (launcher, arguments, algorithmic flops/bytes: DESIGN.md §kernels);
    `summary()` then times each launcher family by replaying exactly those launches back-to-back between two
    HIP events on the stream the kernels are launched on (= torch's current stream)."""

    def __init__(self):
        from gemnet_pytorch_amd import kernels as K
        self.K = K
        self.records = []
        self.saved = {}

    @staticmethod
    def cost(name, args, kwargs, out):
        f32 = 4
        if name == "chain":
            prog = args[0]
            fl = by = 0.0
            for o in prog.ops:
                if o["kind"] == "gemm":
                    N, Kd = o["W"].shape
                    fl += 2.0 * prog.M * N * Kd
                    by += N * Kd * f32
                    for k in ("pre_out", "out", "gadd1", "gadd2", "out2"):
                        if torch.is_tensor(o.get(k)):
                            by += prog.M * N * f32
                    for k in ("mul", "res", "res2", "Z2"):
                        if torch.is_tensor(o.get(k)):
                            by += prog.M * N * f32
                elif o["kind"] == "load":
                    by += prog.M * o["src"].shape[1] * f32 * (2 if torch.is_tensor(o.get("Z2")) else 1)
                elif o["kind"] == "store":
                    by += prog.M * o["out"].shape[1] * f32
                else:
                    for k in ("Z", "out"):
                        if torch.is_tensor(o.get(k)):
                            by += o[k].numel() * f32
            return fl, by
        if name == "gemm_tn":
            A, B = args[0], args[1]
            Kd, M, N = A.shape[0], A.shape[1], B.shape[1]
            return 2.0 * M * N * Kd, (M * Kd + N * Kd + M * N) * f32
        if name == "gemm":
            A, B = args[0], args[1]
            ta = args[2] if len(args) > 2 else kwargs.get("trans_a", False)
            o = out[0] if isinstance(out, tuple) else out
            M, N = o.shape
            Kd = A.shape[0] if ta else A.shape[1]
            by = (M * Kd + N * Kd + M * N * (2 if isinstance(out, tuple) else 1)) * f32
            for k in ("mul", "res", "a_dact_pre"):
                if kwargs.get(k) is not None:
                    by += kwargs[k].numel() * f32
            return 2.0 * M * N * Kd, by
        if name == "bmm":
            o = out
            b, m, n = o.shape
            k = args[0].numel() // (b * m)
            return 2.0 * b * m * n * k, (args[0].numel() + args[1].numel() + o.numel()) * f32
        if name == "bil_dy_multi":
            sp, nb = args[2], len(args[0])
            S, C = args[0][0].shape[1], args[0][0].shape[2]
            ang = kwargs.get("ang") is not None
            fl = 2.0 * nb * sp.size * S * C + (2.0 * 600 * sp.size if ang else 0.0)   # + Y_lm and both angle derivatives
            by = nb * (sp.n_reduce * S * C + sp.n_expand * C) * f32 + sp.size * ((32 if ang else S * f32) + 4)
            return fl, by
        if name == "bil_reduce_project_tan":      # S3 of the quadruplet layer: (ang, tang, x, tx, B, tB, Sm, sp)
            ang, tang, x, tx, B, tB, Sm, sp = args[:8]
            E, S, I = sp.n_reduce, B.shape[1], B.shape[2]
            C = x.shape[1]
            nt = (tang is not None) + (tx is not None)
            fl = 2.0 * nt * sp.size * S * C + 2.0 * E * S * I * C * (2 if tB is not None else 1) + 2.0 * 400 * sp.size
            by = sp.size * (32 + 8) + nt * sp.n_expand * C * f32 + E * (S * C + S * I + I * C) * f32
            if tB is not None:
                by += E * (S * I + S * C) * f32
            return fl, by
        if name == "bil_reduce_t_tan":            # x-adjoint of S4: (ang, tang, D1, D2, sp)
            ang, tang, D1, D2, sp = args[:5]
            E, S, C = D2.shape
            nd = 2 if D1 is not None else 1
            fl = 2.0 * nd * sp.size * S * C + 2.0 * 400 * sp.size
            by = sp.size * (32 + 8) + nd * E * S * C * f32 + sp.n_expand * C * f32
            return fl, by
        if name == "bil_fused_fwd":
            Y, x, B, W2T, sp = args[:5]
            S, C, I, O = Y.shape[1], x.shape[1], B.shape[2], W2T.shape[0]
            E = sp.n_reduce
            fl = 2.0 * sp.size * S * C + 2.0 * E * S * I * C + 2.0 * E * I * C * O
            by = sp.size * (S * f32 + 8) + (sp.n_expand * C + E * (S * C + S * I + O) + W2T.numel()) * f32
            return fl, by
        if name == "bil_fused_bwd":
            g, W2, Sm, B = args[:4]
            E, S, C = Sm.shape
            I, O = B.shape[2], g.shape[1]
            fl = 2.0 * E * I * C * O + 4.0 * E * S * I * C
            by = (g.numel() + W2.numel() + 2 * Sm.numel() + 2 * B.numel()) * f32
            return fl, by
        if name == "segsum_multi":
            terms = args[0]
            return 0.0, (sum(t[0].numel() for t in terms) + out.numel()) * f32 + sum(t[0].shape[0] for t in terms) * 4
        if name in ("bil_reduce", "bil_reduce_t", "bil_dot", "bil_reduce_project", "bil_project_bwd"):
            sp = next(a for a in reversed(args) if hasattr(a, "n_reduce"))
            E = sp.n_reduce
            if name == "bil_project_bwd":
                S, C = args[1].shape[1], args[1].shape[2]
                I = args[2].shape[2]
                want_dY = kwargs.get("want_dY", True)
                # gB = Sm dP^T and dSm = B dP per edge; the per-triplet Y gradient only when it is not deferred to
                # bil_dy_multi (then nothing of size T or Q is touched by this launch)
                fl = 4.0 * E * S * I * C + (2.0 * sp.size * S * C if want_dY else 0.0)
                by = E * (I * C + 2 * S * C + 2 * S * I) * f32
                if want_dY:
                    by += sp.size * (S * f32 + 8) + sp.n_expand * C * f32
                return fl, by
            Y = args[0]
            ang = Y.dim() == 2 and Y.shape[1] == 4 and name != "bil_dot"
            S = 49 if ang else Y.shape[1]
            C = args[1].shape[-1]
            ybytes = 16 if ang else S * f32           # angle form: (sin, cos) of two angles per quadruplet
            by = sp.size * (ybytes + 8) + sp.n_expand * C * f32 + E * S * C * f32
            fl = 2.0 * sp.size * S * C
            if ang:
                fl += 2.0 * 200 * sp.size             # Y_lm rebuilt per quadruplet: ~200 f32 FMA (csrc/bilinear_ang.hip)
            if name == "bil_reduce_project":
                I = args[2].shape[2]
                fl += 2.0 * E * S * I * C
                by += E * (S * I + I * C) * f32
            return fl, by
        numel = sum(a.numel() for a in args if torch.is_tensor(a))
        # tensor keyword operands are READ as well (the running sums an accumulate-into launch adds to: acc_m / acc_rbf of the
        # aggregation adjoint — 9 of its 9 launches per forward+force step carry one)
        numel += sum(v.numel() for v in kwargs.values() if torch.is_tensor(v))
        outs = out if isinstance(out, tuple) else (out,)
        return 0.0, (numel + sum(o.numel() for o in outs if torch.is_tensor(o))) * f32

    FAMILIES = ["rbf_aggregate_fwd", "rbf_aggregate_bwd", "gather_mul", "dist_fwd", "dist_bwd", "dist_jvp", "angle_fwd", "angle_bwd",
                "angle_jvp", "pack_weight_split", "pack_weight_split_grouped",
                "gemm", "gemm_tn", "bmm", "gather", "segsum", "ssilu", "pm", "dact_mul", "chain", "bil_reduce",
                "bil_reduce_t", "bil_dot", "bil_reduce_project", "bil_fused_fwd", "bil_fused_bwd", "bil_project_bwd", "bil_dy_multi", "segsum_multi",
                "bessel_rbf", "sph_radial", "ylm0",
                "ylm", "edge_basis_fwd", "edge_basis_bwd", "trip_basis_fwd", "trip_basis_bwd", "quad_basis_fwd",
                "quad_basis_bwd", "quad_angles_fwd", "quad_angles_bwd", "quad_angles_jvp", "bil_reduce_project_tan",
                "bil_reduce_t_tan"]

    def __enter__(self):
        self.depth = 0
        for name in self.FAMILIES:
            fn = getattr(self.K, name)
            self.saved[name] = fn

            def wrapped(*args, _fn=fn, _name=name, **kwargs):
                if self.depth:  # a launcher calling another launcher (gemm -> gemm_tn): count once
                    return _fn(*args, **kwargs)
                self.depth += 1
                try:
                    out = _fn(*args, **kwargs)
                finally:
                    self.depth -= 1
                fl, by = self.cost(_name, args, kwargs, out)
                if _name == "chain":   # the replay runs outside the sweep's arithmetic context (ops_train._sweep_mode: thread-
                    # local, and this wrapper runs on the launching thread — the autograd engine's for the backward sweeps)
                    kwargs = dict(kwargs, mode=kwargs.get("mode") or self.K.current_mode())
                self.records.append((_name, _fn, args, kwargs, fl, by))
                return out

            setattr(self.K, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.K, name, fn)

    def summary(self, reps=5):
        """Per launcher family: the recorded launches of ONE step (same arguments, same order) are captured
        into a hipGraph and replayed back-to-back `reps` times between two HIP events on the launch stream,
        so the figure is kernel time without host dispatch gaps (what rocprofv3's average duration shows)."""
        torch.cuda.synchronize()
        fam = {}
        for name, fn, args, kwargs, fl, by in self.records:
            d = fam.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, calls=[], ang=False))
            if name.startswith("bil_"):
                Y0 = kwargs.get("ang") if name == "bil_dy_multi" else (args[0] if args and torch.is_tensor(args[0]) else None)
                d["ang"] = d["ang"] or (Y0 is not None and Y0.dim() == 2 and Y0.shape[1] == 4)
            d["flops"] += fl
            d["bytes"] += by
            d["launches"] += 1
            d["calls"].append((fn, args, kwargs))
        with torch.no_grad():
            for name, d in fam.items():
                calls = d.pop("calls")
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for fn, args, kwargs in calls:
                        fn(*args, **kwargs)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for fn, args, kwargs in calls:
                        fn(*args, **kwargs)
                g.replay()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    g.replay()
                e.record()
                torch.cuda.synchronize()
                d["ms"] = s.elapsed_time(e) / reps
                del g
        self.records = []
        return fam


PEAK_SPLIT6_TFLOPS = 2500.0 / 6   # fp32-equivalent ceiling of six bf16 products on the 2.5 PF dense bf16 pipe
PEAK_BF16_TFLOPS = 2500.0
PROFILE_ROUND = "r6"


def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_traffic(name, mode):
    """(HBM-side bytes per launch, source) of a launcher family from the COMMITTED rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of the same workload `mode` ("T" forward+force, "Q", "train"): profiles/<round>_traffic_<mode>.json,
    written by tools/gpu_artifacts.sh + tools/pmc_summary.py on the builder's GPU box (gfx950 FETCH_SIZE correction
    applied).  Not measured in this run: the counters need rocprofv3 around the process.  (None, None) when that
    family / mode was not profiled."""
    fn = f"{PROFILE_ROUND}_traffic_{mode}.json"
    tab = _profile_json(fn)
    try:
        return int(tab[name]["bytes_per_launch"]), f"profiles/{fn} (builder rocprofv3 --pmc pass, not this run)"
    except (TypeError, KeyError, ValueError):
        return None, None


def mfma_busy(name, mode):
    """MFMA-busy percentage of the family's kernels (SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x SIMDs)) from the
    committed SQ counter pass (profiles/<round>_mfma_busy.json); None when not profiled."""
    tab = _profile_json(f"{PROFILE_ROUND}_mfma_busy.json")
    try:
        return float(tab[mode][name]["mfma_busy_pct"])
    except (TypeError, KeyError, ValueError):
        return None


MFMA_FAMILIES = ("gemm", "gemm_tn", "bmm", "chain", "bil_fused_fwd", "bil_fused_bwd")


def family_bound(name, calls_ang):
    """Which roof bounds a launcher family: the matrix pipe, the f32 vector ALU (the angle-form tensor-basis kernels
    rebuild Y_lm per quadruplet: DESIGN.md section 8.4), or HBM."""
    if name in MFMA_FAMILIES:
        return "mfma"
    if (calls_ang and name in ("bil_reduce_project", "bil_reduce_t", "bil_dy_multi")) or name in ("bil_reduce_project_tan",
                                                                                                   "bil_reduce_t_tan"):
        return "valu"
    return "hbm"


def roofline_from(fam, mode="T"):
    from gemnet_pytorch_amd import kernels as K
    name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    sec = d["ms"] * 1e-3
    bound = family_bound(name, d.get("ang", False))
    traffic, src = pmc_traffic(name, mode)
    common = dict(traffic=traffic, traffic_source=src,
                  algorithmic_bytes_per_launch=int(d["bytes"] / d["launches"]),
                  avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2), launches=d["launches"],
                  share_of_kernel_time=round(d["ms"] / sum(v["ms"] for v in fam.values()), 3))
    if bound in ("mfma", "valu"):
        ach = d["flops"] / sec / 1e12
        out = dict(kernel=name, bound=bound, achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                   frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), **common)
        if bound == "mfma":
            out["mfma_busy_pct"] = mfma_busy(name, mode)
        if name == "chain" and K.DEFAULT_CHAIN_MODE != "f32":
            # the Dense stacks execute on the bf16 pipe: fraction of THAT pipe's ceiling for this arithmetic
            nprod = 3 if K.DEFAULT_CHAIN_MODE == "h3" else K.CHAIN_MODES[K.DEFAULT_CHAIN_MODE]
            out["executing_pipe"] = (f"{'f16' if K.DEFAULT_CHAIN_MODE == 'h3' else 'bf16'} MFMA, {nprod} product(s) per fp32 product"
                                     + ("; the loss-scaled sweeps S3 / S4 of the training step run 6 bf16 products"
                                        if K.DEFAULT_CHAIN_MODE == "h3" and mode == "train" else ""))
            out["frac_of_executing_pipe"] = round(ach * nprod / PEAK_BF16_TFLOPS, 4)
            out["peak_executing_pipe_fp32_equiv"] = round(PEAK_BF16_TFLOPS / nprod, 1)
        if bound == "valu":
            out["note"] = "f32 vector-ALU bound (256 FLOP/clk/CU = the f32 MFMA rate); flops include the in-kernel Y_lm rebuild"
        return out
    ach = d["bytes"] / sec / 1e9
    return dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                frac=round(ach / PEAK_HBM_GBS, 4), **common)


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(cfg, n_atoms, budget_s=14.0, n_mol=8, threads=None):
    """Oracle (CPU restatement, fp32, host cores) forward+force on a bounded sample.  `threads`: fixed, else the fastest of
    8 / 16 / 32 / 64 (capped by the host) on one timed step each — torch's CPU kernels on ~1e5-row operands stop scaling
    well before a 128-thread host is full, and oversubscribing SMT threads hurts."""
    from oracle import gemnet_oracle as GO
    inputs, _ = make_batch(cfg, n_mol, n_atoms, first=0, device="cpu")
    params = GO.make_params(cfg, 0, GO.load_scale_factors(SCALE_FILE), dtype=torch.float32)
    ncpu = os.cpu_count() or 1
    tried = {}
    if threads is None:
        for nthr in sorted({min(t, ncpu) for t in (8, 16, 32, 64)}):
            torch.set_num_threads(nthr)
            GO.forward(cfg, params, inputs)  # warm-up (page faults, thread pools)
            t1 = time.time()
            GO.forward(cfg, params, inputs)
            tried[nthr] = round(time.time() - t1, 3)
        threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    cores = threads
    GO.forward(cfg, params, inputs)
    t0 = time.time()
    steps = 0
    while True:
        GO.forward(cfg, params, inputs)
        steps += 1
        if time.time() - t0 > budget_s or steps >= 20:
            break
    dt = time.time() - t0
    return dict(value=round(n_mol * steps / dt, 3), unit="molecules/s", cores=cores, kind="port", cpu_model=cpu_model_name(),
                host_logical_cpus=ncpu, seconds_per_step_by_threads=tried or None,
                sample=f"{steps} steps of forward+force on {n_mol} molecules x {n_atoms} atoms "
                       f"(same generator/config as the GPU workload), torch CPU fp32, {cores} threads",
                ms_per_step=round(dt / steps * 1e3, 1),
                calibration="the build's own CPU restatement (oracle/, kind 'port'; its bilinear layer is the reference's padded "
                            "batched matmul, efficient.py:159-189).  Side by side in the build container (8 vCPU Intel Xeon @ 2.1 GHz, "
                            "8 threads, fp32, the 32 x 32-atom batch of configs[1]): the REFERENCE itself 1.3 s per forward+force = "
                            "24.6 molecules/s (SURVEY.md section 5), this port 1.04 s = 30.7 molecules/s; the reference's full "
                            "training step (fwd + force + loss.backward) 3.8 s = 8.4 molecules/s")


def cpu_baseline_train(cfg, n_atoms, n_mol=8, budget_s=10.0, threads=32):
    """The full training step of the oracle on the host cores (forward + force + loss + the second-order
    loss.backward(), trainer.py:325-360; no optimizer — negligible beside the double backward), bounded sample."""
    from oracle import gemnet_oracle as GO
    inputs, targets = make_batch(cfg, n_mol, n_atoms, first=0, device="cpu")
    params = GO.make_params(cfg, 0, GO.load_scale_factors(SCALE_FILE), dtype=torch.float32)
    leaves = [p.requires_grad_(True) for p in params.values() if p.dim() > 0]
    torch.set_num_threads(threads)

    def step():
        E, F = GO.forward(cfg, params, inputs, create_graph=True)
        loss = GO.training_loss(E[:, :1], F, targets["E"].reshape(-1, 1), targets["F"])
        torch.autograd.grad(loss, leaves, allow_unused=True)
    step()
    t0, steps = time.time(), 0
    while True:
        step()
        steps += 1
        if time.time() - t0 > budget_s or steps >= 10:
            break
    dt = time.time() - t0
    return dict(value=round(n_mol * steps / dt, 3), unit="molecules/s", cores=threads, kind="port",
                sample=f"{steps} training steps (forward + force + loss.backward through the force) on {n_mol} molecules x "
                       f"{n_atoms} atoms, torch CPU fp32, {threads} threads", ms_per_step=round(dt / steps * 1e3, 1))


# ------------------------------------------------------------------------------------------------ timing helpers
def capture(step, warm=2):
    """Warm `step` on a side stream and capture it into a hipGraph; returns (graph, outputs)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    return graph, out


def time_steps(run, steps, warmup, world=1):
    """`warmup` untimed + exactly `steps` timed calls between barrier + synchronize on both sides; max over ranks."""
    import torch.distributed as dist
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def family_roofline(step, mode="T"):
    """Instrument one eager pass of `step` and return (roofline of the dominant family, per-family table).
    `mode` names the workload ("T", "Q", "train") for the committed-counter lookups."""
    with LaunchTimer() as lt:
        step()
    fam = lt.summary()
    return roofline_from(fam, mode), fam


def log_families(title, fam):
    dump = os.environ.get("GEMNET_DUMP_FAMILIES")
    if dump:
        # (tools/gpu_artifacts.sh: the algorithmic bytes / flops per launch of every launcher family, merged into
        # profiles/<round>_traffic_<mode>.json next to the counted bytes by tools/pmc_summary.py)
        slug = "".join(c if c.isalnum() else "_" for c in title)
        with open(f"{dump}_{slug}.json", "w") as f:
            json.dump({k: dict(launches=v["launches"], algorithmic_bytes_per_launch=int(v["bytes"] / max(v["launches"], 1)),
                               algorithmic_flops_per_launch=int(v["flops"] / max(v["launches"], 1)), ms_per_step=round(v["ms"], 4))
                       for k, v in fam.items()}, f, indent=1)
    tot = sum(v["ms"] for v in fam.values())
    log(f"[bench] {title}: per-family kernel time of one step's launches, replayed back-to-back:")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:14]:
        log(f"    {k:16s} {v['ms']:8.3f} ms  {100 * v['ms'] / tot:5.1f} %  {v['launches']:5d} launches"
            f"  {v['flops'] / max(v['ms'], 1e-9) / 1e9:8.2f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s")


# ------------------------------------------------------------------------------------------------ extra measurements
def extra_train_step(cfg, model_seed, inputs, targets, world, batch, steps=10, warmup=3, want_roofline=True, graph=True,
                     roof_mode="train"):
    """The full training step (trainer.py:325-360): forward + force + loss + loss.backward() through the
    force + ONE flat-buffer RCCL all-reduce (world > 1) + shared-gradient rescale + global-norm clip + AdamW(amsgrad) + EMA.
    fwd/force/backward are one hipGraph; collective and optimizer launches follow it."""
    from gemnet_pytorch_amd.model.gemnet import GemNet
    from gemnet_pytorch_amd.training.ddp import TrainStep
    torch.manual_seed(model_seed)
    torch.cuda.reset_peak_memory_stats(inputs["R"].device)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(inputs["R"].device)
    ts = TrainStep(model, world_size=world, fused_optimizer=True)
    inputs = {k: v for k, v in inputs.items() if k != "_plan"}
    graphed = bool(graph)
    try:
        if graph:
            ts.capture(inputs, targets)
    except Exception as ex:  # noqa: BLE001
        log(f"[bench] training-step capture unavailable ({type(ex).__name__}: {ex}); eager")
        ts._graph, graphed = None, False
    elapsed = time_steps(lambda: ts(inputs, targets), steps, warmup, world)
    out = dict(ms_per_step=round(elapsed / steps * 1e3, 3), molecules_per_s=round(world * batch * steps / elapsed, 1),
               steps=steps, warmup=warmup, hipgraph=graphed, rccl_world_size=world,
               collective=("all_reduce(sum) of one flat %.1f MB fp32 gradient buffer per step over RCCL"
                           % (ts.buf.flat.numel() * 4 / 1e6)) if world > 1 else "none (single process)",
               optimizer="fused rescale + clip + AdamW(amsgrad) + EMA, 2 launches (csrc/optim.hip)",
               loss=float(ts.last_loss),
               # peak of the timed steps (capture + replays); the instrumented pass below holds every intermediate of a step
               # alive through its launch records and is not part of it
               peak_memory_gib=round(torch.cuda.max_memory_allocated(inputs["R"].device) / 2**30, 1))
    if world > 1:
        # the collective alone: the flat gradient buffer through RCCL, back to back between two syncs (max over ranks)
        import torch.distributed as dist
        reps = 20
        for _ in range(3):
            dist.all_reduce(ts.buf.flat)
        el = time_steps(lambda: dist.all_reduce(ts.buf.flat), reps, 0, world)
        nbytes = ts.buf.flat.numel() * 4
        out["allreduce_alone_us"] = round(el / reps * 1e6, 1)
        out["allreduce_bytes"] = nbytes
        # ring all-reduce moves 2 (N-1)/N of the buffer through every link
        out["allreduce_bus_gbs"] = round(2 * (world - 1) / world * nbytes / (el / reps) / 1e9, 2)
    if want_roofline:
        held, ts._graph = getattr(ts, "_graph", None), None
        roof, fam = family_roofline(lambda: ts(inputs, targets, step_optimizer=False), mode=roof_mode)
        ts._graph = held
        log_families("training step", fam)
        out["roofline"] = roof
        # launches through the C ABI (kernels.py launchers) of one eager step, + the grouped weight-gradient pair and the
        # two optimizer launches; the remaining ATen launches (gradient fan-in adds of cross-stream consumers, geometry glue
        # of the composite parts) are in profiles/r4_train_kernel_stats.csv
        out["launches_per_step"] = int(sum(v["launches"] for v in fam.values())) + (2 if ts.wgrad is not None else 0) + 2
    del ts, model
    return out


def extra_interaction_block(model, plan, steps=50, warmup=10):
    """ONE InteractionBlockTripletsOnly (interaction_block.py:158-234 / :363-422), forward + backward w.r.t. all its
    inputs (what the force pass runs per block: weights constant), isolated, hipGraph replay."""
    from gemnet_pytorch_amd import kernels as K
    from gemnet_pytorch_amd import ops
    dev = plan.device
    g = torch.Generator(device="cpu").manual_seed(7)
    A, E, T = plan.n_atoms, plan.n_edges, plan.trip.size
    blk = model.int_blocks[1]
    ea = blk.atom_update.layers[0].weight.shape[0]
    ee = blk.dense_ca.weight.shape[0]
    er = blk.trip_interaction.mlp_rbf.weight.shape[1]
    S, I = model.num_spherical, blk.trip_interaction.mlp_cbf.weight.shape[1]

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev).requires_grad_(True)
    h, m = rnd(A, ea), rnd(E, ee)
    rbf3, rbf_h, rbf_W1 = rnd(E, er, scale=0.3), rnd(E, er, scale=0.3), rnd(E, S, I, scale=0.3)
    sph = rnd(T, S, scale=0.5)
    gh, gm = torch.randn(A, ea, generator=g).to(dev), torch.randn(E, ee, generator=g).to(dev)
    leaves = [h, m, rbf3, rbf_W1, sph, rbf_h]

    def step():
        with ops.weight_cache(model._wcache), ops.fused_first_order(True), ops.param_grads(False):
            h2, m2 = blk(h=h, m=m, rbf3=rbf3, cbf3=(rbf_W1, sph), rbf_h=rbf_h, plan=plan)
            return torch.autograd.grad([h2, m2], leaves, [gh, gm])
    graph, _ = capture(step)
    elapsed = time_steps(graph.replay, steps, warmup)
    sec = elapsed / steps
    # algorithmic cost: forward 2 (E 281 600 + T 448 + A 81 920) flop and
    # E 1 612 + T 36 + A 1 024 + 1.43 MB bytes; the backward w.r.t. the inputs repeats both
    flops = 2 * 2.0 * (E * 281600 + T * 448 + A * 81920)
    byts = 2 * (E * 1612 + T * 36 + A * 1024 + 1.43e6)
    return dict(steps_per_s=round(1.0 / sec, 1), ms_per_step=round(sec * 1e3, 4), steps=steps, warmup=warmup,
                rows=dict(atoms=A, edges=E, triplets=T), arithmetic=K.DEFAULT_CHAIN_MODE,
                algorithmic_gflop=round(flops / 1e9, 2), algorithmic_mb=round(byts / 1e6, 1),
                achieved_tflops=round(flops / sec / 1e12, 2), frac_f32_mfma_peak=round(flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                achieved_gbs=round(byts / sec / 1e9, 1), frac_hbm_roofline=round(byts / sec / 1e9 / PEAK_HBM_GBS, 4),
                note="forward + backward w.r.t. (h, m, rbf3, rbf_W1, sph, rbf_h), constant weights, one hipGraph replay per step")


def extra_gemnet_q(n_mol, n_atoms, rank, steps=10, warmup=3, train=True):
    """BASELINE.json configs[2]: GemNet-Q (quadruplet interactions on) forward+force on the same batch."""
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.model.gemnet import GemNet
    cfg = dict(GEMNET_T, triplets_only=False)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
    model.requires_grad_(False)
    inputs, targets = make_batch(cfg, n_mol, n_atoms, first=rank * n_mol, device=dev)
    plan = GraphPlan.from_inputs(inputs, False).warm()
    sizes = dict(atoms=plan.n_atoms, edges=plan.n_edges, triplets=plan.trip.size, interaction_edges=plan.n_int,
                 intermediate_triplets=plan.n_intm, quadruplets=plan.quad.size)
    step = lambda: model(inputs)  # noqa: E731
    for _ in range(2):
        step()
    graph, _ = capture(step)
    elapsed = time_steps(graph.replay, steps, warmup)
    roof, fam = family_roofline(step, mode="Q")
    log_families("GemNet-Q forward+force", fam)
    out = dict(ms_per_step=round(elapsed / steps * 1e3, 3), molecules_per_s=round(n_mol * steps / elapsed, 1),
               steps=steps, warmup=warmup, per_gpu=sizes, hipgraph=True, roofline=roof)
    # a NEW GemNet-Q batch every step: eager against the padded replay (padded.py pads interaction edges, intermediate triplets
    # and quadruplets too since round 5) — the MD loop of the reference's example runs exactly this model
    try:
        out["dynamic_shape"] = extra_dynamic_shape(cfg, model, n_mol, n_atoms, rank, n_batches=3, steps=6, warmup=2)
    except Exception as ex:  # noqa: BLE001
        out["dynamic_shape"] = dict(error=f"{type(ex).__name__}: {ex}"[:300])
    del graph, model, step, plan, fam
    import gc
    gc.collect()
    # The GemNet-Q TRAINING step (trainer.py:325-360 on configs[2]): every layer on the fused sweeps of ops_train.py — since
    # round 5 also the quadruplet geometry and the quadruplet bilinear layer with its tensor basis in ANGLE form
    # (ops_train._QuadAngles2 / _BilinearAng2: tangent rows rebuilt in-kernel by dual numbers, no (Q, 49) array in any of the
    # four sweeps) — captured in ONE hipGraph like the GemNet-T step.
    if train:
        try:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            held = torch.cuda.memory_allocated(dev)      # what the EARLIER extras of this process still hold (graphs, batches)
            ts_out = extra_train_step(cfg, 1234, inputs, targets, 1, n_mol, steps=5, warmup=2, want_roofline=True, graph=True,
                                      roof_mode="Qtrain")
            if "peak_memory_gib" in ts_out:
                ts_out["peak_memory_gib_incl_other_extras"] = ts_out["peak_memory_gib"]
                ts_out["peak_memory_gib"] = round(ts_out["peak_memory_gib"] - held / 2**30, 1)
            ts_out["note"] = ("forward + force + loss.backward() through the force + fused optimizer; quadruplet interaction on the "
                              "fused angle-form twins (round 4: composite closure over the (Q, 49) harmonics, eager, 112-150 ms); "
                              "roofline = dominant LIBRARY launcher family of one step")
            out["train_step"] = ts_out
        except Exception as ex:  # noqa: BLE001
            out["train_step"] = dict(error=f"{type(ex).__name__}: {ex}"[:300])
        torch.cuda.empty_cache()      # (41 GiB of cached blocks would make the configs[4] extra skip itself)
    return out


def extra_strict_f32(cfg, inputs, ref_out, n_mol, steps=20, warmup=5):
    """The headline workload with the Dense stacks on the exact f32-input MFMA (`matmul_precision = "f32"`:
    v_mfma_f32_16x16x4_f32, csrc/chain.hip) instead of the default two-fp16-plane products: what exactness costs, and how far
    the two arithmetics' forces are apart on this batch (both sit ~2e-6 eV/A from the float64 reference at mean|F| = 1:
    tests/test_gpu_fullsize_golden.py::tB32 is this batch with golden weights, test_matmul_arithmetic_modes the per-mode bars)."""
    from gemnet_pytorch_amd.model.gemnet import GemNet
    dev = inputs["R"].device
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
    model.requires_grad_(False)
    model.matmul_precision = "f32"
    step = lambda: model(inputs)  # noqa: E731
    for _ in range(2):
        step()
    graph, g_out = capture(step)
    elapsed = time_steps(graph.replay, steps, warmup)
    E, F = g_out
    fscale = float(ref_out[1].abs().mean())
    return dict(ms_per_step=round(elapsed / steps * 1e3, 4), molecules_per_s=round(n_mol * steps / elapsed, 1), steps=steps,
                warmup=warmup, hipgraph=True, dense_stack_arithmetic="v_mfma_f32_16x16x4_f32 (fp32 operands, exact products)",
                force_mae_vs_default_arithmetic_rel=round(float((F - ref_out[1]).abs().mean()) / fscale, 9),
                energy_maxdiff_vs_default_arithmetic=round(float((E - ref_out[0]).abs().max()), 7),
                note="same weights (seed 1234) and batch as the headline; the difference of the two forces is relative to mean|F| "
                     "of this random-weight model; each arithmetic's own error against the float64 reference: "
                     "tests/test_gpu_model.py::test_matmul_arithmetic_modes")


def extra_config4_shard(rank, n_mol=64, n_atoms=64, steps=3, warmup=2):
    """BASELINE.json configs[4], ONE GPU's shard: GemNet-Q, 64 molecules x 64 atoms (batch 512 over 8 GPUs), forward+force —
    126 M quadruplets, ~50 GiB; one captured hipGraph since round 6 (eager before), in the default (fp32-equivalent)
    arithmetic, with the roofline of the dominant launcher family (counted traffic: profiles/<round>_traffic_config4.json).  The config's "bf16" is not run: a single-plane bf16
    operand mode was measured SLOWER than the default on this shard (116 vs 109 ms: only the Dense stacks, 4 % of the step,
    take bf16 operands; the quadruplet kernels already run split-fp16 products at fp32 accuracy) and 4e-2 eV/A off, and was
    removed from the model's options in round 5 (docs/HISTORY.md section 14)."""
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
    from gemnet_pytorch_amd.model.gemnet import GemNet
    from gemnet_pytorch_amd.synthetic import make_dataset
    cfg = dict(GEMNET_T, triplets_only=False)
    dev = torch.device("cuda", torch.cuda.current_device())
    free = torch.cuda.mem_get_info(dev)[0] / 2**30
    if free < 80:
        return {"skipped": f"needs ~50 GiB of device memory next to the other extras' pools; {free:.0f} GiB free"}
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev).eval()
    model.requires_grad_(False)
    ds = make_dataset(n_mol, n_atoms, config=4, first=rank * n_mol)
    R = torch.tensor(ds["R"], device=dev)
    idx = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], False, device=dev)(R)
    inputs = dict(Z=torch.tensor(ds["Z"], device=dev).long(), R=R, N=torch.tensor(ds["N"], device=dev).long(), **idx)
    plan = GraphPlan.from_inputs(inputs, False).warm()
    sizes = dict(atoms=plan.n_atoms, edges=plan.n_edges, triplets=plan.trip.size, interaction_edges=plan.n_int,
                 intermediate_triplets=plan.n_intm, quadruplets=plan.quad.size)
    E, F = model(inputs)
    scale = 1.0 / float(F.abs().mean())          # unit forces: they are linear in the output heads
    with torch.no_grad():
        for ob in model.out_blocks:
            ob.out_energy.weight.mul_(scale)
    model._wcache.clear()
    torch.cuda.reset_peak_memory_stats(dev)
    out = dict(per_gpu=sizes, steps=steps, warmup=warmup, hipgraph=False)
    res = {}
    for mode in (None,):
        model.matmul_precision = mode
        step = lambda: model(inputs)  # noqa: E731
        run = step
        try:
            # one hipGraph for the ~650 launches of the step (round 5 timed it eagerly); the graph's private pool holds the
            # step's ~50 GiB of intermediates a second time next to the eager warm-up's cached blocks
            torch.cuda.empty_cache()
            g4, _ = capture(step, warm=1)
            run, out["hipgraph"] = g4.replay, True
        except Exception as ex:  # noqa: BLE001
            out["hipgraph_error"] = f"{type(ex).__name__}: {ex}"[:200]
            torch.cuda.synchronize()
        elapsed = time_steps(run, steps, warmup)
        if out["hipgraph"]:
            del g4, run
            torch.cuda.empty_cache()
        E, F = step()
        res[mode or "default"] = (E.detach().clone(), F.detach().clone())
        out[mode or "default"] = dict(ms_per_step=round(elapsed / steps * 1e3, 2), molecules_per_s=round(n_mol * steps / elapsed, 1))
        if mode is None:
            # (before the instrumented pass, which keeps every intermediate of the step alive)
            out["peak_memory_gib"] = round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)
            roof, fam = family_roofline(step, mode="config4")      # (no committed counter pass for this size: traffic null)
            log_families("configs[4] shard forward+force", fam)
            out["roofline"] = roof
    out["arithmetic"] = ("default: fp32 operands as two fp16 planes, fp32 accumulate (fp32-equivalent); the config's 'bf16' operand "
                         "mode is not offered: measured slower (116 vs 109 ms) and 4e-2 eV/A off in round 4")
    model.matmul_precision = None
    del model, inputs, plan, res
    torch.cuda.empty_cache()
    return out


def extra_dynamic_shape(cfg, model, n_mol, n_atoms, rank, n_batches=4, steps=12, warmup=4):
    """A NEW batch every step (data_provider.py:159-165; ase_calculator.py:155-158 rebuilds the graph every MD step):
    positions resident in HBM -> device index construction (csrc/index_gpu.hip) -> GraphPlan (CSR groupings) -> eager
    forward+force.  No hipGraph (shapes change); includes the host sync that returns the array sizes."""
    from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
    from gemnet_pytorch_amd.synthetic import make_dataset
    dev = torch.device("cuda", torch.cuda.current_device())
    data = []
    for b in range(n_batches):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(rank * n_batches + b + 1) * n_mol)
        data.append(dict(R=torch.tensor(ds["R"], device=dev), Z=torch.tensor(ds["Z"], device=dev).long(),
                         N=torch.tensor(ds["N"], device=dev).long(), N_host=ds["N"]))
    builders = [DeviceGraphBuilder(d["N_host"], cfg["cutoff"], cfg["int_cutoff"], cfg["triplets_only"], device=dev)
                for d in data]
    state = {"i": 0}
    t_idx = [0.0]

    def step():
        b = state["i"] % n_batches
        state["i"] += 1
        d = data[b]
        t0 = time.perf_counter()
        idx = builders[b](d["R"])
        t_idx[0] += time.perf_counter() - t0
        return model(dict(Z=d["Z"], R=d["R"].clone(), N=d["N"], **idx))
    for _ in range(warmup):
        step()
    t_idx[0] = 0.0
    elapsed = time_steps(step, steps, 0)
    out = dict(ms_per_step=round(elapsed / steps * 1e3, 3), molecules_per_s=round(n_mol * steps / elapsed, 1),
               steps=steps, warmup=warmup, distinct_batches=n_batches,
               host_ms_in_index_build=round(t_idx[0] / steps * 1e3, 3),
               note="device index build (incl. its size read-back) + GraphPlan (CSR sorts) + eager forward+force per step; "
                    "positions / Z / N resident in HBM")
    # the same loop with every batch padded to fixed capacities and ONE captured hipGraph replayed (padded.py): the index
    # plan is rebuilt on the device inside the graph, the host launches the index build, a dozen small padding ops, the replay
    if True:       # (triplets-only models since round 3, quadruplet models since round 5)
        try:
            from gemnet_pytorch_amd.padded import PaddedGraphRunner
            idxs = [builders[b](data[b]["R"]) for b in range(n_batches)]
            sizes = [PaddedGraphRunner.sizes_of(i) for i in idxs]
            caps = PaddedGraphRunner.suggest_capacities(sizes)
            runner = PaddedGraphRunner(model, data[0]["Z"], data[0]["N"], caps[0], caps[1],
                                       quad_caps=caps[2] if len(caps) > 2 else None)
            state["i"] = 0

            def pstep():
                b = state["i"] % n_batches
                state["i"] += 1
                # a data provider's batch: its positions do not depend on the previous step (an MD loop would pass False)
                return runner.build_and_run(builders[b], data[b]["R"], Z=data[b]["Z"], positions_ready=True)
            psteps = 4 * steps          # 3 ms each: a longer window than the eager loop's (the first replays run slower)
            for _ in range(2 * warmup):
                pstep()
            el = time_steps(pstep, psteps, 0)
            E0, F0 = model(dict(Z=data[0]["Z"], R=data[0]["R"].clone(), N=data[0]["N"], **idxs[0]))
            E1, F1 = runner(data[0]["R"], idxs[0], Z=data[0]["Z"])
            out["padded_graph"] = dict(ms_per_step=round(el / psteps * 1e3, 3), molecules_per_s=round(n_mol * psteps / el, 1),
                                       steps=psteps,
                                       capacities=dict(edges=runner.e_cap, triplets=runner.t_cap, dummy_atoms=runner.GS * runner.G,
                                                       **(dict(zip(("interaction_edges", "intermediate_triplets", "quadruplets"),
                                                                   runner.quad_caps)) if runner.quad_caps else {})),
                                       batch_sizes=sizes, max_abs_force_deviation_vs_eager=float((F1 - F0).abs().max()),
                                       note="every batch padded with a dummy molecule to fixed capacities, one captured "
                                            "hipGraph replayed; index build (with its size read-back) per step, on its own stream "
                                            "(a data provider's batch does not depend on the previous step)")
            if len({tuple(d["N_host"]) for d in data}) == 1:
                # the index build INSIDE the replayed graph (padded.attach_builder, gn_index_gpu_padded_t): a step is
                # positions in -> one replay -> results out, no read-back, no padding launches on the host
                runner.attach_builder(builders[0])
                state["i"] = 0

                def gstep():
                    b = state["i"] % n_batches
                    state["i"] += 1
                    return runner.run_positions(data[b]["R"], Z=data[b]["Z"])
                for _ in range(2 * warmup):
                    gstep()
                el = time_steps(gstep, psteps, 0)
                E2, F2 = runner.run_positions(data[0]["R"], Z=data[0]["Z"])
                torch.cuda.synchronize()
                out["padded_graph"]["index_in_graph"] = dict(
                    ms_per_step=round(el / psteps * 1e3, 3), molecules_per_s=round(n_mol * psteps / el, 1), steps=psteps,
                    index_error_bits=runner.index_error(), max_abs_force_deviation_vs_eager=float((F2 - F0).abs().max()),
                    note="the same loop with the neighbour list / index arrays built by the first nodes of the replayed graph "
                         "from the positions (device-side counts, pad rows written by the commit kernel): positions in -> "
                         "one hipGraph replay -> energies and forces out")
        except Exception as ex:  # noqa: BLE001
            out["padded_graph"] = dict(error=f"{type(ex).__name__}: {ex}")
    return out


def extra_train_dynamic(cfg, model_seed, n_mol, n_atoms, rank, n_batches=4, steps=8, warmup=3):
    """The training step on a NEW batch every step (what a training loop sees: data_provider.py:159-165): eager TrainStep
    per batch against PaddedTrainStep (every batch padded to fixed capacities, one captured hipGraph replayed)."""
    from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
    from gemnet_pytorch_amd.model.gemnet import GemNet
    from gemnet_pytorch_amd.padded import PaddedGraphRunner
    from gemnet_pytorch_amd.synthetic import make_dataset
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(1)
    data = []
    for b in range(n_batches):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(rank * n_batches + b + 1) * n_mol)
        R = torch.tensor(ds["R"], device=dev, dtype=torch.float32)
        idx = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], True, device=dev)(R)
        data.append(dict(R=R, Z=torch.tensor(ds["Z"], device=dev).long(), N=torch.tensor(ds["N"], device=dev).long(), idx=idx,
                         E=torch.randn(n_mol, 1, generator=g).to(dev), F=torch.randn(n_mol * n_atoms, 3, generator=g).to(dev)))
    sizes = [(int(d["idx"]["id_c"].shape[0]), int(d["idx"]["id3_reduce_ca"].shape[0])) for d in data]
    out = dict(batch_sizes=sizes, steps=steps, warmup=warmup)
    state = {"i": 0}
    for kind in ("eager", "padded_graph"):
        torch.manual_seed(model_seed)
        model = GemNet(**cfg, scale_file=SCALE_FILE).to(dev)
        if kind == "eager":
            ts = TrainStep(model, fused_optimizer=True)

            def step():
                d = data[state["i"] % n_batches]
                state["i"] += 1
                return ts(dict(Z=d["Z"], R=d["R"].clone(), N=d["N"], **d["idx"]), {"E": d["E"], "F": d["F"]})
        else:
            ts = PaddedTrainStep(model, data[0]["Z"], data[0]["N"], *PaddedGraphRunner.suggest_capacities(sizes), fused_optimizer=True)

            def step():
                d = data[state["i"] % n_batches]
                state["i"] += 1
                return ts.step(d["R"], d["idx"], d["E"], d["F"], Z=d["Z"])
        state["i"] = 0
        for _ in range(warmup):
            step()
        el = time_steps(step, steps, 0)
        out[kind] = dict(ms_per_step=round(el / steps * 1e3, 3), molecules_per_s=round(n_mol * steps / el, 1))
        del ts, model
    out["note"] = ("index arrays of every batch resident on the device (the loader's job); eager = ~600 launches issued from "
                   "Python per step; padded_graph = pad + one hipGraph replay + optimizer")
    return out


def dry_run(args, rank, world):
    """The N-rank plumbing without a GPU: process group (gloo), the shard of the global batch, one all-reduce; rank 0
    prints the JSON skeleton.  Everything the real run does before touching the device."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard_ids(world, rank, args.batch, args.atoms)
    t = torch.zeros(world * args.batch, dtype=torch.float64)
    t[ids] = 1.0 + rank
    if world > 1:
        dist.all_reduce(t)
    owners = (t - 1.0).long().tolist()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": None, "n_gpus": world, "steps": 0, "warmup": 0,
                          "dry_run": True, "shard_sizes": [owners.count(r) for r in range(world)],
                          "every_molecule_owned_once": bool(((t >= 1.0) & (t <= world)).all()) and len(owners) == world * args.batch,
                          "collective": "gloo all_reduce" if world > 1 else "none"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    import faulthandler
    faulthandler.enable(file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="molecules per GPU")
    ap.add_argument("--atoms", type=int, default=32)
    ap.add_argument("--mode", choices=["force", "train"], default="force")
    ap.add_argument("--model", choices=["T", "Q"], default="T",
                    help="T = GemNet-T (the headline metric); Q = GemNet-Q (BASELINE.json configs[2], reported as a side case)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extra` object (training step, isolated InteractionBlock, GemNet-Q, dynamic shapes)")
    ap.add_argument("--no-config4", action="store_true", help="skip extra.config4_shard (64 x 64-atom GemNet-Q, ~50 GiB, ~20 s)")
    ap.add_argument("--chain-mode", choices=["f32", "split6", "h3"], default=None,
                    help="arithmetic of the Dense stacks (default: kernels.DEFAULT_CHAIN_MODE = h3, fp32 operands as two fp16 planes)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: launch the ranks, shard the global batch, run one gloo all-reduce and print the JSON "
                         "skeleton (proves the N-rank launch path on a machine without GPUs; tests/test_bench_cpu.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
        # 127.0.0.1) and hand their exit code back; rank 0's JSON line passes through on stdout.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        log(f"[bench] --gpus {args.gpus} without a launcher: re-executing under torch.distributed.run (port {port})")
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    if torch.cuda.device_count() < max(world, local + 1):
        raise SystemExit(f"bench.py: {world} ranks requested but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import contextlib
    import __graft_entry__ as ge
    if rank == 0:
        with contextlib.redirect_stdout(sys.stderr):  # stdout carries exactly one JSON line
            ge.build()
    if world > 1:
        dist.barrier()
    from gemnet_pytorch_amd import kernels as K
    from gemnet_pytorch_amd.graph import GraphPlan
    from gemnet_pytorch_amd.model.gemnet import GemNet
    if args.chain_mode:
        K.DEFAULT_CHAIN_MODE = args.chain_mode

    cfg = dict(GEMNET_T)
    if args.model == "Q":
        cfg["triplets_only"] = False
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(device)
    ids = shard_ids(world, rank, args.batch, args.atoms, cfg["cutoff"])
    inputs, targets = make_batch(cfg, args.batch, args.atoms, first=0, device=device, ids=ids)
    n_local = len(ids)
    plan = GraphPlan.from_inputs(inputs, cfg["triplets_only"]).warm()
    sizes = dict(atoms=plan.n_atoms, edges=plan.n_edges, triplets=plan.trip.size)
    if args.model == "Q":
        sizes.update(interaction_edges=plan.n_int, intermediate_triplets=plan.n_intm, quadruplets=plan.quad.size)
    log(f"[bench] rank {rank}/{world}: {n_local} molecules, {sizes}, Dense-stack arithmetic {K.DEFAULT_CHAIN_MODE}")

    extra = {}
    if args.mode == "train":   # explicit request: the training step IS the timed region
        tr = extra_train_step(cfg, 1234, inputs, targets, world, args.batch, steps=args.steps, warmup=args.warmup,
                              want_roofline=not args.no_roofline and rank == 0, graph=not args.no_graph,
                              roof_mode="Qtrain" if args.model == "Q" else "train")
        elapsed = tr["ms_per_step"] * 1e-3 * args.steps
        graph, roof = tr["hipgraph"], tr.pop("roofline", None)
        extra["train_step"] = tr
    else:
        model.eval()
        model.requires_grad_(False)  # inference: only dE/dR is needed, no parameter-gradient graph
        step = lambda: model(inputs)  # noqa: E731
        for _ in range(2):  # eager warm-up (also validates the path before capture)
            step()
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            try:
                graph, g_out = capture(step)
                ref = step()
                torch.cuda.synchronize()
                if not (torch.allclose(g_out[0], ref[0]) and torch.allclose(g_out[1], ref[1])):
                    raise RuntimeError("hipGraph replay disagrees with eager")
            except Exception as ex:  # noqa: BLE001
                log(f"[bench] hipGraph capture unavailable ({type(ex).__name__}: {ex}); timing eager launches")
                graph = None
                torch.cuda.synchronize()
        elapsed = time_steps(graph.replay if graph is not None else step, args.steps, args.warmup, world)
        roof = None
        if not args.no_roofline and rank == 0:
            roof, fam = family_roofline(step, mode=args.model)      # (the committed counter passes are per workload: T | Q)
            log_families("forward+force", fam)
            # the WHOLE step against the same roof: algorithmic flops of every launcher family of one step / the timed step
            step_flops = sum(v["flops"] for v in fam.values())
            roof["step_algorithmic_gflop"] = round(step_flops / 1e9, 2)
            roof["step_frac"] = round(step_flops / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)

    # ---- the numbers SURVEY.md 8(d) asks for besides the headline (never part of `value`)
    if not args.no_extras and args.mode == "force":
        def guarded(name, fn):
            try:
                t0 = time.time()
                extra[name] = fn()
                log(f"[bench] extra.{name}: {json.dumps(extra[name])[:400]}  ({time.time() - t0:.1f} s)")
            except Exception as ex:  # noqa: BLE001
                extra[name] = {"error": f"{type(ex).__name__}: {ex}"}
                log(f"[bench] extra.{name} failed: {extra[name]['error']}")
            torch.cuda.synchronize()
        if world > 1:
            # multi-GPU: the training step, so that the RCCL gradient all-reduce is actually issued and timed (all ranks)
            guarded("train_step", lambda: extra_train_step(cfg, 1234, inputs, targets, world, args.batch,
                                                           want_roofline=False))
        elif args.model == "T":
            if K.DEFAULT_CHAIN_MODE != "f32":
                ref_out = tuple(t.detach().clone() for t in step())
                guarded("strict_f32", lambda: extra_strict_f32(cfg, inputs, ref_out, args.batch))
            guarded("train_step", lambda: extra_train_step(cfg, 1234, inputs, targets, 1, args.batch))
            guarded("interaction_block_fwd_bwd", lambda: extra_interaction_block(model, plan))
            guarded("dynamic_shape", lambda: extra_dynamic_shape(cfg, model, args.batch, args.atoms, rank))
            guarded("train_step_dynamic", lambda: extra_train_dynamic(cfg, 1234, args.batch, args.atoms, rank))
            guarded("gemnet_q", lambda: extra_gemnet_q(args.batch, args.atoms, rank))
            if not args.no_config4:
                torch.cuda.empty_cache()
                guarded("config4_shard", lambda: extra_config4_shard(rank))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "force":
        # the full configs[1] batch (the workload `value` is quoted on) is the baseline; an 8-molecule sample rides along
        cpu = cpu_baseline(cfg, args.atoms, budget_s=14.0, n_mol=args.batch)
        try:
            cpu["sample_8_molecules"] = cpu_baseline(cfg, args.atoms, budget_s=5.0, n_mol=8, threads=cpu["cores"])
        except Exception as ex:  # noqa: BLE001
            cpu["sample_8_molecules"] = {"error": f"{type(ex).__name__}: {ex}"}
        try:
            cpu["train_step"] = cpu_baseline_train(cfg, args.atoms, n_mol=4, budget_s=8.0, threads=cpu["cores"])
        except Exception as ex:  # noqa: BLE001
            cpu["train_step"] = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        mol_per_s = world * args.batch * args.steps / elapsed     # the shards of all ranks hold world x batch molecules
        line = {
            "metric": f"molecules/sec (forward+force) GemNet-{args.model} on COLL-shaped batches"
                      + ("" if args.mode == "force" else " [training step: fwd+force+backward+allreduce+AdamW]"),
            "value": round(mol_per_s, 2), "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GemNet-{args.model} full (4 blocks, emb 128), batch {args.batch} molecules x "
                                   f"{args.atoms} atoms per GPU, forward+force, fp32 (BASELINE.json configs[{1 if args.model == 'T' else 2}])",
                       "mode": args.mode, "hipgraph": bool(graph), "per_gpu": sizes,
                       "runtime_env": {k: os.environ[k] for k in ("DEBUG_HIP_FORCE_GRAPH_QUEUES",) if k in os.environ},
                       "dense_stack_arithmetic": {"f32": "v_mfma_f32_16x16x4_f32",
                                                  "split6": "fp32 operands as 3 bf16 planes, 6 products on v_mfma_f32_16x16x32_bf16, fp32 accumulate (fp32-equivalent: dropped terms < 2^-24)",
                                                  "h3": "fp32 operands as 2 fp16 planes (hi + 2^-11 lo, 22 significand bits), 3 products on v_mfma_f32_16x16x32_f16, fp32 accumulate (operand rounding 2^-22: force MAE 1e-6 eV/A against float64)",
                                                  "split3": "3 bf16-plane products, fp32 accumulate",
                                                  "bf16": "bf16 operands, fp32 accumulate"}[K.DEFAULT_CHAIN_MODE],
                       "parallelism": f"dp{world} (independent molecule shards, no data-path collective"
                                      + ("; the RCCL gradient all-reduce is timed in extra.train_step)" if world > 1 else ")")},
            "roofline": roof, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 is still measuring the roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
