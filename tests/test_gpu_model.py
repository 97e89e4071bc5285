"""-m gpu: GemNet on the HIP path (fp32, MI355X) against the reference's float64 goldens.

Tolerance (BASELINE.json north_star): force MAE within 1e-5 eV/A for fp32 — asserted LITERALLY (absolute) on
fixtures whose forces are O(1) eV/A: the shallow models t1/q1/dt1/dq1/t1m and the deep models whose output heads were
rescaled to mean|F| = 1 (t2s, q2s, dt2s, t4s = pretrained GemNet-T configuration, q4s = pretrained GemNet-Q
configuration; tests/golden/make_golden.py::run_model2).  The unscaled deep fixtures (|F| up to 3e4 with the untrained
test weights) are kept as a RELATIVE-precision check under their own name.  Every test prints the measured MAE.
Second-order (training) gradients — incl. both published 4-block configurations (t4s, q4s) — are compared per parameter
by norm (rtol 2e-3), through 4 fixed +-1 probe projections of EVERY parameter's gradient, and elementwise for the stored ones."""
import numpy as np
import pytest
import torch

from conftest import SCALE_FILE, check_grad_probes
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
from test_oracle_model import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"
FORCE_TOL = 1e-5


def build(cfg, params):
    model = GemNet(**cfg, scale_file=SCALE_FILE)
    model.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
    return model.to(DEV)


def test_multi_target_forces_fused_equals_composite(golden_model2):
    """num_targets = 2: one backward pass per target over the same graph (retain_graph) through the shared-gradient
    sink of the fused bilinear layers; both targets' forces equal the composite path and the reference."""
    g = golden_model2
    cfg, params, inputs = load_case(g, "t1m")
    res = {}
    for mode in ("fused", "composite"):
        model = build(cfg, params).eval()
        model.force_graph = (mode == "composite")
        E, F = model(to_dev(inputs))
        res[mode] = F.detach().cpu().numpy()
        assert F.shape == (inputs["R"].shape[0], 2, 3)
    for t in range(2):
        d_ref = float(np.abs(res["fused"][:, t] - g["t1m.F"][:, t]).mean())
        d_cmp = float(np.abs(res["fused"][:, t] - res["composite"][:, t]).mean())
        print(f"target {t}: fused vs reference {d_ref:.3e}, fused vs composite {d_cmp:.3e}")
        assert d_ref <= FORCE_TOL and d_cmp <= FORCE_TOL


def to_dev(inputs):
    return {k: v.to(DEV) for k, v in inputs.items()}


def _case(golden_model, golden_model2, tag):
    g = golden_model2 if f"{tag}.E" in golden_model2 else golden_model
    return (g,) + load_case(g, tag)


ABS_CASES = ["t1", "q1", "t2s", "q2s", "t4s", "q4s", "dt1", "dq1", "dt2s", "t1m"]


@pytest.mark.parametrize("tag", ABS_CASES)
def test_energy_force_parity(golden_model, golden_model2, tag):
    """mean|F_hip - F_ref| <= 1e-5 eV/A, absolute, on fixtures with mean|F| in [0.4, 1.5] eV/A: autograd forces (T and
    Q, 1 / 2 / 4 blocks incl. both published configurations), direct forces (dT, dQ; coupled and uncoupled), two targets."""
    g, cfg, params, inputs = _case(golden_model, golden_model2, tag)
    model = build(cfg, params).eval()
    E, F = model(to_dev(inputs))
    assert F.is_cuda and (cfg.get("direct_forces", False) or not F.requires_grad)
    Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
    assert tuple(F.shape) == Fref.shape
    fmean = float(np.abs(Fref).mean())
    assert 0.3 <= fmean <= 1.6, fmean          # the fixture itself is O(1) eV/A: the bar below is literal
    f_mae = float(np.abs(F.detach().cpu().numpy() - Fref).mean())
    f_max = float(np.abs(F.detach().cpu().numpy() - Fref).max())
    e_err = float(np.abs(E.detach().cpu().numpy() - Eref).max())
    print(f"{tag}: force MAE {f_mae:.3e} eV/A (max {f_max:.3e}; mean|F_ref| {fmean:.3f}), energy err {e_err:.3e} "
          f"(max|E_ref| {float(np.abs(Eref).max()):.3f})")
    assert f_mae <= FORCE_TOL
    assert e_err <= 2e-5 * max(1.0, float(np.abs(Eref).max()))


@pytest.mark.parametrize("tag", ["t2", "q2", "t4"])
def test_relative_precision_on_unscaled_deep_fixtures(golden_model, tag):
    """The same models with the raw test weights (mean|F| = 5 .. 3.9e3 eV/A): error relative to the force scale."""
    g = golden_model
    cfg, params, inputs = load_case(g, tag)
    model = build(cfg, params).eval()
    E, F = model(to_dev(inputs))
    Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
    fscale = float(np.abs(Fref).mean())
    f_mae = float(np.abs(F.detach().cpu().numpy() - Fref).mean())
    print(f"{tag}: force MAE / mean|F_ref| = {f_mae / fscale:.3e} (mean|F_ref| {fscale:.3e})")
    assert f_mae <= FORCE_TOL * fscale
    assert float(np.abs(E.detach().cpu().numpy() - Eref).max()) <= 2e-5 * max(1.0, float(np.abs(Eref).max()))


@pytest.mark.parametrize("mode,bar", [("f32", 1e-5), ("split6", 1e-5), ("h3", 1e-5), ("split3", 1e-3)])
@pytest.mark.parametrize("tag", ["t4s", "q4s"])
def test_matmul_arithmetic_modes(golden_model2, tag, mode, bar, monkeypatch):
    """The Dense stacks on the f32 MFMA, on the fp16 planes (3 products) and on the bf16 matrix pipe with 6 / 3 split-operand
    products: force MAE of the published 4-block configurations against the float64 reference, every mode with a bar.
    (Plain bf16 operands — BASELINE configs[4]'s arithmetic — are no model option any more: docs/HISTORY.md section 14.)"""
    from gemnet_pytorch_amd import kernels as K
    g = golden_model2
    cfg, params, inputs = load_case(g, tag)
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", mode)
    model = build(cfg, params).eval()
    E, F = model(to_dev(inputs))
    f_mae = float(np.abs(F.detach().cpu().numpy() - g[f"{tag}.F"]).mean())
    print(f"{tag} [{mode}]: force MAE {f_mae:.3e} eV/A at mean|F| = 1")
    assert np.isfinite(f_mae)
    assert f_mae <= bar


@pytest.mark.parametrize("train", [False, True], ids=["inference", "force-training"])
def test_fp16_plane_overflow_falls_back_to_bf16_planes(golden_model2, train, monkeypatch):
    """The default Dense arithmetic ("h3": two fp16 planes) cannot hold an activation beyond 65 504.  A model whose scale
    factors do not keep activations O(1) — here: the input Dense of a residual stack blown up by 1e6, which fp32 and the
    bf16 planes hold without trouble — must not return inf / nan silently: the eager forward notices, switches this model to "split6", warns, and returns what a split6 model returns."""
    from gemnet_pytorch_amd import kernels as K
    cfg, params, inputs = load_case(golden_model2, "t4s")
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")

    def blown(model):
        blk = model.int_blocks[1]
        with torch.no_grad():
            blk.dense_ca.weight.mul_(1e6)                    # activations of ~1 become ~1e6 > 65504
        return model

    ref = blown(build(cfg, params))
    ref.matmul_precision = "split6"
    ref = ref.train() if train else ref.eval()
    E_ref, F_ref = ref(to_dev(inputs))
    assert bool(torch.isfinite(F_ref).all())
    model = blown(build(cfg, params))
    model = model.train() if train else model.eval()
    assert model.matmul_precision is None
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        E, F = model(to_dev(inputs))
    assert model.matmul_precision == "split6"
    assert bool(torch.isfinite(E).all()) and bool(torch.isfinite(F).all())
    assert torch.equal(E, E_ref) and torch.equal(F, F_ref)
    if train:
        (F ** 2).sum().backward()      # the repeated pass left a usable autograd graph
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # without the guard the same input gives non-finite output (that is what it is for)
    from gemnet_pytorch_amd.model import gemnet as G
    monkeypatch.setattr(G, "_H3_GUARD", False)
    raw = blown(build(cfg, params)).eval()
    E2, F2 = raw(to_dev(inputs))
    assert not bool(torch.isfinite(F2).all())


@pytest.mark.parametrize("tag", ["t1", "q1", "t2", "t2s", "q2s", "t4s", "q4s", "dt1", "dq1", "dt2s"])
def test_training_gradients_parity(golden_model, golden_model2, tag):
    """loss.backward() through the force (second order) resp. through the direct-force head (first order, fused
    layers) against the reference's parameter gradients."""
    g, cfg, params, inputs = _case(golden_model, golden_model2, tag)
    model = build(cfg, params).train()
    E, F = model(to_dev(inputs))
    assert F.requires_grad
    loss = GO.training_loss(E[:, :1], F[:, 0] if F.dim() == 3 else F, torch.tensor(g[f"{tag}.Et"], device=DEV)[:, None],
                            torch.tensor(g[f"{tag}.Ft"], device=DEV))
    np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=2e-5)
    loss.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    norms = np.array([0.0 if named[n].grad is None else float(named[n].grad.norm()) for n in names])
    ref = g[f"{tag}.grad_norms"]
    np.testing.assert_allclose(norms, ref, rtol=2e-3, atol=1e-6 * float(ref.max()))
    # every parameter's gradient projected on 4 fixed +-1 probes: pins the elements, not only the norm
    # q4s (4-block GemNet-Q, output heads scaled by 2.6e-5 to reach unit forces: |activations| ~ 1e4 inside) sits at
    # 3.1e-3 of ||g_ref|| in fp32 on the fused training form (rounds 3, 4 and 5 — since round 5 with the quadruplet layer on
    # the fused angle-form twins too) AND at 4.4e-3 on the composite closure (tools/exp/t2_probe_gpu.py,
    # profiles/r3_second_order_probes.txt); in float64 both forms reproduce the reference to 1e-7 (tests/test_train2_cpu.py):
    # it is fp32 rounding of a 4-block double backward, not a defect of a kernel — the bar is 4e-3 (was 8e-3), the others 2e-3
    rtol = 4e-3 if tag == "q4s" else 2e-3
    worst = check_grad_probes(g, tag, {n: named[n].grad for n in names}, rtol=rtol)
    print(f"{tag}: loss {loss.item():.6f}; worst probe error / ({rtol:g} ||g_ref||) = {worst:.3f}")
    worst_el = 0.0
    for n in names:
        key = f"{tag}.grad.{n}"
        if key in g:
            gr = named[n].grad.cpu().numpy()
            worst_el = max(worst_el, float(np.abs(gr - g[key]).max() / np.abs(g[key]).max()))
            np.testing.assert_allclose(gr, g[key], rtol=5e-3, atol=(4e-3 if tag == "q4s" else 2e-4) * float(np.abs(g[key]).max()))   # measured worst: q4s 1.5e-3, t4s 1.4e-4
    print(f"{tag}: worst elementwise |g - g_ref| / max|g_ref| over the stored gradients = {worst_el:.2e}")


def test_repeatable_bitwise(golden_model):
    """No atomics anywhere on the path: two forwards give bit-identical E and F."""
    cfg, params, inputs = load_case(golden_model, "t2")
    model = build(cfg, params).eval()
    dev = to_dev(inputs)
    E1, F1 = model(dev)
    E2, F2 = model(dev)
    assert torch.equal(E1, E2) and torch.equal(F1, F2)


def test_native_library_loaded():
    from gemnet_pytorch_amd import _lib
    lib = _lib.load()
    assert lib.gn_abi_version() == 15
    with open("/proc/self/maps") as f:
        assert "libgemnet_hip.so" in f.read()


@pytest.mark.parametrize("triplets_only", [True, False])
def test_direct_forces_fused_equals_composite(golden_model, triplets_only):
    """GemNet-dT/dQ on the GPU: fused single-launch layers vs the composite op closure."""
    tag = "t1" if triplets_only else "q1"
    cfg, _, inputs = load_case(golden_model, tag)
    cfg = dict(cfg, direct_forces=True, forces_coupled=True)
    params = GO.make_params(cfg, 11, GO.load_scale_factors(SCALE_FILE))
    res = {}
    for mode in ("fused", "composite"):
        model = build(cfg, params).train()
        model.force_graph = (mode == "composite")
        E, F = model(to_dev(inputs))
        (E.sum() + (F ** 2).sum()).backward()
        res[mode] = (E.detach(), F.detach(), {n: p.grad.clone() for n, p in model.named_parameters()
                                              if p.grad is not None})
    fs = max(1.0, float(res["composite"][1].abs().mean()))
    assert float((res["fused"][1] - res["composite"][1]).abs().mean()) <= 1e-5 * fs
    for n, gr in res["composite"][2].items():
        ref = float(gr.norm())
        assert abs(float(res["fused"][2][n].norm()) - ref) <= 2e-3 * ref + 1e-6, n


def test_eval_forward_force_in_hipgraph(golden_model):
    """The whole forward+force step replays from one hipGraph and matches the eager result."""
    cfg, params, inputs = load_case(golden_model, "t2")
    model = build(cfg, params).eval()
    model.requires_grad_(False)
    dev = to_dev(inputs)
    for _ in range(2):
        model(dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model(dev)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Eg, Fg = model(dev)
    graph.replay()
    torch.cuda.synchronize()
    E, F = model(dev)
    assert torch.equal(E, Eg) and torch.equal(F, Fg)


@pytest.mark.parametrize("mode,e_bar", [("f32", 2e-5), ("h3", 5e-5)])
def test_layer_stacks_match_per_layer_path(golden_model, monkeypatch, mode, e_bar):
    """The LDS-resident stack path gives the same E/F as per-layer GEMMs (always the f32 MFMA): with the f32 chain kernel
    (same arithmetic on both sides) and with the default two-plane fp16 chain kernel (22-bit operands on one side: the
    energy of this unscaled fixture, |E| = 0.87, moves by 2.3e-5 — the force bar is the same for both)."""
    from gemnet_pytorch_amd import kernels as K
    from gemnet_pytorch_amd import ops
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", mode)
    cfg, params, inputs = load_case(golden_model, "t2")
    model = build(cfg, params).eval()
    dev = to_dev(inputs)
    monkeypatch.setattr(ops, "USE_STACKS", False)
    E0, F0 = model(dev)
    monkeypatch.setattr(ops, "USE_STACKS", True)
    E1, F1 = model(dev)
    fs = max(1.0, float(F0.abs().mean()))
    assert float((F1 - F0).abs().mean()) <= 1e-5 * fs
    # energies: each path against the float64 REFERENCE (the unscaled fixture sums cancelling terms: the two fp32 paths sit on
    # either side of it), and against each other within the sum of the two bars
    E_ref = torch.tensor(golden_model["t2.E"], device=E0.device, dtype=torch.float32).reshape(E0.shape)
    es = max(1.0, float(E_ref.abs().max()))
    err0, err1 = float((E0 - E_ref).abs().max()), float((E1 - E_ref).abs().max())
    print(f"[{mode}] |E - E_ref|: per-layer {err0:.2e}, stacks {err1:.2e}; |E_stacks - E_per_layer| {float((E1 - E0).abs().max()):.2e}")
    # (3e-5: fp32 rounding of a sum of cancelling per-atom energies — measured 2.0e-5 / 6.2e-6 (f32) and 4.6e-6 / 6.0e-6 (h3) on
    #  MI355X; the FORCE bar above is the physical one)
    assert err0 <= 3e-5 * es and err1 <= max(e_bar, 3e-5) * es
    assert float((E1 - E0).abs().max()) <= (3e-5 + e_bar) * es


def test_fused_trainer_step_matches_torch_optimizers(golden_model):
    """N3: rescale + clip + AdamW/Adam(amsgrad) + EMA in two launches == the reference sequence built from
    scale_shared_grads, clip_grad_norm_, torch.optim and ExponentialMovingAverage (three steps)."""
    import copy
    from gemnet_pytorch_amd.training.ddp import TrainStep, make_optimizer
    from gemnet_pytorch_amd.training.ema_decay import ExponentialMovingAverage
    g = golden_model
    cfg, params, inputs = load_case(g, "t1")
    dev = to_dev(inputs)
    targets = {"E": torch.tensor(g["t1.Et"], device=DEV)[:, None], "F": torch.tensor(g["t1.Ft"], device=DEV)}
    a = build(cfg, params).train()
    b = copy.deepcopy(a)
    clip = 0.5  # active clipping
    ref = TrainStep(a, grad_clip_max=clip, optimizer=make_optimizer(a, learning_rate=2e-3, weight_decay=0.01))
    ema = ExponentialMovingAverage([p for p in a.parameters() if p.requires_grad], 0.9)
    fused = TrainStep(b, grad_clip_max=clip, fused_optimizer=True)
    fused.fused.lr, fused.fused.wd[fused.fused.wd > 0], fused.fused.ema_decay = 2e-3, 0.01, 0.9
    for _ in range(3):
        la = ref(dev, targets)
        ema.update()
        lb = fused(dict(dev), targets)
        assert abs(float(la) - float(lb)) <= 2e-5 * abs(float(la))
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        if pa.requires_grad:
            scale = max(1e-3, float(pa.abs().max()))
            assert float((pa - pb).abs().max()) <= 2e-4 * scale, n
    for sa, sb in zip(ema.shadow_params, fused.fused.ema_parameters()):
        assert float((sa - sb).abs().max()) <= 2e-4 * max(1e-3, float(sa.abs().max()))


def test_grouped_weight_gradients_match_autograd_accumulation(golden_model):
    """TrainStep defers every dW = X^T Y of the final backward into one grouped split-K launch + one grouped fold
    (training/wgrad_queue.py); the flat gradient equals autograd's own accumulation."""
    import copy
    from gemnet_pytorch_amd.training.ddp import TrainStep
    g = golden_model
    cfg, params, inputs = load_case(g, "t2")
    dev = to_dev(inputs)
    targets = {"E": torch.tensor(g["t2.Et"], device=DEV)[:, None], "F": torch.tensor(g["t2.Ft"], device=DEV)}
    a = build(cfg, params).train()
    b = copy.deepcopy(a)
    ta, tb = TrainStep(a), TrainStep(b)
    ta.wgrad = None
    la = ta(dev, targets, step_optimizer=False)
    lb = tb(dict(dev), targets, step_optimizer=False)
    torch.cuda.synchronize()
    assert float(la) == float(lb)
    ref = ta.buf.flat
    assert float(ref.abs().max()) > 0
    assert float((tb.buf.flat - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    # ... parameter by parameter (the column blocks of the concat-Dense weights and the (C, O) blocks of the bilinear
    # weights are strided fold targets: a misplaced block would hide behind the global maximum)
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        if pa.grad is not None:
            sc = float(pa.grad.abs().max())
            assert float((pa.grad - pb.grad).abs().max()) <= 2e-4 * sc + 1e-7 * float(ref.abs().max()), n
    # a second step reuses the tables; gradients are zeroed and rebuilt identically
    g1 = tb.buf.flat.clone()
    tb(dict(dev), targets, step_optimizer=False)
    torch.cuda.synchronize()
    assert torch.equal(g1, tb.buf.flat)


def test_model_matmul_precision_flag(golden_model2):
    """GemNet.matmul_precision selects the Dense-stack arithmetic per model: the three fp32-equivalent forms agree to fp32
    rounding; reduced-precision operand modes ("bf16": removed in round 5, docs/HISTORY.md section 14) and unknown names raise."""
    g = golden_model2
    cfg, params, inputs = load_case(g, "t2s")
    model = build(cfg, params).eval()
    dev = to_dev(inputs)
    F = {}
    for mode in (None, "f32", "split6", "h3"):
        model.matmul_precision = mode
        F[mode] = model(dict(dev))[1].detach()
    for mode in ("f32", "split6", "h3"):
        assert float((F[None] - F[mode]).abs().mean()) <= 1e-5, mode
    for bad in ("bf16", "fp8"):
        model.matmul_precision = bad
        with pytest.raises(ValueError):
            model(dict(dev))


def test_force_graphs_runtime(golden_model2):
    """runtime.ForceGraphs: sub-batches captured once into hipGraphs on their own streams reproduce the eager result;
    set_positions feeds new coordinates to the captured buffers (MD with a fixed neighbour list)."""
    from gemnet_pytorch_amd.runtime import ForceGraphs
    g = golden_model2
    cfg, params, inputs = load_case(g, "t2s")
    model = build(cfg, params).eval()
    model.requires_grad_(False)
    a, b = to_dev(inputs), to_dev(inputs)
    shift = torch.zeros_like(b["R"])
    shift[:, 0] = 0.01 * torch.arange(b["R"].shape[0], device=DEV) / b["R"].shape[0]
    E0, F0 = model(dict(a))
    E1, F1 = model(dict(b, R=(b["R"] + shift)))
    runner = ForceGraphs(model, [a, b])
    runner.set_positions(1, b["R"] + shift)
    runner()
    torch.cuda.synchronize()
    E, F = runner.energies_forces()
    n = a["R"].shape[0]
    assert torch.allclose(E[:E0.shape[0]], E0, atol=1e-6) and torch.allclose(F[:n], F0, atol=1e-6)
    assert torch.allclose(E[E0.shape[0]:], E1, atol=1e-6) and torch.allclose(F[n:], F1, atol=1e-6)
