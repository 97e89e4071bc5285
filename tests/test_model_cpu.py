"""GemNet module wiring (gemnet_pytorch_amd.model) checked on CPU in float64 with the HIP launchers
emulated (tests/cpu_kernels.py): energies, forces and training gradients against the reference's
goldens, state_dict key set against the reference's, and the no-CPU-fallback guarantee."""
import ast
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SCALE_FILE, check_grad_probes
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
import cpu_kernels
from test_oracle_model import load_case


def build(cfg, params, dtype=torch.float64):
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(dtype)   # dtype first: rescaled heads are not fp32-representable
    model.load_state_dict(GO.expand_to_reference_state_dict(params), strict=True)
    model._check_inputs = lambda R: None
    return model


@pytest.mark.parametrize("tag", ["t1", "q1", "t2"])
def test_energy_force_and_training_grads(golden_model, tag):
    g = golden_model
    cfg, params, inputs = load_case(g, tag)
    with cpu_kernels.emulate():
        model = build(cfg, params)
        model.train()
        inputs["R"] = inputs["R"].double()
        E, F = model(inputs)
        Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
        assert np.abs(E.detach().numpy() - Eref).max() <= 1e-9 * max(1.0, np.abs(Eref).max())
        assert np.abs(F.detach().numpy() - Fref).mean() <= 1e-9 * max(1.0, float(np.abs(Fref).mean()))
        loss = GO.training_loss(E, F, torch.tensor(g[f"{tag}.Et"]).double()[:, None],
                                torch.tensor(g[f"{tag}.Ft"]).double())
        np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=1e-9)
        loss.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    norms = np.array([0.0 if named[n].grad is None else float(named[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g[f"{tag}.grad_norms"], rtol=1e-7, atol=1e-12)
    check_grad_probes(g, tag, {n: named[n].grad for n in names}, rtol=1e-7)
    for n in names:
        key = f"{tag}.grad.{n}"
        if key in g:
            np.testing.assert_allclose(named[n].grad.numpy(), g[key], rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("stacks", [False, True])
@pytest.mark.parametrize("tag", ["t1", "q1", "t2"])
def test_eval_mode_fused_first_order_path(golden_model, tag, stacks, monkeypatch):
    """eval(): single-launch fused layers (optionally LDS-resident layer stacks) + first-order
    backward; same E/F as the reference."""
    g = golden_model
    cfg, params, inputs = load_case(g, tag)
    import gemnet_pytorch_amd.kernels as K
    from gemnet_pytorch_amd import ops
    monkeypatch.setattr(ops, "USE_STACKS", stacks)
    calls = {"chain": 0}
    with cpu_kernels.emulate():
        emu_chain = K.chain

        def counting_chain(prog):
            calls["chain"] += 1
            return emu_chain(prog)

        K.chain = counting_chain
        model = build(cfg, params).eval()
        inputs["R"] = inputs["R"].double()
        E, F = model(inputs)
    # every block: 2 edge stacks + atom stack + interaction head(s) (1 for T, 2 for Q) + (T) the up-projection pair,
    # every output block: 1 stack; each once forward, once backward
    heads = 1 if cfg["triplets_only"] else 2
    pairs = 1 if cfg["triplets_only"] else 0
    assert calls["chain"] == (2 * ((3 + heads + pairs) * cfg["num_blocks"] + cfg["num_blocks"] + 1) if stacks else 0), calls
    assert not F.requires_grad and inputs["R"].requires_grad is False
    Fref, Eref = g[f"{tag}.F"], g[f"{tag}.E"]
    assert np.abs(F.numpy() - Fref).mean() <= 1e-9 * max(1.0, float(np.abs(Fref).mean()))
    assert np.abs(E.detach().numpy() - Eref).max() <= 1e-9 * max(1.0, np.abs(Eref).max())


@pytest.mark.parametrize("triplets_only", [True, False])
def test_direct_forces_fused_equals_composite(golden_model, triplets_only):
    """GemNet-dT/dQ (first-order training): fused layers give the same outputs and parameter
    gradients as the composite op closure."""
    from gemnet_pytorch_amd import ops
    tag = "t1" if triplets_only else "q1"
    cfg, _, inputs = load_case(golden_model, tag)
    cfg = dict(cfg, direct_forces=True, forces_coupled=True)
    sf = GO.load_scale_factors(SCALE_FILE)
    params = GO.make_params(cfg, 11, sf)
    inputs["R"] = inputs["R"].double()
    res = {}
    with cpu_kernels.emulate():
        for mode in ("fused", "composite"):
            model = build(cfg, params).train()
            model.force_graph = (mode == "composite")  # graph=True routes through the composite ops
            E, F = model(inputs)
            assert F.shape == (inputs["R"].shape[0], 1, 3)
            (E.sum() + (F ** 2).sum()).backward()
            res[mode] = (E.detach(), F.detach(), {n: p.grad.clone() for n, p in model.named_parameters()
                                                  if p.grad is not None})
    assert torch.allclose(res["fused"][0], res["composite"][0], rtol=1e-10, atol=1e-12)
    assert torch.allclose(res["fused"][1], res["composite"][1], rtol=1e-10, atol=1e-12)
    assert res["fused"][2].keys() == res["composite"][2].keys() and len(res["fused"][2]) > 20
    for n, gr in res["fused"][2].items():
        assert torch.allclose(gr, res["composite"][2][n], rtol=1e-8, atol=1e-11), n


@pytest.mark.parametrize("variant,extra", [("T", {}), ("Q", {}), ("dT", {"direct_forces": True})])
def test_state_dict_keys_match_reference(variant, extra):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        ref = json.load(f)[variant]
    cfg = dict(num_spherical=7, num_radial=6, num_blocks=1, emb_size_atom=64, emb_size_edge=64,
               emb_size_trip=32, emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32,
               emb_size_bil_quad=32, emb_size_bil_trip=32, num_before_skip=1, num_after_skip=1,
               num_concat=1, num_atom=2, triplets_only=variant != "Q", **extra)
    model = GemNet(**cfg, scale_file=SCALE_FILE)
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sd == ref["state_dict"]
    assert [n for n, _ in model.named_parameters()] == ref["named_parameters"]


@pytest.mark.parametrize("name", ["GemNet-T", "GemNet-Q"])
def test_published_model_configurations_and_weight_files(name, tmp_path):
    """pretrained/*/model_kwargs.json of the reference: the same keyword set builds the model, its
    state_dict has the reference's keys and shapes, and a `model.pth` written in that format loads (load_weights)."""
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        ref = json.load(f)["pretrained/" + name]
    kw = dict(ref["model_kwargs"], scale_file=SCALE_FILE)
    model = GemNet(**kw)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == ref["state_dict"]
    assert [n for n, _ in model.named_parameters()] == ref["named_parameters"]
    torch.manual_seed(7)
    blob = {k: v.clone() for k, v in GemNet(**kw).state_dict().items()}   # (aliased keys stay consistent)
    assert any(not torch.equal(v, blob[k]) for k, v in model.state_dict().items())
    path = str(tmp_path / "model.pth")
    torch.save(blob, path)
    model.load_weights(path)
    for k, v in model.state_dict().items():
        assert torch.equal(v, blob[k]), k
    model.save_weights(path)
    assert set(torch.load(path)) == set(ref["state_dict"])


def test_no_cpu_fallback():
    cfg = dict(num_spherical=7, num_radial=6, num_blocks=1, emb_size_atom=16, emb_size_edge=16,
               emb_size_trip=16, emb_size_quad=16, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=16,
               emb_size_bil_quad=16, emb_size_bil_trip=16, num_before_skip=1, num_after_skip=1,
               num_concat=1, num_atom=1, triplets_only=True)
    model = GemNet(**cfg, scale_file=SCALE_FILE)
    with pytest.raises(RuntimeError, match="HIP device only"):
        model({"R": torch.zeros(2, 3), "Z": torch.ones(2, dtype=torch.long)})
    from gemnet_pytorch_amd import kernels
    with pytest.raises(RuntimeError, match="no CPU fallback|CPU fallback"):
        kernels.ssilu(torch.zeros(4), 0)


def test_derived_weight_cache_is_owned_by_the_model(golden_model):
    """Transposed / contiguous weight copies are cached per model instance and only while that model's forward
    runs: a process-global, address-keyed cache handed one model's transposes to the next model allocated at
    the same addresses (seen on the GPU as a force error of 0.46 in a multi-model test session)."""
    from gemnet_pytorch_amd import ops
    g = golden_model
    cfg, params, inputs = load_case(g, "t1")
    assert ops._WT_CACHE is None
    W = torch.randn(8, 4, dtype=torch.float64)
    assert ops.transposed(W) is not ops.transposed(W)          # nothing cached outside a model forward
    with cpu_kernels.emulate():
        a = build(cfg, params).eval().requires_grad_(False)   # frozen weights: derived copies are cached
        inputs["R"] = inputs["R"].double()
        a(inputs)
        assert len(a._wcache) > 0 and ops._WT_CACHE is None
        b = build(cfg, {k: v * 1.5 for k, v in params.items()}).eval()
        assert b._wcache == {} and b._wcache is not a._wcache
        import copy
        c = copy.deepcopy(a)
        assert c._wcache == {} and len(a._wcache) > 0
        a.double()
        assert a._wcache == {}


@pytest.mark.parametrize("tag", ["t2", "q1"])
def test_multi_target_forces_fused_equals_composite(golden_model, tag):
    """num_targets = 2 with autograd forces: GemNet.forward differentiates once per target with retain_graph=True
    (gemnet.py:605-611), so the shared-gradient sink of the fused bilinear layers sees several backward passes over
    one graph.  Every target's force must equal the composite (force_graph=True) result."""
    cfg, _, inputs = load_case(golden_model, tag)
    cfg = dict(cfg, num_targets=2)
    params = GO.make_params(cfg, 17, GO.load_scale_factors(SCALE_FILE))
    inputs["R"] = inputs["R"].double()
    res = {}
    with cpu_kernels.emulate():
        for mode in ("fused", "composite"):
            model = build(cfg, params).eval()
            model.force_graph = (mode == "composite")
            E, F = model(dict(inputs))
            assert F.shape == (inputs["R"].shape[0], 2, 3)
            res[mode] = (E.detach(), F.detach())
            E2, F2 = model(dict(inputs))           # the sink state must not leak into the next forward
            assert torch.allclose(F2.detach(), F.detach(), rtol=1e-12, atol=1e-14)
    scale = float(res["composite"][1].abs().mean())
    for t in range(2):
        d = float((res["fused"][1][:, t] - res["composite"][1][:, t]).abs().max())
        assert d <= 1e-10 * max(1.0, scale), (t, d, scale)
    assert torch.allclose(res["fused"][0], res["composite"][0], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("tag", ["t2s", "q2s", "dt1", "dq1", "dt2s", "t1m", "t4s"])
def test_round2_fixtures_energy_force(golden_model2, tag, mode):
    """Unit-force deep models, direct-force models (GemNet-dT/dQ, coupled and uncoupled) and the two-target model on
    the emulated launchers, composite (train) and fused (eval) execution modes, ABSOLUTE 1e-9 against the reference."""
    g = golden_model2
    cfg, params, inputs = load_case(g, tag)
    with cpu_kernels.emulate():
        model = build(cfg, params)
        model = model.train() if mode == "train" else model.eval()
        inputs["R"] = inputs["R"].double()
        E, F = model(inputs)
    Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
    assert F.shape == Fref.shape
    assert np.abs(F.detach().numpy() - Fref).max() <= 1e-9 * max(1.0, float(np.abs(Fref).max()))
    assert np.abs(E.detach().numpy() - Eref).max() <= 1e-9 * max(1.0, float(np.abs(Eref).max()))


@pytest.mark.parametrize("tag", ["t2s", "dt1", "dq1", "dt2s"])
def test_round2_training_gradients(golden_model2, tag):
    g = golden_model2
    cfg, params, inputs = load_case(g, tag)
    with cpu_kernels.emulate():
        model = build(cfg, params).train()
        inputs["R"] = inputs["R"].double()
        E, F = model(inputs)
        loss = GO.training_loss(E[:, :1], F[:, 0] if F.dim() == 3 else F, torch.tensor(g[f"{tag}.Et"]).double()[:, None],
                                torch.tensor(g[f"{tag}.Ft"]).double())
        np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=1e-9)
        loss.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    norms = np.array([0.0 if named[n].grad is None else float(named[n].grad.norm()) for n in names])
    np.testing.assert_allclose(norms, g[f"{tag}.grad_norms"], rtol=1e-7, atol=1e-12)
    check_grad_probes(g, tag, {n: named[n].grad for n in names}, rtol=1e-7)
    for n in names:
        key = f"{tag}.grad.{n}"
        if key in g:
            np.testing.assert_allclose(named[n].grad.numpy(), g[key], rtol=1e-6, atol=1e-10)
