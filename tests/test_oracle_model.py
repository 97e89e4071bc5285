"""The plain-PyTorch restatement (oracle) vs the reference's own E / F / parameter gradients
(float64, same deterministic weights; fixtures from tests/golden/make_golden.py)."""
import ast

import numpy as np
import pytest
import torch

from oracle import gemnet_oracle as GO
from conftest import check_grad_probes, SCALE_FILE

CASES = ["t1", "q1", "t2", "q2", "t4"]
# model2.npz (round 2): unit-force deep models, direct-force models, a two-target model
CASES2 = ["t2s", "q2s", "dt1", "dq1", "dt2s", "t1m", "t4s", "q4s"]
HEAD_KEYS = ("out_energy.weight", "out_forces.weight")


def load_case(g, tag, dtype=torch.float64):
    cfg = ast.literal_eval(str(g[f"{tag}.cfg"]))
    seed = int(g[f"{tag}.seed"])
    params = GO.make_params(cfg, seed, GO.load_scale_factors(SCALE_FILE), dtype=dtype)
    if f"{tag}.out_scale" in g:   # output heads rescaled so that mean|F| = 1 eV/A (make_golden.py::scale_heads)
        sc = float(g[f"{tag}.out_scale"])
        params = {k: (v * sc if k.endswith(HEAD_KEYS) else v) for k, v in params.items()}
    inputs = {}
    prefix = tag + "."
    skip = {"cfg", "seed", "E", "F", "Et", "Ft", "loss", "grad_names", "grad_norms", "grad_proj", "out_scale"}
    for k, v in g.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        if name in skip or name.startswith("grad.") or name.startswith("L."):
            continue
        t = torch.tensor(v)
        inputs[name] = t.long() if t.dtype in (torch.int32, torch.int64) else t
    return cfg, params, inputs


@pytest.mark.parametrize("tag", CASES)
def test_energy_force(golden_model, tag):
    g = golden_model
    cfg, params, inputs = load_case(g, tag)
    E, F = GO.forward(cfg, params, inputs)
    Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
    scale = max(1.0, float(np.abs(Fref).mean()))
    assert np.abs(E.detach().numpy() - Eref).max() <= 1e-9 * max(1.0, np.abs(Eref).max())
    assert np.abs(F.detach().numpy() - Fref).mean() <= 1e-9 * scale


@pytest.mark.parametrize("tag", CASES2)
def test_energy_force_round2(golden_model2, tag):
    """Oracle == reference on the round-2 fixtures: ABSOLUTE 1e-9 (forces are O(1) eV/A by construction), including the
    direct-force head (gemnet.py:586-597) and the per-target force loop (gemnet.py:599-609)."""
    g = golden_model2
    cfg, params, inputs = load_case(g, tag)
    E, F = GO.forward(cfg, params, inputs)
    Eref, Fref = g[f"{tag}.E"], g[f"{tag}.F"]
    assert F.shape == Fref.shape
    assert np.abs(F.detach().numpy() - Fref).max() <= 1e-9 * max(1.0, float(np.abs(Fref).max()))
    assert np.abs(E.detach().numpy() - Eref).max() <= 1e-9 * max(1.0, np.abs(Eref).max())
    if tag.endswith("s"):
        assert abs(float(np.abs(Fref).mean()) - 1.0) < 1e-9


@pytest.mark.parametrize("tag", ["t1", "q1", "t2", "t2s", "q2s", "dt1", "dq1", "dt2s", "t4s"])
def test_training_gradients(golden_model, golden_model2, tag):
    g = golden_model2 if f"{tag}.E" in golden_model2 else golden_model
    cfg, params, inputs = load_case(g, tag)
    for p in params.values():
        if p.dim() > 0:
            p.requires_grad_(True)
    E, F = GO.forward(cfg, params, inputs, create_graph=True)
    loss = GO.training_loss(E[:, :1], F[:, 0] if F.dim() == 3 else F, torch.tensor(g[f"{tag}.Et"]).double()[:, None],
                            torch.tensor(g[f"{tag}.Ft"]).double())
    np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=1e-9)
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    grads = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    norms = np.array([0.0 if gr is None else float(gr.norm()) for gr in grads])
    np.testing.assert_allclose(norms, g[f"{tag}.grad_norms"], rtol=1e-7, atol=1e-12)
    check_grad_probes(g, tag, dict(zip(names, grads)), rtol=1e-7)
    for n, gr in zip(names, grads):
        key = f"{tag}.grad.{n}"
        if key in g:
            np.testing.assert_allclose(gr.numpy(), g[key], rtol=1e-6, atol=1e-10)
