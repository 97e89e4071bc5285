"""The plain-PyTorch restatement (oracle) vs the reference's own E / F / parameter gradients
(float64, same deterministic weights; fixtures from tests/golden/make_golden.py)."""
import ast

import numpy as np
import pytest
import torch

from oracle import gemnet_oracle as GO
from conftest import SCALE_FILE

CASES = ["t1", "q1", "t2", "q2", "t4"]


def load_case(g, tag, dtype=torch.float64):
    cfg = ast.literal_eval(str(g[f"{tag}.cfg"]))
    seed = int(g[f"{tag}.seed"])
    params = GO.make_params(cfg, seed, GO.load_scale_factors(SCALE_FILE), dtype=dtype)
    inputs = {}
    prefix = tag + "."
    skip = {"cfg", "seed", "E", "F", "Et", "Ft", "loss", "grad_names", "grad_norms"}
    for k, v in g.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        if name in skip or name.startswith("grad."):
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


@pytest.mark.parametrize("tag", ["t1", "q1", "t2"])
def test_training_gradients(golden_model, tag):
    g = golden_model
    cfg, params, inputs = load_case(g, tag)
    for p in params.values():
        if p.dim() > 0:
            p.requires_grad_(True)
    E, F = GO.forward(cfg, params, inputs, create_graph=True)
    loss = GO.training_loss(E, F, torch.tensor(g[f"{tag}.Et"]).double()[:, None],
                            torch.tensor(g[f"{tag}.Ft"]).double())
    np.testing.assert_allclose(loss.item(), float(g[f"{tag}.loss"]), rtol=1e-9)
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    grads = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    norms = np.array([0.0 if gr is None else float(gr.norm()) for gr in grads])
    np.testing.assert_allclose(norms, g[f"{tag}.grad_norms"], rtol=1e-7, atol=1e-12)
    for n, gr in zip(names, grads):
        key = f"{tag}.grad.{n}"
        if key in g:
            np.testing.assert_allclose(gr.numpy(), g[key], rtol=1e-6, atol=1e-10)
