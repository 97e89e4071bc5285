"""G3 per-layer fixtures on CPU (float64, launchers emulated): the module wiring of every hot-path layer — P1 block,
P2 triplet / P3 quadruplet interaction, P4 bilinear, P5 down projection, P10 atom update, P13 output head — reproduces
the REFERENCE module's recorded output from its recorded input, in both execution modes."""
import numpy as np
import pytest
import torch

import cpu_kernels
from layer_cases import LAYER_CASES, replay
from test_model_cpu import build
from test_oracle_model import load_case


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "composite"])
@pytest.mark.parametrize("case,layer", LAYER_CASES)
def test_layer_matches_reference_module(golden_model2, case, layer, fused):
    g = golden_model2
    cfg, params, inputs = load_case(g, case)
    with cpu_kernels.emulate():
        model = build(cfg, params).eval()
        ours, ref = replay(model, g, case, layer, inputs, "cpu", torch.float64, fused)
    assert len(ours) == len(ref)
    for o, r in zip(ours, ref):
        assert o.shape == r.shape
        scale = max(1.0, float(r.abs().max()))
        assert float((o - r).abs().max()) <= 1e-9 * scale, (case, layer, float((o - r).abs().max()), scale)
