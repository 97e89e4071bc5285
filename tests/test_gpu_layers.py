"""-m gpu: the per-layer fixtures through the HIP layer modules — P1 InteractionBlock[TripletsOnly],
P2 TripletInteraction, P3 QuadrupletInteraction, P4 EfficientInteractionBilinear, P5 EfficientInteractionDownProjection,
P10 AtomUpdateBlock, P13 OutputBlock — recorded inputs in, compared with the REFERENCE module's recorded outputs
(tests/golden/model2.npz, `<case>.L.*`), in the fused first-order mode and in the composite mode."""
import pytest
import torch

from layer_cases import LAYER_CASES, replay
from test_gpu_model import build, to_dev
from test_oracle_model import load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "composite"])
@pytest.mark.parametrize("case,layer", LAYER_CASES)
def test_layer_matches_reference_module(golden_model2, case, layer, fused):
    g = golden_model2
    cfg, params, inputs = load_case(g, case)
    model = build(cfg, params).eval()
    ours, ref = replay(model, g, case, layer, to_dev(inputs), "cuda", torch.float32, fused)
    assert len(ours) == len(ref)
    worst = 0.0
    for o, r in zip(ours, ref):
        assert o.shape == r.shape
        scale = max(1.0, float(r.abs().max()))
        err = float((o - r).abs().max()) / scale
        worst = max(worst, err)
    print(f"{case} {layer} [{'fused' if fused else 'composite'}]: max err / scale {worst:.2e}")
    assert worst <= 2e-5
