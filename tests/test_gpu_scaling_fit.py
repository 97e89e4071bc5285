"""-m gpu: H3 on the device — the scale-factor fit of fit_scaling.py:94-159 (gemnet/model/layers/scaling.py:84-147) run
through the native GemNet / Trainer / AutomaticFit with the HIP kernels in fp32, against the values the REFERENCE classes
fitted on the same batches in float64 (tests/golden/scaling_fit.json, tests/golden/make_golden.py::golden_scaling).
Each factor is a ratio of two activation variances, and every factor feeds the statistics of the later ones: fp32
against float64 is asserted at 2e-3 relative (the CPU emulation of the same code holds 1e-5)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.model.scaling import AutomaticFit
from gemnet_pytorch_amd.model.utils import read_json, write_json
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.metrics import Metrics
from gemnet_pytorch_amd.training.trainer import Trainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["T", "Q"])
def test_fit_all_scale_factors_on_the_device(tag, tmp_path):
    with open(os.path.join(GOLDEN, "scaling_fit.json")) as f:
        g = json.load(f)[tag]
    cfg = g["cfg"]
    data = dict(N=np.array(g["N"], np.int32), Z=np.array(g["Z"], np.int32), R=np.array(g["R"], np.float32),
                E=np.zeros(len(g["N"]), np.float32), F=np.zeros((len(g["Z"]), 3), np.float32))
    dc = DataContainer.from_arrays(data, 5.0, 10.0, triplets_only=cfg["triplets_only"])
    batches = g["batches"]

    def stream():
        i = 0
        while True:
            b = dc[batches[i % len(batches)]]
            yield {k: v for k, v in b.items() if k not in ("E", "F")}, {"E": b["E"], "F": b["F"]}
            i += 1

    scale_file = str(tmp_path / "scaling.json")
    write_json(scale_file, {"comment": "test"})
    order = []
    try:
        AutomaticFit.set2fitmode()
        model = GemNet(**cfg, scale_file=scale_file)
        params = GO.make_params(cfg, g["seed"], None, torch.float64)
        model.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
        model = model.to("cuda")
        trainer = Trainer(model)
        metrics = Metrics("train", trainer.tracked_metrics, None)
        it = stream()
        while not AutomaticFit.fitting_completed():
            for _ in range(len(batches)):
                trainer.test_on_batch(it, metrics)
            order.append(AutomaticFit.activeVar._name)
            AutomaticFit.activeVar.fit()
    finally:
        AutomaticFit.fitting_mode = False
        AutomaticFit.reset()
    assert order == g["order"]
    fitted = read_json(scale_file)
    fitted.pop("comment")
    assert sorted(fitted) == sorted(g["fitted"])
    worst = max(abs(fitted[k] - v) / abs(v) for k, v in g["fitted"].items())
    print(f"scale-factor fit on the device, GemNet-{tag}: {len(fitted)} factors, worst relative deviation from the "
          f"float64 reference fit {worst:.2e}")
    for k, v in g["fitted"].items():
        assert abs(fitted[k] - v) <= 2e-3 * abs(v), (k, fitted[k], v)
