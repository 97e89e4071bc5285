"""H3: scale-factor fitting (fit_scaling.py:94-159 protocol) through the native GemNet / Trainer /
AutomaticFit against the values the REFERENCE classes fitted on the same batches
(tests/golden/scaling_fit.json, made by tests/golden/make_golden.py::golden_scaling)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import gemnet_oracle as GO
import cpu_kernels
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.model.scaling import AutomaticFit
from gemnet_pytorch_amd.model.utils import read_json, write_json
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.metrics import Metrics
from gemnet_pytorch_amd.training.trainer import Trainer


@pytest.mark.parametrize("tag", ["T", "Q"])
def test_fit_all_scale_factors_like_fit_scaling_py(tag, tmp_path):
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
            inputs = {k: v for k, v in b.items() if k not in ("E", "F")}
            inputs["R"] = inputs["R"].double()
            yield inputs, {"E": b["E"].double(), "F": b["F"].double()}
            i += 1

    scale_file = str(tmp_path / "scaling.json")
    write_json(scale_file, {"comment": "test"})
    order = []
    try:
        with cpu_kernels.emulate():
            AutomaticFit.set2fitmode()
            model = GemNet(**cfg, scale_file=scale_file)
            model.load_state_dict(GO.expand_to_reference_state_dict(GO.make_params(cfg, g["seed"], None, torch.float64)),
                                  strict=True)
            model = model.double()
            model._check_inputs = lambda R: None
            trainer = Trainer(model)
            trainer.dict2device = lambda d, device=None: d
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
    for k, v in g["fitted"].items():
        assert abs(fitted[k] - v) <= 1e-5 * abs(v), (k, fitted[k], v)
