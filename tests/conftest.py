import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCALE_FILE = os.path.join(ROOT, "gemnet_pytorch_amd", "scaling_factors.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_basis():
    return dict(np.load(os.path.join(GOLDEN, "basis.npz")))


@pytest.fixture(scope="session")
def golden_indices():
    return dict(np.load(os.path.join(GOLDEN, "indices.npz")))


@pytest.fixture(scope="session")
def golden_model():
    return dict(np.load(os.path.join(GOLDEN, "model.npz")))


@pytest.fixture(scope="session")
def golden_model2():
    return dict(np.load(os.path.join(GOLDEN, "model2.npz")))
