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


def grad_probes(name, numel, k=4):
    """The k fixed +-1 probe vectors of tests/golden/make_golden.py::grad_probes (seeded by the parameter name): the
    goldens store <probe, gradient> for EVERY parameter (`<tag>.grad_proj`), which pins the gradient's elements."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, 2, size=(k, numel)).astype(np.float64) * 2.0 - 1.0


def check_grad_probes(g, tag, grads_by_name, rtol):
    """|<probe, grad - grad_ref>| <= rtol * ||grad_ref|| for every parameter and probe (a random +-1 projection of an
    error vector e has magnitude ~ ||e||).  Returns the largest ratio for the test's print-out."""
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    ref_proj, ref_norm = g[f"{tag}.grad_proj"], g[f"{tag}.grad_norms"]
    worst = 0.0
    for i, n in enumerate(names):
        gr = grads_by_name.get(n)
        v = np.zeros(ref_proj.shape[1]) if gr is None else \
            grad_probes(n, gr.numel()) @ gr.detach().double().cpu().reshape(-1).numpy()
        bar = rtol * max(float(ref_norm[i]), 1e-6 * float(ref_norm.max()))
        err = float(np.abs(v - ref_proj[i]).max())
        worst = max(worst, err / bar)
        assert err <= bar, (tag, n, err, bar)
    return worst
