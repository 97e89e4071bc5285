"""CPU: the host index builder (csrc/index_build.cpp) and the oracle against the REFERENCE's index arrays at the sizes
BASELINE.json names — one 32-atom molecule, one 64-atom molecule (configs[4]'s molecule size), the 32 x 32 batch of configs[1]
as GemNet-T and as GemNet-Q (9.0 M quadruplets) — through sizes + SHA-256 of the canonical arrays
(tests/golden/fullsize_index.json; reference: training/data_container.py:244-489); and the oracle's forward+force against the
reference's float64 E / F for the 64-atom GemNet-T molecule and the configs[1] batch (gemnet/model/gemnet.py:453-615)."""
import numpy as np
import pytest
import torch

from fullsize_common import dataset, digest, load_digests, load_fullsize, params_of, triplets_only
from oracle import gemnet_oracle as GO
from oracle import index_oracle as IO
from gemnet_pytorch_amd.training import data_container as DC

TAGS = ["t64s", "q64s", "tB32", "qB4", "idx32.T", "idx32.Q", "idxB32.Q"]


@pytest.fixture(scope="module", autouse=True)
def _built():
    import os
    if not os.path.exists(DC.INDEX_LIB_PATH):
        import __graft_entry__ as ge
        ge.build()


@pytest.mark.parametrize("tag", TAGS)
def test_host_builder_matches_reference_digest(tag):
    ds, to = dataset(tag), triplets_only(tag)
    mine = DC.build_indices(ds["R"], ds["N"], 5.0, 10.0, to)
    ref = load_digests()[tag]
    got = digest(mine, to)
    assert sorted(got) == sorted(ref)
    for k in ref:
        assert got[k] == ref[k], (tag, k, got[k]["n"], ref[k]["n"])


@pytest.mark.parametrize("tag", ["t64s", "q64s", "tB32", "qB4", "idx32.Q"])
def test_oracle_matches_reference_digest(tag):
    ds, to = dataset(tag), triplets_only(tag)
    got = digest(IO.build_indices(ds["R"], ds["N"], 5.0, 10.0, to), to)
    assert got == load_digests()[tag]


def test_fixture_inputs_are_the_generated_ones():
    g = load_fullsize()
    for tag in ("t64s", "q64s", "tB32", "qB4"):
        ds = dataset(tag)
        for k in ("N", "Z", "R"):
            assert np.array_equal(g[f"{tag}.{k}"], ds[k]), (tag, k)
        assert abs(float(np.abs(g[f"{tag}.F"]).mean()) - 1.0) < 1e-9      # unit-force fixtures: the 1e-5 eV/A bar is literal
        # the reference's own float32 forces ride along (make_golden.py::run_fullsize): its fp32 rounding on the fixture
        noise = float(np.abs(g[f"{tag}.F32"].astype(np.float64) - g[f"{tag}.F"]).mean())
        print(f"{tag}: reference float32 vs float64 force MAE {noise:.3e}")
        assert (noise < 1e-5) == (tag in ("t64s", "tB32"))          # GemNet-T within the bar, GemNet-Q 1.5e-4 .. 2.1e-4


@pytest.mark.parametrize("tag", ["t64s", "tB32"])
def test_oracle_forward_force_matches_reference(tag):
    g = load_fullsize()
    cfg, params = params_of(g, tag, dtype=torch.float64)
    ds = dataset(tag)
    idx = IO.build_indices(ds["R"], ds["N"], 5.0, 10.0, True)
    inputs = dict(Z=torch.tensor(ds["Z"]).long(), R=torch.tensor(ds["R"]).double(), N=torch.tensor(ds["N"]).long(),
                  **{k: torch.tensor(v) for k, v in idx.items()})
    E, F = GO.forward(cfg, params, inputs)
    e = float(np.abs(E.detach().numpy() - g[f"{tag}.E"]).max())
    f = float(np.abs(F.detach().numpy() - g[f"{tag}.F"]).mean())
    print(f"{tag}: oracle vs reference (float64) energy err {e:.2e}, force MAE {f:.2e}")
    assert e <= 1e-8 * max(1.0, float(np.abs(g[f"{tag}.E"]).max())) and f <= 1e-9
