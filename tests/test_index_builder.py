"""The native host index builder (csrc/index_build.cpp via include/gemnet_index.h) against the
reference DataContainer's output (canonicalised; bit-exact integers) and the docstring known-answers."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import index_oracle as IO
from gemnet_pytorch_amd.training import data_container as DC
from gemnet_pytorch_amd.synthetic import make_dataset


@pytest.fixture(scope="module", autouse=True)
def _built():
    import os
    if not os.path.exists(DC.INDEX_LIB_PATH):
        import __graft_entry__ as ge
        ge.build()


def test_numba_helper_known_answers():
    lib = DC._load_index_lib()
    lib.gn_repeat_blocks.restype = ctypes.c_int64
    lib.gn_ragged_range.restype = ctypes.c_int64

    def rb(sizes, reps):
        s, r = np.array(sizes, np.int64), np.array(reps, np.int64)
        out = np.zeros(64, np.int64)
        n = lib.gn_repeat_blocks(s.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p), len(s),
                                 out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(64))
        return out[:n].tolist()

    assert rb([1, 3, 2], [3, 2, 3]) == [0, 0, 0, 1, 2, 3, 1, 2, 3, 4, 5, 4, 5, 4, 5]
    assert rb([0, 3, 2], [3, 2, 3]) == [0, 1, 2, 0, 1, 2, 3, 4, 3, 4, 3, 4]
    assert rb([2, 3, 2], [2, 0, 2]) == [0, 1, 0, 1, 5, 6, 5, 6]
    s = np.array([1, 3, 2], np.int64)
    out = np.zeros(16, np.int64)
    n = lib.gn_ragged_range(s.ctypes.data_as(ctypes.c_void_p), 3, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(16))
    assert out[:n].tolist() == [0, 0, 1, 2, 0, 1]


@pytest.mark.parametrize("variant", ["T", "Q"])
def test_matches_reference_goldens(golden_indices, variant):
    g = golden_indices
    to = variant == "T"
    keys = IO.INDEX_KEYS_T + ([] if to else IO.INDEX_KEYS_Q)
    for name in [str(n) for n in g["names"]]:
        tag = f"{name}.{variant}"
        ref = IO.canonicalize({k: g[f"{tag}.{k}"] for k in keys}, to)
        mine = DC.build_indices(g[f"{tag}.R"], g[f"{tag}.N"], 5.0, 10.0, to)
        for k in keys:
            assert mine[k].dtype == np.int64
            assert np.array_equal(mine[k], ref[k]), (tag, k)


def test_matches_oracle_on_coll_shaped_batch():
    ds = make_dataset(3, 32, config=2)
    a = DC.build_indices(ds["R"], ds["N"], 5.0, 10.0, False)
    b = IO.build_indices(ds["R"], ds["N"], 5.0, 10.0, False)
    for k in b:
        assert np.array_equal(a[k], b[k]), k


def test_float64_positions_follow_float64_distances():
    R = np.array([[0, 0, 0], [5.0000001, 0, 0]], dtype=np.float64)
    assert len(DC.build_indices(R, [2], 5.0, 10.0, True)["id_a"]) == 0          # beyond the cutoff in f64
    assert len(DC.build_indices(R.astype(np.float32), [2], 5.0, 10.0, True)["id_a"]) == 2  # rounds to 5.0f


def test_datacontainer_dict_contract():
    ds = make_dataset(4, 12, config=1)
    dc = DC.DataContainer.from_arrays(ds, 5.0, 10.0, triplets_only=False)
    batch = dc[[0, 2]]
    import torch
    for k in dc.index_keys + ["Z", "N"]:
        assert batch[k].dtype == torch.int64, k
    for k in ("R", "E", "F"):
        assert batch[k].dtype == torch.float32
    assert batch["E"].shape == (2, 1) and batch["R"].shape == (24, 3)
    assert batch["batch_seg"].tolist() == [0] * 12 + [1] * 12
    one = dc[1]
    assert one["N"].tolist() == [12]


def test_single_molecule_subclass_contract_of_ase_calculator():
    """ase_calculator.py:23-99 subclasses DataContainer WITHOUT calling its __init__: it sets index_keys,
    cutoffs, keys, R (float64 ASE positions), Z, N, E, F, N_cumsum, addID, merges get_dtypes() and then
    uses __getitem__(0).  The same pattern must work on the native container."""
    from gemnet_pytorch_amd.training.data_container import DataContainer, INDEX_KEYS_Q, INDEX_KEYS_T

    class OneMolecule(DataContainer):
        def __init__(self, R, Z, cutoff, int_cutoff, triplets_only=False):
            self.index_keys = list(INDEX_KEYS_T) + ([] if triplets_only else list(INDEX_KEYS_Q))
            self.triplets_only, self.cutoff, self.int_cutoff = triplets_only, cutoff, int_cutoff
            self.keys = ["N", "Z", "R", "F", "E"]
            self.R, self.Z = R, Z
            self.N = np.array([len(Z)], dtype=np.int32)
            self.E = np.zeros((1, 1), dtype=np.float32)
            self.F = np.zeros((len(Z), 3), dtype=np.float32)
            self.N_cumsum = np.concatenate([[0], np.cumsum(self.N)])
            self.addID = False
            self.dtypes, more = self.get_dtypes()
            self.dtypes.update(more)

    from gemnet_pytorch_amd.synthetic import make_molecule
    mol = make_molecule(9, 77, box=4.5)
    for triplets_only in (True, False):
        one = OneMolecule(mol["R"].astype(np.float64), mol["Z"], 5.0, 10.0, triplets_only)
        first = one[0]
        assert first["R"].dtype == torch.float32 and first["id_a"].dtype == torch.int64
        assert set(one.index_keys) <= set(first)
        moved = mol["R"].astype(np.float64) + 0.05 * np.random.RandomState(0).standard_normal((9, 3))
        one.R = moved  # Molecule.update
        second = one[0]
        ref = IO.build_indices(moved, np.array([9]), 5.0, 10.0, triplets_only)  # float64 distances
        assert second["id3_reduce_ca"].shape[0] > 0
        for k in one.index_keys:
            np.testing.assert_array_equal(second[k].numpy(), ref[k])
