"""Index-construction restatement vs the reference DataContainer's output (canonicalised)
and the reference's docstring known-answers (data_container.py:526-533, :554-556)."""
import numpy as np
import pytest

from oracle import index_oracle as IO


def test_repeat_blocks_known_answers():
    assert IO.repeat_blocks([1, 3, 2], [3, 2, 3]).tolist() == [0, 0, 0, 1, 2, 3, 1, 2, 3, 4, 5, 4, 5, 4, 5]
    assert IO.repeat_blocks([0, 3, 2], [3, 2, 3]).tolist() == [0, 1, 2, 0, 1, 2, 3, 4, 3, 4, 3, 4]
    assert IO.repeat_blocks([2, 3, 2], [2, 0, 2]).tolist() == [0, 1, 0, 1, 5, 6, 5, 6]


def test_ragged_range_known_answers():
    assert IO.ragged_range([1, 3, 2]).tolist() == [0, 0, 1, 2, 0, 1]
    assert IO.ragged_range([1, 4, 2, 3]).tolist() == [0, 0, 1, 2, 3, 0, 1, 0, 1, 2]


def _cases(g):
    return [str(n) for n in g["names"]]


@pytest.mark.parametrize("variant", ["T", "Q"])
def test_indices_match_reference(golden_indices, variant):
    g = golden_indices
    to = variant == "T"
    keys = IO.INDEX_KEYS_T + ([] if to else IO.INDEX_KEYS_Q)
    for name in _cases(g):
        tag = f"{name}.{variant}"
        R, N = g[f"{tag}.R"], g[f"{tag}.N"]
        ref = IO.canonicalize({k: g[f"{tag}.{k}"] for k in keys}, to)
        mine = IO.build_indices(R, N, 5.0, 10.0, to)
        for k in keys:
            assert mine[k].dtype == np.int64
            assert np.array_equal(mine[k], ref[k]), (tag, k)


def test_invariants(golden_indices):
    """The commented-out asserts of data_container.py:340-344,393-405 hold for the restatement."""
    g = golden_indices
    R, N = g["batch3.Q.R"], g["batch3.Q.N"]
    d = IO.build_indices(R, N, 5.0, 10.0, False)
    ida, idc = d["id_a"], d["id_c"]
    assert np.array_equal(d["id_swap"][d["id_swap"]], np.arange(len(ida)))
    assert np.array_equal(ida[d["id_swap"]], idc)
    r, x = d["id3_reduce_ca"], d["id3_expand_ba"]
    assert np.all(ida[r] == ida[x]) and np.all(idc[r] != idc[x])
    assert np.all(np.diff(r) >= 0)
    rc, xd = d["id4_reduce_ca"], d["id4_expand_db"]
    c, a, b, dd = idc[rc], ida[rc], ida[xd], idc[xd]
    assert np.all(c != b) and np.all(a != dd) and np.all(c != dd)
    assert np.array_equal(rc, d["id4_reduce_intm_ca"][d["id4_reduce_cab"]])
    assert np.array_equal(xd, d["id4_expand_intm_db"][d["id4_expand_abd"]])
    assert np.array_equal(a, d["id4_int_a"][d["id4_reduce_intm_ab"]][d["id4_reduce_cab"]])
    assert np.array_equal(b, d["id4_int_b"][d["id4_expand_intm_ab"]][d["id4_expand_abd"]])


def test_no_edge_case(golden_indices):
    g = golden_indices
    d = IO.build_indices(g["noedge.Q.R"], g["noedge.Q.N"], 5.0, 10.0, False)
    for k in IO.INDEX_KEYS_T[1:] + IO.INDEX_KEYS_Q:
        assert len(d[k]) == 0
    assert d["batch_seg"].tolist() == [0, 0]
