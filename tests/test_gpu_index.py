"""Device-resident index construction (csrc/index_gpu.hip) against the oracle restatement of
the reference's DataContainer (bit-exact integers, canonical order) and against the reference's own goldens."""
import numpy as np
import pytest
import torch

from oracle import index_oracle as IO

pytestmark = pytest.mark.gpu


def _check(R, N, triplets_only, cutoff=5.0, int_cutoff=10.0):
    from gemnet_pytorch_amd.index_device import build_indices_device
    ref = IO.build_indices(R, N, cutoff, int_cutoff, triplets_only)
    out = build_indices_device(torch.tensor(R, device="cuda"), N, cutoff, int_cutoff, triplets_only)
    assert sorted(out) == sorted(ref)
    for k, v in ref.items():
        got = out[k].cpu().numpy()
        assert got.dtype == np.int64 and got.shape == v.shape, (k, got.shape, v.shape)
        np.testing.assert_array_equal(got, v, err_msg=k)


@pytest.mark.parametrize("triplets_only", [True, False])
def test_matches_reference_goldens(golden_indices, triplets_only):
    g = golden_indices
    tagc = "T" if triplets_only else "Q"
    for name in [str(n) for n in g["names"]]:
        tag = f"{name}.{tagc}"
        R, N = g[f"{tag}.R"], g[f"{tag}.N"]
        from gemnet_pytorch_amd.index_device import build_indices_device
        out = build_indices_device(torch.tensor(R, device="cuda"), N, 5.0, 10.0, triplets_only)
        keys = [k[len(tag) + 1:] for k in g if k.startswith(tag + ".") and k[len(tag) + 1:] not in ("R", "N")]
        ref = IO.canonicalize({k: g[f"{tag}.{k}"] for k in keys}, triplets_only)
        for k in keys:
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=f"{tag}.{k}")


@pytest.mark.parametrize("triplets_only", [True, False])
def test_coll_shaped_batch_and_ragged_sizes(triplets_only):
    from gemnet_pytorch_amd.synthetic import make_molecule
    mols = [make_molecule(n, 900 + i) for i, n in enumerate((32, 5, 17, 32, 2, 1, 24))]
    R = np.concatenate([m["R"] for m in mols]).astype(np.float32)
    N = np.array([len(m["R"]) for m in mols])
    _check(R, N, triplets_only)


def test_float64_positions_and_cutoff_boundary():
    R = np.array([[0, 0, 0], [5.0, 0, 0], [0, 3.0, 0], [5.0000005, 3.0, 0.0], [2.5, 1.5, 1.0],
                  [0, 0, 0], [5.0000001, 0, 0]], dtype=np.float64)
    N = np.array([5, 2])
    _check(R, N, False)
    _check(R.astype(np.float32), N, False)


def test_no_edges_and_int32_outputs():
    from gemnet_pytorch_amd.index_device import build_indices_device
    R = torch.tensor([[0.0, 0, 0], [7.5, 0, 0]], device="cuda")
    out = build_indices_device(R, [2], 5.0, 10.0, False)
    assert out["batch_seg"].tolist() == [0, 0]
    assert all(v.numel() == 0 for k, v in out.items() if k != "batch_seg")
    out32 = build_indices_device(torch.rand(12, 3, device="cuda") * 4, [12], 5.0, 10.0, True, dtype=torch.int32)
    assert all(v.dtype == torch.int32 for v in out32.values())


def test_model_forward_on_device_built_graph(golden_model):
    """End to end: positions in HBM -> device-built graph -> GemNet forward+force == golden E/F."""
    import ast
    from test_oracle_model import load_case
    from test_gpu_model import build
    from gemnet_pytorch_amd.index_device import build_indices_device
    g = golden_model
    for tag in ("t2", "q1"):
        cfg, params, inputs = load_case(g, tag)
        model = build(cfg, params).eval()
        R = inputs["R"].to("cuda")
        idx = build_indices_device(R, inputs["N"].numpy(), cfg.get("cutoff", 5.0), cfg.get("int_cutoff", 10.0), cfg["triplets_only"])
        dev = dict(Z=inputs["Z"].to("cuda"), R=R, N=inputs["N"].to("cuda"), **idx)
        E, F = model(dev)
        Fref = g[f"{tag}.F"]
        assert float(np.abs(F.detach().cpu().numpy() - Fref).mean()) <= 1e-5 * max(1.0, float(np.abs(Fref).mean()))
        assert float(np.abs(E.detach().cpu().numpy() - g[f"{tag}.E"]).max()) <= 2e-5 * max(1.0, float(np.abs(g[f"{tag}.E"]).max()))


@pytest.mark.parametrize("n,n_rows", [(0, 5), (1, 1), (1000, 7), (18122, 1024), (332072, 18122), (2_000_003, 600_001), (4096, 2 ** 20)])
def test_native_csr_build_equals_the_stable_sort(n, n_rows):
    """gn_csr_build_i32 (csrc/csr.hip: rocPRIM radix sort of (key, position) pairs over the significant key bits + a lower-bound
    launch) against the torch construction it replaces in graph.RowIndex.csr: the permutation of the STABLE argsort and the
    row offsets, bit for bit — also for rows without entries, a single row, and keys that use the top bit of their range."""
    import torch
    from gemnet_pytorch_amd import kernels as K
    g = torch.Generator().manual_seed(n + n_rows)
    keys = torch.randint(0, n_rows, (n,), generator=g, dtype=torch.int64)
    if n > 10:
        keys[::3] = n_rows - 1                                  # heavy duplicates: stability matters
        keys[1::7] = torch.randint(0, max(n_rows // 100, 1), (len(keys[1::7]),), generator=g)
    dev = keys.to("cuda")
    perm, seg = K.csr_build(dev.to(torch.int32), n_rows)
    ref_perm = torch.argsort(dev, stable=True)
    bounds = torch.arange(n_rows + 1, device="cuda")
    ref_seg = torch.searchsorted(dev[ref_perm].contiguous(), bounds)
    assert perm.dtype == torch.int32 and seg.dtype == torch.int32 and seg.shape[0] == n_rows + 1
    assert torch.equal(perm.long(), ref_perm) and torch.equal(seg.long(), ref_seg)
    srt = dev[ref_perm].to(torch.int32)
    assert torch.equal(K.seg_offsets(srt, n_rows).long(), ref_seg)


def test_expanded_csr_equals_the_stable_sort_of_the_item_keys():
    """gn_expanded_csr_i32: items sorted by edge, grouped by the row (atom) of their edge, from the edge CSR — against the
    stable argsort of the item keys and the lower bounds of the sorted keys; edges without items, rows without edges, sorted
    (perm_e = None) and unsorted edge keys."""
    from gemnet_pytorch_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    for E, n_rows, sorted_keys in ((5000, 257, False), (5000, 257, True), (1, 3, False), (40000, 1200, False)):
        row_of_edge = torch.randint(0, n_rows, (E,), generator=g, dtype=torch.int32)
        if n_rows > 4:
            row_of_edge[row_of_edge == 2] = 3                      # a row without edges
        if sorted_keys:
            row_of_edge = torch.sort(row_of_edge).values
        cnt = torch.randint(0, 30, (E,), generator=g, dtype=torch.int32)
        cnt[::7] = 0                                               # edges without items
        so = torch.zeros(E + 1, dtype=torch.int32)
        so[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        T = int(so[-1])
        item_key = torch.repeat_interleave(row_of_edge.long(), cnt.long())
        want_perm = torch.argsort(item_key, stable=True).to(torch.int32)
        want_seg = torch.searchsorted(item_key[want_perm.long()].contiguous(), torch.arange(n_rows + 1)).to(torch.int32)
        rk = row_of_edge.to("cuda")
        if sorted_keys:
            perm_e, seg_e = None, K.seg_offsets(rk, n_rows)
        else:
            perm_e, seg_e = K.csr_build(rk, n_rows)
        perm, seg = K.expanded_csr(perm_e, seg_e, so.to("cuda"), T)
        torch.cuda.synchronize()
        assert torch.equal(perm.cpu(), want_perm) and torch.equal(seg.cpu(), want_seg), (E, n_rows, sorted_keys)
