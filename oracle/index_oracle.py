"""TEST INFRASTRUCTURE (oracle) — numpy restatement of GemNet's id3_*/id4_* index construction.

Follows gemnet/training/data_container.py of the reference:
  * __getitem__ (edges, id_swap, id_undir, batch_seg)   :156-316
  * get_triplets                                          :410-425  (+ :318-338)
  * get_quadruplets                                       :427-489  (+ :354-391)
  * repeat_blocks / ragged_range (numba helpers)          :520-565
No scipy.sparse, no numba.  Output order is CANONICAL: the reference sorts triplets and
quadruplets with numpy's unstable default argsort (:326,:371) so their order inside one
reduce segment is build dependent; here they are lexsorted by (reduce edge, expand edge).
`canonicalize()` brings reference output to the same form for exact integer comparison.
Pinned by tests/golden/index_*.npz and the docstring known-answers (:526-533,:554-556).
"""
import numpy as np

INDEX_KEYS_T = [
    "batch_seg", "id_undir", "id_swap", "id_c", "id_a",
    "id3_expand_ba", "id3_reduce_ca", "Kidx3",
]
INDEX_KEYS_Q = [
    "id4_int_b", "id4_int_a", "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab",
    "id4_expand_abd", "Kidx4", "id4_reduce_intm_ca", "id4_expand_intm_db",
    "id4_reduce_intm_ab", "id4_expand_intm_ab",
]


def repeat_blocks(sizes, repeats):
    """data_container.py:520-546.  sizes=[1,3,2], repeats=[3,2,3] -> [0 0 0 1 2 3 1 2 3 4 5 4 5 4 5]."""
    sizes = np.asarray(sizes, dtype=np.int64)
    repeats = np.asarray(repeats, dtype=np.int64)
    out = np.empty(int((sizes * repeats).sum()), dtype=np.int64)
    start = 0
    oi = 0
    for size, rep in zip(sizes, repeats):
        blk = np.arange(start, start + size)
        for _ in range(rep):
            out[oi:oi + size] = blk
            oi += size
        start += size
    return out


def ragged_range(sizes):
    """data_container.py:548-565.  sizes=[1,3,2] -> [0 0 1 2 0 1]."""
    sizes = np.asarray(sizes, dtype=np.int64)
    if sizes.size == 0:
        return np.zeros(0, dtype=np.int64)
    out = np.empty(int(sizes.sum()), dtype=np.int64)
    start = 0
    for size in sizes:
        out[start:start + size] = np.arange(size)
        start += size
    return out


def _pairs_within(R, cutoff):
    """Row-major (i, j), i != j, with ||R_i - R_j|| <= cutoff evaluated in R's dtype (:255-258)."""
    D = np.linalg.norm(R[:, None, :] - R[None, :, :], axis=-1)
    adj = D <= cutoff
    np.fill_diagonal(adj, False)
    t, s = np.nonzero(adj)  # row-major == scipy CSR nonzero() order
    return t, s


def build_indices(R, N, cutoff, int_cutoff, triplets_only):
    """R (A,3) float32|float64, N (B,) atoms per molecule -> dict of int64 arrays (canonical order)."""
    R = np.asarray(R)
    N = np.asarray(N, dtype=np.int64)
    A = int(N.sum())
    out = {"batch_seg": np.repeat(np.arange(len(N), dtype=np.int64), N)}
    keys = INDEX_KEYS_T[1:] + ([] if triplets_only else INDEX_KEYS_Q)

    ts, ss, its, iss = [], [], [], []
    off = 0
    for n in N:
        Rm = R[off:off + n]
        t, s = _pairs_within(Rm, cutoff)
        ts.append(t + off)
        ss.append(s + off)
        if not triplets_only:
            t, s = _pairs_within(Rm, int_cutoff)
            its.append(t + off)
            iss.append(s + off)
        off += n
    idx_t = np.concatenate(ts) if ts else np.zeros(0, np.int64)
    idx_s = np.concatenate(ss) if ss else np.zeros(0, np.int64)

    if len(idx_t) == 0:  # :282-285
        for k in keys:
            out[k] = np.zeros(0, dtype=np.int64)
        return out

    # undirected edges once (t<s), then the reversed list (:289-293)
    mask = idx_t < idx_s
    t_half, s_half = idx_t[mask], idx_s[mask]
    id_a = np.concatenate([t_half, s_half]).astype(np.int64)  # target
    id_c = np.concatenate([s_half, t_half]).astype(np.int64)  # source
    E = len(id_a)
    half = E // 2
    ind = np.arange(half, dtype=np.int64)
    out["id_undir"] = np.concatenate([ind, ind])
    out["id_swap"] = np.concatenate([ind + half, ind])
    out["id_c"], out["id_a"] = id_c, id_a

    # incoming edges per atom, sorted by source atom (scipy canonical CSR column order)
    order_in = np.lexsort((id_c, id_a))
    in_ptr = np.zeros(A + 1, dtype=np.int64)
    np.add.at(in_ptr, id_a + 1, 1)
    in_ptr = np.cumsum(in_ptr)

    def in_edges(atom):
        return order_in[in_ptr[atom]:in_ptr[atom + 1]]

    # ---- triplets: reduce edge r=(c->a), expand edges x=(b->a), b != c, x ascending ----
    red, exp = [], []
    for r in range(E):
        x = np.sort(in_edges(id_a[r]))
        x = x[id_c[x] != id_c[r]]
        red.append(np.full(len(x), r, dtype=np.int64))
        exp.append(x)
    id3_reduce = np.concatenate(red)
    id3_expand = np.concatenate(exp)
    out["id3_reduce_ca"], out["id3_expand_ba"] = id3_reduce, id3_expand
    out["Kidx3"] = _kidx(id3_reduce)
    if triplets_only:
        return out

    # ---- quadruplets (:427-489) ----
    int_t = np.concatenate(its).astype(np.int64)  # a
    int_s = np.concatenate(iss).astype(np.int64)  # b
    out["id4_int_a"], out["id4_int_b"] = int_t, int_s
    nN_t = (in_ptr[int_t + 1] - in_ptr[int_t])
    nN_s = (in_ptr[int_s + 1] - in_ptr[int_s])
    red_intm_ca = np.concatenate([in_edges(a) for a in int_t]) if len(int_t) else np.zeros(0, np.int64)
    exp_intm_db = np.concatenate([in_edges(b) for b in int_s]) if len(int_s) else np.zeros(0, np.int64)
    out["id4_reduce_intm_ca"] = red_intm_ca.astype(np.int64)
    out["id4_expand_intm_db"] = exp_intm_db.astype(np.int64)
    out["id4_reduce_intm_ab"] = np.repeat(np.arange(len(int_t), dtype=np.int64), nN_t)
    out["id4_expand_intm_ab"] = np.repeat(np.arange(len(int_t), dtype=np.int64), nN_s)

    reduce_cab = repeat_blocks(nN_t, nN_s)
    reduce_ca = red_intm_ca[reduce_cab]
    Nrep = np.repeat(nN_t, nN_s)
    expand_abd = np.repeat(np.arange(len(exp_intm_db), dtype=np.int64), Nrep)
    expand_db = exp_intm_db[expand_abd]
    c, a = id_c[reduce_ca], id_a[reduce_ca]
    b, d = id_a[expand_db], id_c[expand_db]
    m = (c != b) & (a != d) & (c != d)
    reduce_ca, expand_db, reduce_cab, expand_abd = reduce_ca[m], expand_db[m], reduce_cab[m], expand_abd[m]
    o = np.lexsort((expand_db, reduce_ca))
    out["id4_reduce_ca"] = reduce_ca[o]
    out["id4_expand_db"] = expand_db[o]
    out["id4_reduce_cab"] = reduce_cab[o]
    out["id4_expand_abd"] = expand_abd[o]
    out["Kidx4"] = _kidx(out["id4_reduce_ca"])
    return out


def _kidx(sorted_ids):
    """Position inside each run of equal values (np.unique counts -> ragged_range, :329-333)."""
    if len(sorted_ids) == 0:
        return np.zeros(0, dtype=np.int64)
    _, K = np.unique(sorted_ids, return_counts=True)
    return ragged_range(K)


def canonicalize(idx, triplets_only):
    """Bring an index dict (e.g. the reference's output) to canonical within-segment order."""
    out = {k: np.asarray(v).astype(np.int64) for k, v in idx.items()}
    if len(out.get("id3_reduce_ca", [])):
        o = np.lexsort((out["id3_expand_ba"], out["id3_reduce_ca"]))
        out["id3_reduce_ca"] = out["id3_reduce_ca"][o]
        out["id3_expand_ba"] = out["id3_expand_ba"][o]
        out["Kidx3"] = _kidx(out["id3_reduce_ca"])
    if not triplets_only and len(out.get("id4_reduce_ca", [])):
        o = np.lexsort((out["id4_expand_db"], out["id4_reduce_ca"]))
        for k in ("id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd"):
            out[k] = out[k][o]
        out["Kidx4"] = _kidx(out["id4_reduce_ca"])
    return out
