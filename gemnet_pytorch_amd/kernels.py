"""Thin launchers: torch tensors in, torch tensors out, one C-ABI call each (include/gemnet_hip.h).

PyTorch only provides device memory and the current stream here.  No fallbacks: CPU tensors or a
missing library raise.  (tests/cpu_kernels.py monkeypatches these launchers with CPU
restatements to exercise the autograd/model logic on machines without a GPU; that emulation
lives under tests/ and is never imported by the product.)
"""
import contextlib
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import GemmArgs, addr, check, ptr, require_device, stream


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"HIP path is fp32; got {t.dtype}")
    return t.contiguous()


def _rowmajor(t):
    """2-D fp32 operand with unit inner stride (row slices of a weight are passed in place)."""
    if t.dtype != torch.float32:
        raise TypeError(f"HIP path is fp32; got {t.dtype}")
    if t.dim() != 2:
        raise ValueError("2-D operand expected")
    if t.shape[0] > 0 and t.shape[1] > 0 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


def _rowsize(t):
    c = 1
    for s in t.shape[1:]:
        c *= int(s)
    return c

def gemm(A, B, trans_a=False, trans_b=False, *, a_dact_pre=None, act=False, pre_out=False,
         mul=None, alpha=1.0, res=None, beta=1.0, gadd1=None, gidx1=None, gadd2=None, gidx2=None,
         ridx=None, res2=None, beta2=1.0, cfg=-1, out=None):
    """C = epilogue(opA(A) @ opB(B)); see gn_gemm_f32 in include/gemnet_hip.h.
    trans_b=False means B is a torch Linear weight (N, K).  Returns C or (C, pre).
    `out`: write C there instead of into a new tensor; `out is res2` accumulates in place (every element is read and
    rewritten by the same thread)."""
    require_device(A, B)
    if (trans_a and trans_b and cfg < 0 and a_dact_pre is None and not act and not pre_out and mul is None
            and res is None and res2 is None and gadd1 is None and gadd2 is None):
        return gemm_tn(A, B, alpha)
    A, B = _rowmajor(A), _rowmajor(B)
    if B.numel() <= (1 << 20) and (B.stride(0) % 4 or B.data_ptr() % 16) and B.shape[1] % 4 == 0:
        # weight slice with an unaligned row pitch (edge embedding: columns of a (128, 262) matrix): a <= 4 MB copy
        # buys the pipelined kernel instead of the scalar-staging generic one (19 us vs 7 us at M = 1024)
        B = B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != Kb:
        raise ValueError(f"gemm shape mismatch: {tuple(A.shape)} ta={trans_a} x {tuple(B.shape)} tb={trans_b}")
    if out is not None:
        assert tuple(out.shape) == (M, N) and out.dtype == torch.float32 and out.is_contiguous()
    C = out if out is not None else torch.empty((M, N), device=A.device, dtype=torch.float32)
    pre = torch.empty_like(C) if pre_out else None
    a = GemmArgs()
    a.A, a.B, a.C = ptr(A), ptr(B), ptr(C)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), N
    a.trans_a, a.trans_b = int(trans_a), int(trans_b)
    if a_dact_pre is not None:
        a_dact_pre = _f32c(a_dact_pre)
        A = _f32c(A)
        assert a_dact_pre.shape == A.shape
        a.A, a.lda = ptr(A), A.stride(0)
    a.a_dact_pre = ptr(a_dact_pre)
    a.act = int(act)
    a.pre_out = ptr(pre)
    if mul is not None:
        mul = _f32c(mul)
        assert mul.shape == C.shape
    a.mul, a.ldmul = ptr(mul), N
    a.alpha = float(alpha)
    if res is not None:
        res = _f32c(res)
        if ridx is None:
            assert res.shape == C.shape
        else:
            assert res.shape[1] == N and ridx.dtype == torch.int32 and ridx.shape[0] == M
    a.res, a.ldres = ptr(res), N
    a.beta = float(beta)
    a.ridx = ptr(ridx)
    if res2 is not None:
        res2 = _f32c(res2)
        assert res2.shape == C.shape
    a.res2, a.ldres2 = ptr(res2), N
    a.beta2 = float(beta2)
    if gadd1 is not None:
        gadd1 = _f32c(gadd1)
        assert gadd1.shape[1] == N and gidx1.dtype == torch.int32 and gidx1.shape[0] == M
    if gadd2 is not None:
        gadd2 = _f32c(gadd2)
        assert gadd2.shape[1] == N and gidx2.dtype == torch.int32 and gidx2.shape[0] == M
    a.gadd1, a.gidx1 = ptr(gadd1), ptr(gidx1)
    a.gadd2, a.gidx2 = ptr(gadd2), ptr(gidx2)
    a.ldg = N
    # split-K for "weight-gradient shaped" products: few output tiles, very long contraction
    a.splitk, a.splitk_ws = 0, None
    plain = (a_dact_pre is None and not act and not pre_out and mul is None and res is None and res2 is None
             and gadd1 is None and gadd2 is None)
    tiles = ((M + 31) // 32) * ((N + 127) // 128)
    if plain and K >= 2048 and tiles <= 64:
        splitk = max(2, min(256, K // 128, 1024 // tiles))
        ws = torch.empty((splitk, M, N), device=A.device, dtype=torch.float32)
        a.splitk, a.splitk_ws = splitk, ptr(ws)
    check(_lib.load().gn_gemm_f32_cfg(ctypes.byref(a), int(cfg), stream()), "gn_gemm_f32")
    return (C, pre) if pre_out else C



def gemm_tn(A, B, alpha=1.0, splitk=-1):
    """alpha * A^T @ B for A (K,M), B (K,N): weight-gradient shaped products (split-K, deterministic)."""
    require_device(A, B)
    A, B = _rowmajor(A), _rowmajor(B)
    K, M = A.shape
    N = B.shape[1]
    if B.shape[0] != K:
        raise ValueError(f"gemm_tn shape mismatch: {tuple(A.shape)} x {tuple(B.shape)}")
    lib = _lib.load()
    C = torch.empty((M, N), device=A.device, dtype=torch.float32)
    if splitk < 0:
        splitk = lib.gn_gemm_tn_splitk(M, N, K)
    ws = torch.empty((splitk, M, N), device=A.device, dtype=torch.float32) if splitk > 1 else None
    check(lib.gn_gemm_tn_f32(ptr(A), ptr(B), ptr(C), M, N, K, A.stride(0), B.stride(0), N, float(alpha), ptr(ws),
                             int(splitk), stream()), "gn_gemm_tn_f32")
    return C


def gather(x, idx32):
    require_device(x, idx32)
    x = _f32c(x)
    y = torch.empty((idx32.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    check(_lib.load().gn_gather_rows_f32(ptr(x), ptr(idx32), ptr(y), idx32.shape[0], _rowsize(x), stream()),
          "gn_gather_rows_f32")
    return y


def gather_mul(x, idx32, m, scale=1.0):
    """scale * x[idx] (.) m in one launch (gn_gather_mul_f32)."""
    require_device(x, idx32, m)
    x, m = _f32c(x), _f32c(m)
    assert m.shape[0] == idx32.shape[0] and _rowsize(m) == _rowsize(x)
    y = torch.empty_like(m)
    check(_lib.load().gn_gather_mul_f32(ptr(x), ptr(idx32), ptr(m), ptr(y), idx32.shape[0], _rowsize(x), float(scale), stream()),
          "gn_gather_mul_f32")
    return y


def segsum(y, perm, seg_off, n_rows):
    require_device(y, seg_off)
    y = _f32c(y)
    x = torch.empty((n_rows,) + tuple(y.shape[1:]), device=y.device, dtype=torch.float32)
    check(_lib.load().gn_segsum_rows_f32(ptr(y), ptr(perm), ptr(seg_off), ptr(x), n_rows, _rowsize(y), stream()),
          "gn_segsum_rows_f32")
    return x


def segsum_multi(terms, n_rows):
    """x[n] = sum_k sign_k * segsum(y_k, perm_k, seg_k)[n] for up to 4 terms (y, perm, seg_off, sign) sharing the row
    space, rows of <= 4 floats, in one launch (gn_segsum_multi_f32)."""
    ys = [_f32c(t[0]) for t in terms]
    require_device(*ys)
    C = _rowsize(ys[0])
    assert 1 <= len(terms) <= 4 and C <= 4 and all(_rowsize(y) == C for y in ys)
    n = len(terms)
    x = torch.empty((n_rows,) + tuple(ys[0].shape[1:]), device=ys[0].device, dtype=torch.float32)
    P = ctypes.c_void_p * n
    yp = P(*[ptr(y) for y in ys])
    pp = P(*[ptr(t[1]) for t in terms])
    sp = P(*[ptr(t[2]) for t in terms])
    sg = (ctypes.c_float * n)(*[float(t[3]) for t in terms])
    check(_lib.load().gn_segsum_multi_f32(n, yp, pp, sp, sg, ptr(x), n_rows, C, stream()), "gn_segsum_multi_f32")
    return x


USE_NATIVE_CSR = os.environ.get("GEMNET_NATIVE_CSR", "1") == "1"
USE_NATIVE_EXPANDED = os.environ.get("GEMNET_NATIVE_EXPANDED", "1") == "1"     # graph.expanded_csr through gn_expanded_csr_i32


def expanded_csr(perm_e, seg_e, seg_off_of_edge, n_items):
    """(perm int32 (n_items), seg_off int32 (n_rows + 1)) of items sorted by edge, grouped by the row of their edge, from the edge
    CSR (perm_e or None, seg_e) — gn_expanded_csr_i32 (two launches; graph.expanded_csr's torch form was ~13)."""
    require_device(seg_e, seg_off_of_edge)
    assert seg_e.dtype == torch.int32 and seg_off_of_edge.dtype == torch.int32 and (perm_e is None or perm_e.dtype == torch.int32)
    E, n_rows = int(seg_off_of_edge.shape[0]) - 1, int(seg_e.shape[0]) - 1
    dev = seg_e.device
    perm = torch.empty(int(n_items), dtype=torch.int32, device=dev)
    seg = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    ws = torch.empty(E + 1, dtype=torch.int32, device=dev)
    check(_lib.load().gn_expanded_csr_i32(ptr(perm_e), ptr(seg_e), n_rows, ptr(seg_off_of_edge), E, ptr(perm), ptr(seg), ptr(ws),
                                          stream()), "gn_expanded_csr_i32")
    return perm, seg


def csr_build(idx32, n_rows):
    """(perm int32, seg_off int32 (n_rows + 1)) of the int32 keys `idx32` by row, stable (gn_csr_build_i32: one radix sort over
    the significant key bits + one lower-bound launch; the torch form was argsort + gather + arange + searchsorted + casts)."""
    require_device(idx32)
    assert idx32.dtype == torch.int32 and idx32.dim() == 1
    idx32 = idx32.contiguous()
    n = int(idx32.shape[0])
    lib = _lib.load()
    nbytes = int(lib.gn_csr_ws_bytes(n, int(n_rows)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=idx32.device)
    perm = torch.empty(n, dtype=torch.int32, device=idx32.device)
    seg = torch.empty(int(n_rows) + 1, dtype=torch.int32, device=idx32.device)
    check(lib.gn_csr_build_i32(ptr(idx32), n, int(n_rows), ptr(perm), ptr(seg), ptr(ws), nbytes, stream()), "gn_csr_build_i32")
    return perm, seg


def seg_offsets(sorted_idx32, n_rows):
    """seg_off int32 (n_rows + 1) of SORTED int32 keys (gn_seg_offsets_i32)."""
    require_device(sorted_idx32)
    assert sorted_idx32.dtype == torch.int32
    sorted_idx32 = sorted_idx32.contiguous()
    seg = torch.empty(int(n_rows) + 1, dtype=torch.int32, device=sorted_idx32.device)
    check(_lib.load().gn_seg_offsets_i32(ptr(sorted_idx32), int(sorted_idx32.shape[0]), int(n_rows), ptr(seg), stream()),
          "gn_seg_offsets_i32")
    return seg


def nonfinite_flag(x, flag, bit=1):
    """flag[0] |= bit when x holds an inf / NaN (gn_nonfinite_flag_f32; no host read-back: runtime.RangeFlag)."""
    require_device(x, flag)
    x = x.contiguous()
    assert x.dtype == torch.float32 and flag.dtype == torch.int32
    check(_lib.load().gn_nonfinite_flag_f32(ptr(x), x.numel(), ptr(flag), int(bit), stream()), "gn_nonfinite_flag_f32")


def index_padded_t(builder, R, e_cap, t_cap, a_cap, n_groups, deg_bound, staging, bufs, state):
    """The triplets-only index build of `builder` (index_device.DeviceGraphBuilder) for positions R, straight into the
    capacity-sized arrays `bufs` (id_c, id_a, id_swap, id_undir, id3_reduce_ca, id3_expand_ba: int32) with the dummy
    molecule's pad rows behind the batch — no read-back, capturable (gn_index_gpu_padded_t; state: int32[4], see the header)."""
    require_device(R, staging, state)
    assert builder.triplets_only and R.is_contiguous() and tuple(R.shape) == (builder.A, 3)
    assert staging.dtype == torch.int32 and staging.numel() >= 4 * e_cap + 2 * t_cap and state.dtype == torch.int32
    for k in ("id_c", "id_a", "id_swap", "id_undir"):
        assert bufs[k].dtype == torch.int32 and bufs[k].numel() == e_cap and bufs[k].is_contiguous()
    for k in ("id3_reduce_ca", "id3_expand_ba"):
        assert bufs[k].dtype == torch.int32 and bufs[k].numel() == t_cap and bufs[k].is_contiguous()
    check(_lib.load().gn_index_gpu_padded_t(
        ptr(R), int(R.dtype == torch.float64), ptr(builder.mol_off), ptr(builder.sq_off), builder.B, builder.A, builder.nmax,
        builder.sum_n2, builder.cutoff, ptr(builder.ws), int(e_cap), int(t_cap), int(a_cap), int(n_groups), int(deg_bound),
        ptr(staging), ptr(bufs["id_c"]), ptr(bufs["id_a"]), ptr(bufs["id_swap"]), ptr(bufs["id_undir"]),
        ptr(bufs["id3_reduce_ca"]), ptr(bufs["id3_expand_ba"]), ptr(state), stream()), "gn_index_gpu_padded_t")


PADDED_Q_KEYS = ("id_c", "id_a", "id_swap", "id_undir", "id3_reduce_ca", "id3_expand_ba", "id4_int_a", "id4_int_b",
                 "id4_reduce_intm_ca", "id4_expand_intm_db", "id4_reduce_intm_ab", "id4_expand_intm_ab",
                 "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd")


def index_padded_q(builder, R, caps, a_cap, n_groups, deg_bound, staging, bufs, state):
    """The quadruplet index build of `builder` for positions R straight into the capacity-sized arrays `bufs` (the sixteen
    arrays PADDED_Q_KEYS, int32) with the dummy molecule's pad rows — no read-back, capturable (gn_index_gpu_padded_q;
    caps = (e_cap, t_cap, eint_cap, i_cap, q_cap); state: int32[8])."""
    require_device(R, staging, state)
    assert not builder.triplets_only and R.is_contiguous() and tuple(R.shape) == (builder.A, 3)
    e_cap, t_cap, eint_cap, i_cap, q_cap = (int(c) for c in caps)
    assert staging.dtype == torch.int32 and staging.numel() >= 4 * e_cap + 2 * eint_cap
    assert state.dtype == torch.int32 and state.numel() >= 8
    want = dict(zip(PADDED_Q_KEYS, (e_cap,) * 4 + (t_cap,) * 2 + (eint_cap,) * 2 + (i_cap,) * 4 + (q_cap,) * 4))
    for k, n in want.items():
        assert bufs[k].dtype == torch.int32 and bufs[k].numel() == n and bufs[k].is_contiguous(), k
    c_arr = (ctypes.c_void_p * 16)(*[addr(bufs[k]) for k in PADDED_Q_KEYS])
    check(_lib.load().gn_index_gpu_padded_q(
        ptr(R), int(R.dtype == torch.float64), ptr(builder.mol_off), ptr(builder.sq_off), builder.B, builder.A, builder.nmax,
        builder.sum_n2, builder.cutoff, builder.int_cutoff, ptr(builder.ws), e_cap, t_cap, eint_cap, i_cap, q_cap, int(a_cap),
        int(n_groups), int(deg_bound), ptr(staging), c_arr, ptr(state), stream()),
        "gn_index_gpu_padded_q")


def cbf_project_supported(rad, y, W):
    return rad.dim() == 3 and rad.shape[1] * rad.shape[2] <= 64 and W.shape[0] <= 16 and W.shape[1] == rad.shape[1] * rad.shape[2] \
        and y.shape[1] == rad.shape[1]


def cbf_project_fwd(rad, ie32, y, W):
    """out[i, n] = sum_{l, r} rad[ie[i], l, r] y[i, l] W[n, l R + r] (gn_cbf_project_fwd_f32)."""
    require_device(rad, ie32, y, W)
    rad, y, W = _f32c(rad), _f32c(y), _f32c(W)
    I, S, R, N = int(y.shape[0]), int(rad.shape[1]), int(rad.shape[2]), int(W.shape[0])
    out = torch.empty((I, N), device=y.device, dtype=torch.float32)
    check(_lib.load().gn_cbf_project_fwd_f32(ptr(rad), ptr(ie32), ptr(y), ptr(W), ptr(out), I, S, R, N, stream()),
          "gn_cbf_project_fwd_f32")
    return out


def cbf_project_bwd(g, rad, seg_off, y, W):
    """-> (g_rad like rad, g_y like y): the adjoint of cbf_project_fwd for a frozen W (gn_cbf_project_bwd_f32)."""
    require_device(g, rad, seg_off, y, W)
    g, rad, y, W = _f32c(g), _f32c(rad), _f32c(y), _f32c(W)
    E, S, R, N = int(rad.shape[0]), int(rad.shape[1]), int(rad.shape[2]), int(W.shape[0])
    if seg_off.dtype != torch.int32:
        seg_off = seg_off.to(torch.int32)
    assert seg_off.numel() == E + 1
    g_rad, g_y = torch.empty_like(rad), torch.empty_like(y)
    check(_lib.load().gn_cbf_project_bwd_f32(ptr(g), ptr(rad), ptr(seg_off), ptr(y), ptr(W), ptr(g_rad), ptr(g_y), E, S, R, N,
                                             stream()), "gn_cbf_project_bwd_f32")
    return g_rad, g_y


def force_loss(E, Et, F, Ft, w_e, w_f, mask=None, w_f_dev=None):
    """-> (loss (), gE like E, gF like F): loss = w_e sum|E - Et| + w_f [* w_f_dev] sum_a mask_a |F_a - Ft_a|_2 and its
    cotangents in one launch (gn_force_loss_f32)."""
    require_device(E, Et, F, Ft)
    E, Et, F, Ft = _f32c(E), _f32c(Et), _f32c(F), _f32c(Ft)
    assert E.shape == Et.shape and F.shape == Ft.shape and F.dim() == 2 and F.shape[1] == 3
    if mask is not None:
        mask = _f32c(mask)
        assert mask.numel() == F.shape[0]
    loss = torch.empty((), device=E.device, dtype=torch.float32)
    gE, gF = torch.empty_like(E), torch.empty_like(F)
    check(_lib.load().gn_force_loss_f32(ptr(E), ptr(Et), E.numel(), ptr(F), ptr(Ft), F.shape[0], ptr(mask), float(w_e), float(w_f),
                                        ptr(w_f_dev), ptr(loss), ptr(gE), ptr(gF), stream()), "gn_force_loss_f32")
    return loss, gE, gF


def index_poison(x, state):
    """x <- NaN when the index build of this step reported an error (gn_index_poison_f32)."""
    require_device(x, state)
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(_lib.load().gn_index_poison_f32(ptr(x), x.numel(), ptr(state), stream()), "gn_index_poison_f32")


def rbf_aggregate_supported(m, rbf, W):
    return m.shape[1] == 128 and rbf.shape[1] == 16 and tuple(W.shape) == (128, 16)


def rbf_aggregate_fwd(m, rbf, W, perm, seg_off, n_atoms, scale):
    """out[a] = scale * sum_{e -> a} m[e] * (W rbf[e]) in one pass (gn_rbf_aggregate_fwd_f32)."""
    require_device(m, rbf, W)
    m, rbf, W = _f32c(m), _f32c(rbf), _f32c(W)
    out = torch.empty((n_atoms, m.shape[1]), device=m.device, dtype=torch.float32)
    check(_lib.load().gn_rbf_aggregate_fwd_f32(ptr(m), ptr(rbf), ptr(W), ptr(perm), ptr(seg_off), ptr(out), n_atoms,
                                               m.shape[1], rbf.shape[1], float(scale), stream()), "gn_rbf_aggregate_fwd_f32")
    return out


def rbf_aggregate_bwd(g_out, m, rbf, W, id_a32, scale, want_m=True, want_rbf=True, acc_m=None, acc_rbf=None):
    """-> (g_m (E,C) or None, g_rbf (E,R) or None) (gn_rbf_aggregate_bwd_f32).  acc_m / acc_rbf: running gradients
    the contribution is ADDED to in the same pass (and which are returned) instead of new tensors."""
    require_device(g_out, m, rbf, W)
    g_out, m, rbf, W = _f32c(g_out), _f32c(m), _f32c(rbf), _f32c(W)
    for t, like in ((acc_m, m), (acc_rbf, rbf)):
        assert t is None or (t.shape == like.shape and t.dtype == torch.float32 and t.is_contiguous())
    g_m = acc_m if acc_m is not None else (torch.empty_like(m) if want_m else None)
    g_rbf = acc_rbf if acc_rbf is not None else (torch.empty_like(rbf) if want_rbf else None)
    accum = (1 if acc_m is not None else 0) | (2 if acc_rbf is not None else 0)
    check(_lib.load().gn_rbf_aggregate_bwd_f32(ptr(g_out), ptr(m), ptr(rbf), ptr(W), ptr(id_a32), ptr(g_m), ptr(g_rbf),
                                               m.shape[0], m.shape[1], rbf.shape[1], float(scale), accum, stream()),
          "gn_rbf_aggregate_bwd_f32")
    return g_m, g_rbf


def bmm(A, B, ta, tb):
    """(A^T if ta else A) @ (B^T if tb else B), batched over dim 0."""
    require_device(A, B)
    A, B = _f32c(A), _f32c(B)
    b = A.shape[0]
    m, k = (A.shape[2], A.shape[1]) if ta else (A.shape[1], A.shape[2])
    n = B.shape[1] if tb else B.shape[2]
    C = torch.empty((b, m, n), device=A.device, dtype=torch.float32)
    check(_lib.load().gn_bmm_f32(ptr(A), ptr(B), ptr(C), b, m, n, k, int(ta), int(tb), stream()), "gn_bmm_f32")
    return C


def ssilu(x, k):
    require_device(x)
    x = _f32c(x)
    out = torch.empty_like(x)
    check(_lib.load().gn_ssilu_f32(ptr(x), ptr(out), x.numel(), k, stream()), "gn_ssilu_f32")
    return out


def pm(z, k, a=None, b=None, d=None, c=1.0):
    """c * ssilu^(k)(z) * a * b * d  (k = -1: no activation factor; a, b, d optional, all of one shape)."""
    ref = z if z is not None else a
    require_device(ref)
    ts = [None if t is None else _f32c(t) for t in (z, a, b, d)]
    for t in ts:
        assert t is None or t.shape == ref.shape, "pm operands must share one shape"
    out = torch.empty(ref.shape, device=ref.device, dtype=torch.float32)
    check(_lib.load().gn_pm_f32(ptr(ts[0]), int(k), ptr(ts[1]), ptr(ts[2]), ptr(ts[3]), float(c), ptr(out),
                                out.numel(), stream()), "gn_pm_f32")
    return out


def dact_mul(g, z, act, mul, c, want_gmul=False):
    """dz = g*c*(mul or 1)*(ssilu'(z) if act else 1); gmul = g*c*(ssilu(z) if act else z)."""
    require_device(g)
    g = _f32c(g)
    z = None if z is None else _f32c(z)
    mul = None if mul is None else _f32c(mul)
    dz = torch.empty_like(g)
    gmul = torch.empty_like(g) if want_gmul else None
    check(_lib.load().gn_dact_mul_f32(ptr(g), ptr(z), int(bool(act)), ptr(mul), float(c), ptr(dz), ptr(gmul),
                                      g.numel(), stream()), "gn_dact_mul_f32")
    return dz, gmul


def bil_reduce(Y, x, sp):
    """Sm[e,s,c] = sum_{t in seg(e)} Y[t,s] x[g(t),c]."""
    require_device(Y, x)
    Y, x = _f32c(Y), _f32c(x)
    S, C = Y.shape[1], x.shape[1]
    Sm = torch.empty((sp.n_reduce, S, C), device=x.device, dtype=torch.float32)
    check(_lib.load().gn_bil_reduce_f32(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(Sm),
                                        sp.n_reduce, S, C, stream()), "gn_bil_reduce_f32")
    return Sm


# The fused per-atom x-adjoint of the tensor basis (gn_bil_expand_atoms_ang_f32: one workgroup per target atom sums the atom's
# expand rows in LDS, edge by edge — no 1.15 GB of per-quadruplet rows written and read back).  Exact and deterministic
# (tests/test_gpu_kernels.py), but measured SLOWER than the two-pass form on MI355X (profiles/r4_q_atom_blocks.txt: the
# bil_reduce_t family of a GemNet-Q step 3.39-3.54 ms against 3.18 ms): one workgroup per CU (its LDS holds the atom's rows)
# walking ~18 edges behind a barrier each cannot hide what 20 independent waves per CU hide in the two-pass kernels.  Off by
# default; GEMNET_ATOM_BLOCKS=1 selects it.
USE_ATOM_BLOCKS = os.environ.get("GEMNET_ATOM_BLOCKS", "0") == "1"
# Round 6: the same sum with the loops turned inside out (gn_bil_expand_rows_ang_f32: a WAVE owns 32 expand rows of one atom and
# walks the atom's reduce edges with the accumulators in registers — no LDS accumulation, no barrier, rows written once; the
# quadruplet of (edge, row) from a dense per-atom grid built with the index plan, graph.SegmentPlan.row_grid).
USE_ROW_GRID = os.environ.get("GEMNET_ROW_GRID", "1") == "1"
ATOM_BLOCK_MAX_ROWS = 848
# Which angle-form kernels run their products on the fp16 matrix pipe (`arith` = GN_ANG_F16 of the launch): bit 0 = K1 of
# bil_reduce_project (only under the "h3" Dense arithmetic: x unscaled), bit 1 = the angle gradient bil_dy_multi, bit 2 = the
# x-adjoint bil_expand (both under an exact per-edge scale: any magnitude).  Read-only configuration (A/B runs, tests).
GN_ANG_F16 = 1
ANG_F16_MASK = int(os.environ.get("GEMNET_ANG_F16", "7")) & 7


# GemNet-Q x-adjoint: GEMNET_EXPAND_POS=1 writes the per-quadruplet rows in DESTINATION order (SegmentPlan.expand_pos) so that
# the segmented sum reads them contiguously.  Measured (profiles/r5_expand_pos_ab.txt): 11.58-11.64 ms against 11.42-11.44 —
# the scattered 128-byte row writes cost more than the contiguous reads save.  Off.
USE_EXPAND_POS = os.environ.get("GEMNET_EXPAND_POS", "0") == "1"


def bil_reduce_t(Y, D, sp):
    """dx[j,c] = sum_{t: g(t)=j} sum_s Y[t,s] D[r(t),s,c].
    Angle form (GemNet-Q): the row-stationary kernel when the plan carries the per-atom grid (USE_ROW_GRID, the default: no
    per-quadruplet rows in memory), else the per-atom workgroup of round 4 (USE_ATOM_BLOCKS), else per-quadruplet rows + a
    segmented sum.  Spherical basis of the triplets: the atom-grouped kernel; other shapes: the scalar kernels."""
    require_device(Y, D)
    Y, D = _f32c(Y), _f32c(D)
    S, C = D.shape[1], D.shape[2]
    permT, segT = sp.expand.csr
    if is_angle_form(Y, S):
        rg = sp.row_grid if (USE_ROW_GRID and C == 32) else None
        if rg is not None:
            a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0, n_tasks = rg
            dx = torch.empty((sp.n_expand, C), device=Y.device, dtype=torch.float32)     # the tasks partition the rows
            check(_lib.load().gn_bil_expand_rows_ang_f32(ptr(Y), ptr(D), ptr(a_perm), ptr(a_seg), ptr(j_off), ptr(qmap),
                                                         ptr(g_off), ptr(task_atom), ptr(task_row0), int(n_tasks), ptr(dx), S, C,
                                                         (GN_ANG_F16 if ANG_F16_MASK & 4 else 0) | (sp.ROW_TILE << 8), stream()),
                  "gn_bil_expand_rows_ang_f32")
            return dx
        ab = sp.atom_blocks if USE_ATOM_BLOCKS else None
        if ab is not None and 0 < ab[3] <= ATOM_BLOCK_MAX_ROWS and C == 32:
            # quadruplets: the rows of one target atom summed in LDS, no per-quadruplet rows in memory
            a_perm, a_seg, j_off, max_J = ab
            dx = torch.empty((sp.n_expand, C), device=Y.device, dtype=torch.float32)   # j_off partitions the rows: all written
            check(_lib.load().gn_bil_expand_atoms_ang_f32(ptr(Y), ptr(D), ptr(sp.seg_off), ptr(sp.expand.idx32), ptr(a_perm),
                                                          ptr(a_seg), ptr(j_off), ptr(dx), a_seg.shape[0] - 1, int(max_J), S, C,
                                                          stream()), "gn_bil_expand_atoms_ang_f32")
            return dx
        dxt = torch.empty((sp.size, C), device=Y.device, dtype=torch.float32)
        pos = sp.expand_pos if USE_EXPAND_POS and permT is not None else None
        check(_lib.load().gn_bil_expand_ang_f32(ptr(Y), ptr(D), ptr(sp.seg_off), ptr(dxt), sp.n_reduce, S, C,
                                                GN_ANG_F16 if ANG_F16_MASK & 4 else 0, ptr(pos), stream()), "gn_bil_expand_ang_f32")
        # rows written in the order of their expand row: the sum reads them contiguously (same rows, same order of addition)
        return segsum(dxt, None if pos is not None else permT, segT, sp.n_expand)
    if S > 8:
        # tensor basis: per-quadruplet rows grouped by reduce edge (dSm[e] read once per edge), then one CSR sum
        dxt = torch.empty((sp.size, C), device=Y.device, dtype=torch.float32)
        check(_lib.load().gn_bil_expand_f32(ptr(Y), ptr(D), ptr(sp.seg_off), ptr(dxt), sp.n_reduce, S, C, stream()),
              "gn_bil_expand_f32")
        return segsum(dxt, permT, segT, sp.n_expand)
    grp = sp.groups if S == 7 and C == 64 else None
    if grp is not None and 0 < grp[4] * S * C * 4 <= 160 * 1024:
        # triplets: both edges end in the same atom — that atom's dSm blocks are parked in LDS once
        rows, off, kseg, rposT, max_rows = grp
        dx = torch.empty((sp.n_expand, C), device=Y.device, dtype=torch.float32)
        check(_lib.load().gn_bil_reduce_t_grouped_f32(ptr(Y), ptr(D), ptr(rows), ptr(off), ptr(kseg), ptr(permT),
                                                      ptr(rposT), ptr(dx), off.shape[0] - 1, max_rows, S, C, stream()),
              "gn_bil_reduce_t_grouped_f32")
        return dx
    dx = torch.empty((sp.n_expand, C), device=Y.device, dtype=torch.float32)
    check(_lib.load().gn_bil_reduce_t_f32(ptr(Y), ptr(D), ptr(sp.reduce.idx32), ptr(permT), ptr(segT),
                                          ptr(dx), sp.n_expand, S, C, stream()), "gn_bil_reduce_t_f32")
    return dx


def bil_dot(D, x, sp):
    """dY[t,s] = sum_c D[r(t),s,c] x[g(t),c]."""
    require_device(D, x)
    D, x = _f32c(D), _f32c(x)
    S, C = D.shape[1], x.shape[1]
    dY = torch.empty((sp.size, S), device=x.device, dtype=torch.float32)
    check(_lib.load().gn_bil_dot_f32(ptr(D), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(dY),
                                     sp.n_reduce, S, C, stream()), "gn_bil_dot_f32")
    return dY


def bessel_rbf(d, freq, cutoff, p, kd, kf):
    require_device(d, freq)
    d, freq = _f32c(d), _f32c(freq)
    out = torch.empty((d.shape[0], freq.shape[0]), device=d.device, dtype=torch.float32)
    check(_lib.load().gn_bessel_rbf_f32(ptr(d), ptr(freq), ptr(out), d.shape[0], freq.shape[0],
                                        cutoff, p, kd, kf, stream()), "gn_bessel_rbf_f32")
    return out


def sph_radial(d, z, nrm, cutoff, p, kd):
    """z (S,R) float32 roots, nrm (S,R) float64 normalisers (device tensors)."""
    require_device(d, z, nrm)
    d = _f32c(d)
    assert z.dtype == torch.float32 and nrm.dtype == torch.float64 and z.is_contiguous() and nrm.is_contiguous()
    S, R = z.shape
    out = torch.empty((d.shape[0], S, R), device=d.device, dtype=torch.float32)
    check(_lib.load().gn_sph_radial_f32(ptr(d), ptr(z), ptr(nrm), ptr(out), d.shape[0], S, R,
                                        cutoff, p, kd, stream()), "gn_sph_radial_f32")
    return out


def ylm0(theta, S, k):
    require_device(theta)
    theta = _f32c(theta)
    out = torch.empty((theta.shape[0], S), device=theta.device, dtype=torch.float32)
    check(_lib.load().gn_ylm0_f32(ptr(theta), ptr(out), theta.shape[0], S, k, stream()), "gn_ylm0_f32")
    return out


def ylm(theta, phi, S, kt, kp):
    require_device(theta, phi)
    theta, phi = _f32c(theta), _f32c(phi)
    out = torch.empty((theta.shape[0], S * S), device=theta.device, dtype=torch.float32)
    check(_lib.load().gn_ylm_f32(ptr(theta), ptr(phi), ptr(out), theta.shape[0], S, kt, kp, stream()),
          "gn_ylm_f32")
    return out


def edge_basis_fwd(R, idx_c, idx_a, freq, z, nrm, cutoff, p, want_V=False, want_rbf=True):
    """-> D (E,), V (E,3)|None, rbf (E,NR)|None, rad (E,S,NR)   (gemnet.py:261-286 + basis_layers.py:45-49,121-128)."""
    require_device(R, idx_c, idx_a, z, nrm)
    R = _f32c(R)
    E = idx_c.shape[0]
    S, NR = z.shape
    D = torch.empty(E, device=R.device, dtype=torch.float32)
    V = torch.empty((E, 3), device=R.device, dtype=torch.float32) if want_V else None
    rbf = torch.empty((E, NR), device=R.device, dtype=torch.float32) if want_rbf else None
    rad = torch.empty((E, S, NR), device=R.device, dtype=torch.float32)
    if want_rbf:
        freq = _f32c(freq)
    # without rbf the kernel still walks NR "rbf" work items per edge; give it a scratch target
    rbf_t = rbf if want_rbf else torch.empty((E, NR), device=R.device, dtype=torch.float32)
    fr = freq if want_rbf else torch.zeros(NR, device=R.device, dtype=torch.float32)
    check(_lib.load().gn_edge_basis_fwd_f32(ptr(R), ptr(idx_c), ptr(idx_a), ptr(fr), ptr(z), ptr(nrm), ptr(D),
                                            ptr(V), ptr(rbf_t), ptr(rad), E, NR, S, cutoff, p, stream()),
          "gn_edge_basis_fwd_f32")
    return D, V, rbf, rad


def edge_basis_bwd(gD, g_rbf, g_rad, R, idx_c, idx_a, freq, z, nrm, cutoff, p):
    """-> W (E,3) per-edge position gradient: dE/dR = segsum(W, id_a) - segsum(W, id_c)."""
    require_device(R, idx_c, idx_a)
    R = _f32c(R)
    E = idx_c.shape[0]
    S, NR = z.shape
    gD = None if gD is None else _f32c(gD)
    g_rbf = None if g_rbf is None else _f32c(g_rbf)
    g_rad = None if g_rad is None else _f32c(g_rad)
    fr = _f32c(freq) if freq is not None else torch.zeros(NR, device=R.device, dtype=torch.float32)
    W = torch.empty((E, 3), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_edge_basis_bwd_f32(ptr(gD), ptr(g_rbf), ptr(g_rad), ptr(R), ptr(idx_c), ptr(idx_a),
                                            ptr(fr), ptr(z), ptr(nrm), ptr(W), E, NR, S, cutoff, p, stream()),
          "gn_edge_basis_bwd_f32")
    return W


def trip_basis_fwd(R, tc, ta, tb, S, want_theta=False):
    """-> Y (T,S) = Y_l0(angle c<-a->b), theta (T,)|None   (gemnet.py:288-311,420-451 + basis_layers.py:130-131)."""
    require_device(R, tc, ta, tb)
    R = _f32c(R)
    T = tc.shape[0]
    Y = torch.empty((T, S), device=R.device, dtype=torch.float32)
    theta = torch.empty(T, device=R.device, dtype=torch.float32) if want_theta else None
    check(_lib.load().gn_trip_basis_fwd_f32(ptr(R), ptr(tc), ptr(ta), ptr(tb), ptr(Y), ptr(theta), T, S, stream()),
          "gn_trip_basis_fwd_f32")
    return Y, theta


def trip_basis_bwd(gY, R, tc, ta, tb):
    """-> Gc, Gb (T,3): dE/dR_c, dE/dR_b per triplet (dE/dR_a = -(Gc+Gb))."""
    require_device(gY, R)
    gY, R = _f32c(gY), _f32c(R)
    T, S = gY.shape
    Gc = torch.empty((T, 3), device=R.device, dtype=torch.float32)
    Gb = torch.empty((T, 3), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_trip_basis_bwd_f32(ptr(gY), ptr(R), ptr(tc), ptr(ta), ptr(tb), ptr(Gc), ptr(Gb), T, S,
                                            stream()), "gn_trip_basis_bwd_f32")
    return Gc, Gb


def dist_fwd(R, id_c, id_a):
    """D (E,) = |R[id_a] - R[id_c]|   (gemnet.py:261-286)."""
    require_device(R, id_c, id_a)
    R = _f32c(R)
    D = torch.empty(id_c.shape[0], device=R.device, dtype=torch.float32)
    check(_lib.load().gn_dist_fwd_f32(ptr(R), ptr(id_c), ptr(id_a), ptr(D), id_c.shape[0], stream()), "gn_dist_fwd_f32")
    return D


def dist_bwd(gD, R, id_c, id_a):
    """-> W (E,3) = gD dD/dR_a: dE/dR = segsum(W, id_a) - segsum(W, id_c)."""
    require_device(gD, R)
    gD, R = _f32c(gD), _f32c(R)
    W = torch.empty((id_c.shape[0], 3), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_dist_bwd_f32(ptr(gD), ptr(R), ptr(id_c), ptr(id_a), ptr(W), id_c.shape[0], stream()), "gn_dist_bwd_f32")
    return W


def dist_jvp(R, tR, gD, id_c, id_a, want_D=True, want_H=True):
    """Tangent pass of the distances along the position tangent tR (A,3): -> (Ddot (E,) | None, H (E,3) | None) with
    H = d/dR_a [gD dD/dR_a] (tR[a] - tR[c])."""
    require_device(R, tR)
    R, tR = _f32c(R), _f32c(tR)
    E = id_c.shape[0]
    gD = None if gD is None else _f32c(gD)
    Dd = torch.empty(E, device=R.device, dtype=torch.float32) if want_D else None
    H = torch.empty((E, 3), device=R.device, dtype=torch.float32) if want_H else None
    check(_lib.load().gn_dist_jvp_f32(ptr(R), ptr(tR), ptr(gD), ptr(id_c), ptr(id_a), ptr(Dd), ptr(H), E, stream()),
          "gn_dist_jvp_f32")
    return Dd, H


def angle_fwd(R, tc, ta, tb):
    """theta (T,) of the angle c <- a -> b   (gemnet.py:288-311, :420-451)."""
    require_device(R, tc, ta, tb)
    R = _f32c(R)
    th = torch.empty(tc.shape[0], device=R.device, dtype=torch.float32)
    check(_lib.load().gn_angle_fwd_f32(ptr(R), ptr(tc), ptr(ta), ptr(tb), ptr(th), tc.shape[0], stream()), "gn_angle_fwd_f32")
    return th


def angle_bwd(g, R, tc, ta, tb):
    """-> Gc, Gb (T,3) = g dtheta/dR_c, g dtheta/dR_b   (dtheta/dR_a = -(Gc + Gb))."""
    require_device(g, R)
    g, R = _f32c(g), _f32c(R)
    T = tc.shape[0]
    Gc = torch.empty((T, 3), device=R.device, dtype=torch.float32)
    Gb = torch.empty((T, 3), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_angle_bwd_f32(ptr(g), ptr(R), ptr(tc), ptr(ta), ptr(tb), ptr(Gc), ptr(Gb), T, stream()),
          "gn_angle_bwd_f32")
    return Gc, Gb


def angle_jvp(R, tR, g, tc, ta, tb, want_theta=True, want_H=True):
    """Tangent pass of the angles along tR (A,3): -> (thdot (T,) | None, Hc, Hb (T,3) | None): the directional
    derivative of the first adjoint g dtheta/d(R_c, R_b)."""
    require_device(R, tR)
    R, tR = _f32c(R), _f32c(tR)
    T = tc.shape[0]
    g = None if g is None else _f32c(g)
    thd = torch.empty(T, device=R.device, dtype=torch.float32) if want_theta else None
    Hc = torch.empty((T, 3), device=R.device, dtype=torch.float32) if want_H else None
    Hb = torch.empty((T, 3), device=R.device, dtype=torch.float32) if want_H else None
    check(_lib.load().gn_angle_jvp_f32(ptr(R), ptr(tR), ptr(g), ptr(tc), ptr(ta), ptr(tb), ptr(thd), ptr(Hc), ptr(Hb), T,
                                       stream()), "gn_angle_jvp_f32")
    return thd, Hc, Hb


class ChainProgram:
    """A program for gn_chain_f32: ops over the three LDS slots of a row tile (see include/gemnet_hip.h).
    Operands named `mul/res/res2` are either an int (LDS slot) or a tensor (global (M,N))."""
    NSLOT = 2

    def __init__(self, M):
        self.M = int(M)
        self.ops = []

    def load(self, slot, src, rows=None, alpha=1.0, y2=-1, alpha2=1.0, Z2=None, mode2=0, add2=None):
        """slot <- src[rows] * alpha; optional second tensor y2 (slot) <- that * alpha2 * phi2(Z2) (as for gemm)
        [+ add2: a source term, see `source`]."""
        self.ops.append(dict(kind="load", slot=slot, src=src, rows=rows, alpha=float(alpha), y2=int(y2),
                             alpha2=float(alpha2), Z2=Z2, mode2=int(mode2), add2=add2))

    def scale(self, dst, a, alpha=1.0, Z=None, out=None, width=None, mode=0, add=None):
        """dst <- a * alpha * phi(Z) [+ add]: mode 0 phi = ssilu', 1 identity (Hadamard with Z), 2 ssilu;
        `add`: a source term (see `source`; its phis(Z) uses this op's Z)."""
        self.ops.append(dict(kind="scale", slot=dst, a_slot=a, alpha=float(alpha), Z=Z, out=out, width=width,
                             mode=int(mode), add=add))

    @staticmethod
    def source(P, Q=None, d2=False, alpha=1.0):
        """The extra summand  alpha * (ssilu''(Z) if d2 else 1) * P * Q  of the second-order sweeps (gn_chain_op.src_*);
        Z is the factor tensor of the op (stage) it is attached to.  P, Q: global (M,N) tensors (Q optional)."""
        return dict(P=P, Q=Q, mode=1 if d2 else 0, alpha=float(alpha))

    def gemm(self, W, a_slot, y_slot=-1, act=False, alpha=1.0, gadd1=None, gidx1=None, gadd2=None, gidx2=None,
             pre_out=None, mul=None, res=None, res_rows=None, beta=1.0, res2=None, beta2=1.0, out=None, packed=None,
             mul_mode=1, y2=-1, y2_src=0, alpha2=1.0, Z2=None, mode2=0, out2=None, add=None, add2=None, pre_deriv=False):
        """W: the (N,K) fp32 weight; `packed`: its split fragment form (pack_weight_split(W), format of the launch mode) if the caller
        caches it — the split-operand kernel packs on the fly otherwise.
        mul_mode (global `mul` only): 1 identity, 2 ssilu'(mul), 3 ssilu(mul).
        Second output: y2 (slot) / out2 (global) <- (y2_src ? activation output : final y) * alpha2 * phi2(Z2),
        phi2 by mode2: 0 ssilu', 1 identity, 2 ssilu (Z2 None: plain scale).
        add / add2: a source term (`source`) added to y after the mul and alpha stages (its ssilu'' factor reads the
        GLOBAL mul operand) / to the second output (ssilu'' of Z2); at most one of them per op.
        pre_deriv (with act and pre_out; split-operand kernel only): `pre_out` receives ssilu'(z) instead of z."""
        assert add is None or add2 is None, "one source term per op"
        assert not pre_deriv or (act and pre_out is not None)
        self.ops.append(dict(kind="gemm", W=W, packed=packed, a_slot=a_slot, slot=y_slot, act=bool(act), alpha=float(alpha),
                             pre_deriv=bool(pre_deriv),
                             add=add, add2=add2,
                             mul_mode=int(mul_mode), y2=int(y2), y2_src=int(y2_src), alpha2=float(alpha2), Z2=Z2,
                             mode2=int(mode2), out2=out2,
                             gadd1=gadd1, gidx1=gidx1, gadd2=gadd2, gidx2=gidx2, pre_out=pre_out, mul=mul,
                             res=res, res_rows=res_rows, beta=float(beta), res2=res2, beta2=float(beta2), out=out))

    def store(self, slot, out):
        self.ops.append(dict(kind="store", slot=slot, out=out))


def fuse_program(prog):
    """Peephole pass over a ChainProgram: SCALE ops that directly follow the GEMM producing their operand move into
    that GEMM's epilogue (every stand-alone SCALE is a full LDS round trip + barrier: 4-5 us at 18 k rows, and the
    adjoint of a residual layer used to carry three of them per two GEMMs).  Rewrites, with g the GEMM writing slot y:
      A  g; scale(y <- y * a * phi(Z))            ->  g(mul = Z, mul_mode = phi, alpha *= a)       [g: plain epilogue]
      B  g; [park]; scale(y <- y * c) ...          ->  g(final factor * c); park alpha / c
         g; scale(y <- y * c, out = T)             ->  g(final factor * c, out = T)
      C  g; ...; scale(o <- y * a2 * phi2(Z2)[, out2])  ->  g(y2 = o, alpha2 = a2, Z2, mode2, out2)
    The result computes the same values (tests/test_ops_cpu.py runs both forms through the interpreter)."""
    ops = [dict(o) for o in prog.ops]
    out = []
    i = 0
    while i < len(ops):
        g = ops[i]
        i += 1
        out.append(g)
        if g["kind"] == "load" and g["slot"] in (0, 1):
            # a LOAD produces its slot like a GEMM does: [park]; y *= c; o = y * a2 * phi2(Z2) ride on it (B, C)
            y, N = g["slot"], g["src"].shape[1]
            parks = []
            while i < len(ops) and ops[i]["kind"] == "scale" and ops[i]["slot"] == 2 and ops[i]["a_slot"] == y:
                parks.append(ops[i])
                i += 1
            while (i < len(ops) and ops[i]["kind"] == "scale" and ops[i]["slot"] == y and ops[i]["a_slot"] == y
                   and (ops[i]["width"] or N) == N and ops[i]["Z"] is None and ops[i]["out"] is None
                   and ops[i].get("add") is None and ops[i]["alpha"] != 0.0 and g["y2"] < 0):
                g["alpha"] *= ops[i]["alpha"]
                for pk in parks:
                    pk["alpha"] /= ops[i]["alpha"]
                i += 1
            if i < len(ops) and g["y2"] < 0:
                sc = ops[i]
                if (sc["kind"] == "scale" and sc["slot"] in (0, 1) and sc["slot"] != y and sc["a_slot"] == y
                        and (sc["width"] or N) == N and sc["out"] is None):
                    g.update(y2=sc["slot"], alpha2=sc["alpha"], Z2=sc["Z"], mode2=sc.get("mode", 0), add2=sc.get("add"))
                    i += 1
            out.extend(parks)
            continue
        if g["kind"] != "gemm" or g["slot"] not in (0, 1):
            continue
        y, N = g["slot"], g["W"].shape[0]

        def is_scale(o, dst, src):
            return o["kind"] == "scale" and o["slot"] == dst and o["a_slot"] == src and (o["width"] or N) == N

        def fold_factor(c):
            if g["res2"] is not None:
                g["beta2"] *= c
            elif g["res"] is not None:
                g["beta"] *= c
            else:
                g["alpha"] *= c

        # A: activation-derivative / Hadamard factor of a plain GEMM
        if (i < len(ops) and is_scale(ops[i], y, y) and ops[i]["Z"] is not None
                and g["mul"] is None and g["res"] is None and g["res2"] is None and g["out"] is None and g["y2"] < 0
                and g.get("out2") is None and g.get("add") is None and g.get("add2") is None):
            sc = ops[i]
            g["mul"], g["mul_mode"] = sc["Z"], {0: 2, 1: 1, 2: 3}[sc.get("mode", 0)]
            g["alpha"] *= sc["alpha"]
            g["out"] = sc["out"]          # the value after the mul / alpha (/ source) stages
            g["add"] = sc.get("add")      # y = a phi(Z) alpha + source: the GEMM's stage-1 source term
            i += 1
        parks = []
        while i < len(ops) and ops[i]["kind"] == "scale" and ops[i]["slot"] == 2 and ops[i]["a_slot"] == y:
            parks.append(ops[i])
            i += 1
        # B: in-place plain scales of y
        while (i < len(ops) and is_scale(ops[i], y, y) and ops[i]["Z"] is None and ops[i]["alpha"] != 0.0
               and ops[i].get("add") is None and g.get("add") is None and g["y2"] < 0 and g.get("out2") is None):
            sc = ops[i]
            if sc["out"] is not None:
                if g["out"] is not None:
                    break
                g["out"] = sc["out"]
            elif g["out"] is not None:
                break          # g already stores the unscaled value
            fold_factor(sc["alpha"])
            for pk in parks:
                pk["alpha"] /= sc["alpha"]
            i += 1
        # C: a second tensor derived from y into the other slot
        if i < len(ops) and g["y2"] < 0 and g.get("out2") is None:
            sc = ops[i]
            o = sc.get("slot")
            used = {y} | {x for x in (g["mul"], g["res"], g["res2"]) if isinstance(x, int)}
            if (sc["kind"] == "scale" and o in (0, 1) and o not in used and sc["a_slot"] == y and (sc["width"] or N) == N
                    and not (sc.get("add") is not None and g.get("add") is not None)):
                g.update(y2=o, y2_src=0, alpha2=sc["alpha"], Z2=sc["Z"], mode2=sc.get("mode", 0), out2=sc["out"],
                         add2=sc.get("add"))
                i += 1
        out.extend(parks)
    fused = ChainProgram(prog.M)
    fused.ops = out
    return fused


def _sel(x):
    """(slot, tensor) of a slot-or-global operand."""
    if x is None:
        return -1, None
    if isinstance(x, int):
        return x, None
    return -1, x


# Arithmetic of the chain GEMMs: "f32" = v_mfma_f32_16x16x4_f32 (csrc/chain.hip); "split6" / "split3" / "bf16" =
# bf16 matrix pipe with 6 / 3 / 1 products of split operands (csrc/chain2.hip; split6 is fp32-equivalent).
# nprod codes of gn_chain_split_f32: 6 / 3 / 1 products of bf16 planes; 2 = GN_CHAIN_F16X2, the two-plane fp16 form
# (three products, csrc/chain2.hip "format H").  "h3" needs its weights packed with SPLIT_FORMAT 1.
CHAIN_MODES = {"f32": 0, "split6": 6, "split3": 3, "bf16": 1, "h3": 2}
SPLIT_FORMAT = {"split6": 0, "split3": 0, "bf16": 0, "h3": 1}
# The process DEFAULT (read once; `bench.py --chain-mode` and tests replace it before any model runs).  The mode a launch runs
# in is per-call state, never a mutable global: a forward pass selects it for its own thread (`use_mode`, thread-local — the
# model's `matmul_precision`), every autograd Function records it at forward time (`ctx.mode`) and its backward — which the
# autograd engine runs on ITS thread, possibly long after the forward's context has closed and after another model with
# another precision has run — restores it from there (`ops._in_mode`, `ops_train._sweep_mode`).
DEFAULT_CHAIN_MODE = os.environ.get("GEMNET_CHAIN_MODE", "h3")
_tls = threading.local()


def current_mode():
    """Arithmetic of the chain / bilinear launches issued by this thread right now."""
    return getattr(_tls, "mode", None) or DEFAULT_CHAIN_MODE


@contextlib.contextmanager
def use_mode(mode):
    """Select the arithmetic (CHAIN_MODES) for the launches this THREAD issues inside the block; None keeps the current one."""
    if mode is None:
        yield
        return
    if mode not in CHAIN_MODES:
        raise ValueError(f"matmul_precision must be one of {sorted(CHAIN_MODES)}; got {mode!r}")
    prev = getattr(_tls, "mode", None)
    _tls.mode = mode
    try:
        yield
    finally:
        _tls.mode = prev


# Kernel layout of the "h3" launches: "tall" = csrc/chain2.hip (one 8-wave workgroup per CU on row tiles of <= 80 rows, a wave
# owns 16 columns), "wide" = csrc/chain3.hip (workgroups of 4 waves x 32 columns on row tiles of <= 48 rows, two per CU;
# GN_CHAIN_WIDE).  Same results bit for bit (tests/test_gpu_kernels.py).  Measured on MI355X (profiles/r4_chain_layouts.txt):
# the edge-row programs take the same time in both (a wave of either layout does the same amount of work per op, and two
# co-resident workgroups neither help nor disturb each other), the atom-row programs are 25-35 % slower in the wide one —
# "tall" stays the default; programs that use the parking slot always take it (the library decides).
# "row" (round 6) = csrc/chain4.hip: a wave owns 16 ROWS and all columns, the slots are fp32 register arrays, the fp16 planes of
# every GEMM operand are formed under a fresh row scale (no fp16 range limit on activations, no `h3_hazards`), only the weights
# pass through LDS (GN_CHAIN_ROW; weights packed with GN_SPLIT_F16X2_ROW = 2).  Same arithmetic ("h3": two fp16 planes, three
# products) to fp32 rounding.  Measured slower than "tall" on the batch sizes of BASELINE.json (profiles/r6_chain4_*.txt: 1 133
# row blocks on 1 024 SIMDs put two 16-row waves on one SIMD of every CU, and a wave's ds_read_b128 stream of weight fragments
# runs at ~28 B/clk) — selectable, not the default.
CHAIN_LAYOUT = os.environ.get("GEMNET_CHAIN_LAYOUT", "tall")
GN_CHAIN_WIDE = 0x100
GN_CHAIN_ROW = 0x200
GN_SPLIT_F16X2_ROW = 2
# per-launch tuning of the wide layout (GN_CHAIN_WIDE_ROWS / GN_CHAIN_WIDE_STAGGER bits of `nprod`; 0 = automatic / none)
WIDE_TILE_ROWS = int(os.environ.get("GN_CHAIN_TILE_ROWS", "0"))
WIDE_STAGGER = int(os.environ.get("GN_CHAIN_STAGGER", "0"))
# The fp16 planes of "h3" cover the magnitudes the MODEL fixes (activations, first-order adjoints dE/d.): sweeps whose
# scale follows the caller's loss (S3 / S4 and the energy-only final adjoint of force training, ops_train.py) run in this
# mode instead when the stack's mode is "h3" — bf16 planes have the fp32 exponent range.
CHAIN_MODE_LINEAR = os.environ.get("GEMNET_CHAIN_MODE_LINEAR", "split6")
assert SPLIT_FORMAT.get(CHAIN_MODE_LINEAR) == 0, "GEMNET_CHAIN_MODE_LINEAR: one of the bf16-plane modes"


def linear_mode(mode):
    """Arithmetic of the loss-scaled sweeps of a stack whose forward ran in `mode`."""
    return CHAIN_MODE_LINEAR if mode == "h3" else mode


def split_format(mode=None):
    """Packed-weight format (GN_SPLIT_*) of a chain mode in the current kernel layout; None for the f32 kernel."""
    fmt = SPLIT_FORMAT.get(mode or current_mode())
    return GN_SPLIT_F16X2_ROW if (fmt == 1 and CHAIN_LAYOUT == "row") else fmt


def pack_weight_split(W, trans=False, fmt=None):
    """(N,K) fp32 weight (or, trans, the (K,N) matrix whose transpose is the weight) -> packed planes (uint8) in the
    format of the current chain mode (`fmt`: GN_SPLIT_BF16X3 = 0 three bf16 planes, GN_SPLIT_F16X2 = 1 two fp16 planes)."""
    require_device(W)
    W = _rowmajor(W)
    if fmt is None:
        fmt = split_format() or 0
    N, Kd = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
    nbytes = -(-N // 16) * -(-Kd // 32) * (2 if fmt else 3) * 64 * 16
    out = torch.empty(nbytes, device=W.device, dtype=torch.uint8)
    check(_lib.load().gn_pack_weight_split_fmt(ptr(W), N, Kd, W.stride(0), int(bool(trans)), int(fmt), ptr(out), stream()),
          "gn_pack_weight_split_fmt")
    out._gn_fmt = fmt
    return out


def pack_job_table(entries):
    """Device job table (gn_pack_job) for `pack_weight_split_grouped`: entries = [(W, trans, packed uint8 tensor)]; the
    format of each job is the one its buffer was first packed in (`packed._gn_fmt`).
    -> (table tensor, total_units).  Host -> device copy: not inside a stream capture."""
    import numpy as np
    JOB = np.dtype([("W", "<u8"), ("out", "<u8"), ("N", "<i4"), ("K", "<i4"), ("ldw", "<i4"), ("trans", "<i4"),
                    ("unit_begin", "<i4"), ("fmt", "<i4")])
    assert JOB.itemsize == 40
    jobs = np.zeros(len(entries), dtype=JOB)
    unit = 0
    for i, (W, trans, packed) in enumerate(entries):
        W = _rowmajor(W)
        assert W.data_ptr() == entries[i][0].data_ptr(), "registered weights must have unit inner stride"
        N, Kd = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
        jobs[i] = (W.data_ptr(), packed.data_ptr(), N, Kd, W.stride(0), int(bool(trans)), unit, getattr(packed, "_gn_fmt", 0))
        unit += -(-N // 16) * -(-Kd // 32) * 64
    dev = entries[0][0].device
    table = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    return table, unit


def pack_weight_split_grouped(table, n_jobs, total_units):
    check(_lib.load().gn_pack_weight_split_grouped(ptr(table), int(n_jobs), int(total_units), stream()),
          "gn_pack_weight_split_grouped")


def chain_split_supported(prog):
    """What gn_chain_split_f32 accepts: N % 16 == 0, slot 2 only as a parking slot."""
    for o in prog.ops:
        if o["kind"] == "gemm":
            N = o["W"].shape[0]
            if N % 16 or o["a_slot"] > 1 or o["y2"] > 1:
                return False
        elif o["kind"] == "scale":
            if o["a_slot"] > 1 or (o["slot"] == 2 and (o["Z"] is not None or o["out"] is not None)):
                return False
        elif o["slot"] > 1 or o.get("y2", -1) > 1:
            return False
    return True


def chain_row_supported(prog):
    """What the row-resident layout (csrc/chain4.hip) takes on top of chain_split_supported: K and every width a multiple of 16."""
    for o in prog.ops:
        if o["kind"] == "gemm":
            if o["W"].shape[1] % 16:
                return False
        elif o["kind"] == "load":
            if o["src"].shape[1] % 16:
                return False
        elif o["kind"] == "store":
            if o["out"].shape[1] % 16:
                return False
        else:
            Z, out = o["Z"], o["out"]
            w = o["width"] or (Z.shape[1] if Z is not None else out.shape[1])
            if w % 16:
                return False
    return True


def chain_is_linear(prog):
    """No activation inside: an adjoint / tangent program, every LDS-resident value is linear in the loaded rows."""
    return not any(o["kind"] == "gemm" and o["act"] for o in prog.ops)


def h3_hazards(prog):
    """Ops of a LINEAR program that the row-scaled fp16 form ("h3", csrc/chain2.hip) cannot take: the scale of a row
    in LDS is fixed by the LOAD that brought it in and inherited by what is computed from it, so an op that ADDS a
    global tensor of foreign magnitude (gathered rows, a global residual, a second-order source term) into a value that
    stays in LDS may leave the fp16 range.  Non-linear programs (activations: O(1) by the model's normalisation) carry no
    row scale and have no such restriction.  -> list of op indices."""
    if not chain_is_linear(prog):
        return []
    bad = []
    for i, o in enumerate(prog.ops):
        if o["kind"] == "gemm":
            stays = o["slot"] in (0, 1) or o.get("y2", -1) >= 0
            foreign = (o["gadd1"] is not None or o["gadd2"] is not None or o.get("add") is not None
                       or o.get("add2") is not None
                       or any(o[k] is not None and not isinstance(o[k], int) for k in ("res", "res2")))
            if stays and foreign:
                bad.append(i)
        elif o["kind"] == "scale" and o["slot"] in (0, 1) and o.get("add") is not None:
            bad.append(i)
        elif o["kind"] == "load" and o.get("add2") is not None:
            bad.append(i)
    return bad


_CHAIN_PACK = None


def _chain_packer():
    """(struct.Struct of one gn_chain_op, default field values, offset of ops[0], sizeof(op), sizeof(args)) derived from
    the ctypes mirror — `chain()` fills the argument block with one pack_into per op instead of ~50 ctypes attribute
    stores (171 us -> ~60 us of host time per launch: the eager / dynamic-shape path issues 50 of them per forward+force).
    The field indices become module globals F_<name>."""
    global _CHAIN_PACK
    if _CHAIN_PACK is None:
        import struct
        from ._lib import ChainArgs, ChainOp
        code = {ctypes.c_int: "i", ctypes.c_float: "f", ctypes.c_void_p: "P"}
        names, fmt = [], "@"
        for name, ct in ChainOp._fields_:
            names.append(name)
            fmt += code[ct]
        st = struct.Struct(fmt)
        # native alignment of `struct` == the C / ctypes layout: every field lands on its ctypes offset
        probe = bytearray(ctypes.sizeof(ChainOp))
        vals = list(range(1, len(names) + 1))
        st.pack_into(probe, 0, *[float(v) if c == ctypes.c_float else v for v, (_, c) in zip(vals, ChainOp._fields_)])
        chk = ChainOp.from_buffer_copy(bytes(probe))
        assert all(getattr(chk, n) == (float(v) if c == ctypes.c_float else v)
                   for v, (n, c) in zip(vals, ChainOp._fields_)) and st.size <= ctypes.sizeof(ChainOp)
        default = [0.0 if c == ctypes.c_float else 0 for _, c in ChainOp._fields_]
        for i, n in enumerate(names):
            globals()["F_" + n] = i
            if n in ("slot", "a_slot", "mul_slot", "res_slot", "res2_slot", "y2_slot"):
                default[i] = -1
        _CHAIN_PACK = (st, default, ChainArgs.ops.offset, ctypes.sizeof(ChainOp), ctypes.sizeof(ChainArgs),
                       struct.Struct("@ii"))
    return _CHAIN_PACK


def _mat(t, cols=None):
    """Address of a contiguous fp32 2-D device tensor (the chain kernel's global operands)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()):
        require_device(t)
        raise AssertionError("chain operands: contiguous fp32 2-D")
    if cols is not None and t.shape[1] != cols:
        raise AssertionError((tuple(t.shape), cols))
    return addr(t)


def _source(v, src, stage, cols):
    v[F_src_stage], v[F_src_mode], v[F_src_alpha] = stage, int(src["mode"]), float(src["alpha"])
    v[F_srcP] = _mat(src["P"], cols)
    v[F_srcQ] = _mat(src["Q"], cols) if src["Q"] is not None else 0


def chain(prog, mode=None):
    """Run a ChainProgram (one launch)."""
    from ._lib import GN_CHAIN_MAX_OPS, GN_OP_GEMM, GN_OP_LOAD, GN_OP_SCALE, GN_OP_STORE
    ops = prog.ops
    if len(ops) > GN_CHAIN_MAX_OPS:
        raise ValueError("chain program too long")
    mode = mode or current_mode()
    nprod = CHAIN_MODES[mode]
    if nprod and not chain_split_supported(prog):
        nprod = 0
    fmt = SPLIT_FORMAT.get(mode, 0)
    row = nprod == CHAIN_MODES["h3"] and CHAIN_LAYOUT == "row"
    if row and not chain_row_supported(prog):
        raise RuntimeError("chain: the row-resident layout needs K and every width to be a multiple of 16 "
                           "(GEMNET_CHAIN_LAYOUT=tall runs other shapes)")
    if row:
        fmt = GN_SPLIT_F16X2_ROW
    if nprod == CHAIN_MODES["h3"] and not row and h3_hazards(prog):
        raise RuntimeError("chain: this linear program adds a global tensor into an LDS-resident value "
                           f"(ops {h3_hazards(prog)}): not representable in the row-scaled fp16 form 'h3' — "
                           "launch it in kernels.linear_mode('h3')")
    st, default, ops_off, op_size, args_size, hdr = _chain_packer()
    keep = []
    buf = bytearray(args_size)
    M = prog.M
    hdr.pack_into(buf, 0, M, len(ops))
    has_src = False
    off = ops_off
    for o in ops:
        v = default[:]
        kind = o["kind"]
        if kind == "gemm":
            W = o["W"]
            N, Kd = W.shape
            if nprod:
                Wp = o.get("packed")     # with the packed planes given, `W` only carries the shape (any strides)
                if Wp is None:
                    _mat(W)
                    Wp = pack_weight_split(W, fmt=fmt)
                elif getattr(Wp, "_gn_fmt", 0) != fmt:
                    raise RuntimeError(f"chain: weight planes packed in format {getattr(Wp, '_gn_fmt', 0)} handed to a "
                                       f"launch in mode {mode!r} (format {fmt})")
                keep.append(Wp)
                v[F_W] = addr(Wp)
            else:
                v[F_W] = _mat(W)
            v[F_kind], v[F_N], v[F_K], v[F_a_slot], v[F_slot] = GN_OP_GEMM, N, Kd, o["a_slot"], o["slot"]
            v[F_act], v[F_alpha], v[F_beta], v[F_beta2] = int(o["act"]) | (2 if o.get("pre_deriv") else 0), o["alpha"], o["beta"], o["beta2"]
            if o.get("pre_deriv") and not nprod:
                raise RuntimeError("chain: pre_deriv needs the split-operand kernel (not CHAIN_MODE 'f32')")
            t = o["gadd1"]
            if t is not None:
                v[F_gadd1], v[F_gidx1] = _mat(t, N), addr(o["gidx1"])
            t = o["gadd2"]
            if t is not None:
                v[F_gadd2], v[F_gidx2] = _mat(t, N), addr(o["gidx2"])
            t = o["pre_out"]
            if t is not None:
                v[F_pre_out] = _mat(t, N)
            t = o["out"]
            if t is not None:
                v[F_out] = _mat(t, N)
            t = o["mul"]
            if t is not None:
                if isinstance(t, int):
                    v[F_mul_slot] = t
                else:
                    v[F_mul_g] = _mat(t, N)
            t = o["res"]
            if t is not None:
                if isinstance(t, int):
                    v[F_res_slot] = t
                else:
                    v[F_res_g] = _mat(t, N)
            t = o["res2"]
            if t is not None:
                if isinstance(t, int):
                    v[F_res2_slot] = t
                else:
                    v[F_res2_g] = _mat(t, N)
            t = o["res_rows"]
            if t is not None:
                v[F_res_rows] = addr(t)
            mm = o["mul_mode"]
            v[F_mul_mode], v[F_y2_slot], v[F_y2_src], v[F_mode2], v[F_alpha2] = mm, o["y2"], o["y2_src"], o["mode2"], o["alpha2"]
            t = o["Z2"]
            if t is not None:
                v[F_Z2] = _mat(t, N)
            t = o["out2"]
            if t is not None:
                v[F_out2] = _mat(t, N)
            if mm > 1:
                assert v[F_mul_slot] < 0 and v[F_mul_g], "mul_mode applies to a global mul operand"
            t = o.get("add")
            if t is not None:
                _source(v, t, 1, N)
                has_src = True
            t = o.get("add2")
            if t is not None:
                _source(v, t, 2, N)
                has_src = True
        elif kind == "load":
            src = o["src"]
            sp = _mat(src)
            assert o["rows"] is not None or src.shape[0] == M
            w = src.shape[1]
            v[F_kind], v[F_slot], v[F_width], v[F_ld], v[F_src] = GN_OP_LOAD, o["slot"], w, src.stride(0), sp
            t = o["rows"]
            if t is not None:
                v[F_rows] = addr(t)
            v[F_alpha], v[F_y2_slot], v[F_alpha2], v[F_mode2] = o.get("alpha", 1.0), o.get("y2", -1), o.get("alpha2", 1.0), o.get("mode2", 0)
            t = o.get("Z2")
            if t is not None:
                v[F_Z2] = _mat(t, w)
            t = o.get("add2")
            if t is not None:
                _source(v, t, 2, w)
                has_src = True
        elif kind == "scale":
            Z, out = o["Z"], o["out"]
            w = o["width"] or (Z.shape[1] if Z is not None else out.shape[1])
            v[F_kind], v[F_slot], v[F_a_slot], v[F_width], v[F_ld] = GN_OP_SCALE, o["slot"], o["a_slot"], w, w
            v[F_alpha], v[F_act] = o["alpha"], o.get("mode", 0)
            if Z is not None:
                v[F_src] = _mat(Z, w)
            if out is not None:
                v[F_out] = _mat(out, w)
            t = o.get("add")
            if t is not None:
                _source(v, t, 1, w)
                has_src = True
        else:   # store
            out = o["out"]
            v[F_kind], v[F_slot], v[F_width], v[F_ld], v[F_out] = GN_OP_STORE, o["slot"], out.shape[1], out.stride(0), _mat(out)
        st.pack_into(buf, off, *v)
        off += op_size
    if has_src and not nprod:
        raise RuntimeError("chain programs with second-order source terms run on the split-operand kernel only "
                           "(CHAIN_MODE f32 / an unsupported shape): use the composite training path")
    cbuf = (ctypes.c_char * args_size).from_buffer(buf)
    if row:
        nprod |= GN_CHAIN_ROW
    elif nprod == CHAIN_MODES["h3"] and CHAIN_LAYOUT == "wide":
        if WIDE_TILE_ROWS and (WIDE_TILE_ROWS % 8 or not 8 <= WIDE_TILE_ROWS <= 48):
            raise ValueError("GN_CHAIN_TILE_ROWS: a multiple of 8 in 8..48")
        nprod |= GN_CHAIN_WIDE | ((WIDE_TILE_ROWS // 8) << 12) | ((max(WIDE_STAGGER, 0) & 0xffff) << 16)
    if nprod:
        check(_lib.load().gn_chain_split_f32(ctypes.addressof(cbuf), nprod, stream()), "gn_chain_split_f32")
    else:
        check(_lib.load().gn_chain_f32(ctypes.addressof(cbuf), stream()), "gn_chain_f32")


def bil_train_supported(S, C, I):
    """Shapes for which the bilinear layer has its fused twice-differentiable training form (ops_train._Bilinear2): the
    spherical basis of the triplet interaction (the extended K1 + K2 kernel gn_bil_reduce_project2_f32)."""
    return (S, C, I) == (7, 64, 16)


def bil_reduce_project(Y, x, B, sp, Sm_init=None, B2=None, Sm2=None, want_P=True):
    """Fused K1+K2 -> (Sm (E,S,C), P (E,I,C)); B = rbf_W1 (E,S,I).
    Extended form (gn_bil_reduce_project2_f32, spherical-basis shapes only): Sm starts from `Sm_init` instead of zero,
    P = B^T Sm + B2^T Sm2 (a second K2 term from given blocks), `want_P=False` skips K2 — the tangent sweep of the
    training step: dSm = K1(dY, x) + K1(Y, dx), dP = B^T dSm + dB^T Sm."""
    require_device(Y, x, B)
    Y, x, B = _f32c(Y), _f32c(x), _f32c(B)
    S, C, I = B.shape[1], x.shape[1], B.shape[2]
    Sm = torch.empty((sp.n_reduce, S, C), device=x.device, dtype=torch.float32)
    P = torch.empty((sp.n_reduce, I, C), device=x.device, dtype=torch.float32) if want_P else None
    if Sm_init is not None or B2 is not None or not want_P:
        assert bil_train_supported(S, C, I) and not is_angle_form(Y, S) and (B2 is None) == (Sm2 is None)
        ts = [None if t is None else _f32c(t) for t in (Sm_init, B2, Sm2)]
        assert ts[0] is None or ts[0].shape == Sm.shape
        assert ts[1] is None or (ts[1].shape == B.shape and ts[2].shape == Sm.shape)
        check(_lib.load().gn_bil_reduce_project2_f32(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(B),
                                                     ptr(ts[0]), ptr(ts[1]), ptr(ts[2]), ptr(Sm), ptr(P), sp.n_reduce,
                                                     S, C, I, stream()), "gn_bil_reduce_project2_f32")
        return Sm, P
    if is_angle_form(Y, S):   # Y_lm rebuilt in-kernel from (sin, cos) of the two angles
        # K1 on the fp16 pipe takes x unscaled (fp16 range): only under the fp16-plane Dense arithmetic, whose overflow guard
        # (model/gemnet.py) covers it — a model on "split6" / "f32" keeps the f32-input MFMA here (`arith` is an argument of
        # the launch: no library state)
        arith = GN_ANG_F16 if (ANG_F16_MASK & 1 and current_mode() == "h3") else 0
        check(_lib.load().gn_bil_reduce_project_ang_f32(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(B),
                                                        ptr(Sm), ptr(P), sp.n_reduce, S, C, I, arith, stream()),
              "gn_bil_reduce_project_ang_f32")
        return Sm, P
    check(_lib.load().gn_bil_reduce_project_f32(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(B),
                                                ptr(Sm), ptr(P), sp.n_reduce, S, C, I, stream()),
          "gn_bil_reduce_project_f32")
    return Sm, P


USE_K3_F16 = os.environ.get("GEMNET_K3_F16", "1") == "1"


def bil_fused_fwd(Y, x, B, W2T, sp, alpha=1.0, W2T_planes=None):
    """K1+K2+K3 in one launch -> (Sm (E,S,C), out (E,O)); W2T (O, I*C) = the bilinear weight, k-contiguous.
    W2T_planes: pack_weight_split(W2T, fmt=1) — K3 then runs on the fp16 matrix pipe with split operands.
    Spherical basis only (S, C, I, O) = (7, 64, 16, 64); see `bil_fused_fwd_supported`."""
    require_device(Y, x, B, W2T)
    Y, x, B, W2T = _f32c(Y), _f32c(x), _f32c(B), _f32c(W2T)
    S, C, I, O = Y.shape[1], x.shape[1], B.shape[2], W2T.shape[0]
    Sm = torch.empty((sp.n_reduce, S, C), device=x.device, dtype=torch.float32)
    out = torch.empty((sp.n_reduce, O), device=x.device, dtype=torch.float32)
    if W2T_planes is not None:
        assert getattr(W2T_planes, "_gn_fmt", None) == 1 and W2T_planes.numel() == 4 * 32 * 2 * 64 * 16
    check(_lib.load().gn_bil_fused_fwd_f32(ptr(Y), ptr(x), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(B), ptr(W2T),
                                           ptr(W2T_planes) if USE_K3_F16 else None,
                                           ptr(Sm), ptr(out), sp.n_reduce, S, C, I, O, float(alpha), stream()),
          "gn_bil_fused_fwd_f32")
    return Sm, out


def bil_fused_fwd_supported(S, C, I, O):
    return (S, C, I, O) == (7, 64, 16, 64)


def is_angle_form(Y, S):
    """The tensor basis given as (Q,4) = (sin, cos) of Phi_cab and Theta_cabd instead of the (Q,49) harmonics."""
    return S == 49 and Y.dim() == 2 and Y.shape[1] == 4


def quad_angles_fwd(R, qc, qa, qb, qd):
    """-> ang (Q,4): (sin, cos) of the polar angle Phi_cab and of the azimuth Theta_cabd (gemnet.py:334-418)."""
    require_device(R, qc, qa, qb, qd)
    R = _f32c(R)
    Q = qc.shape[0]
    ang = torch.empty((Q, 4), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_quad_angles_fwd_f32(ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(ang), Q, stream()),
          "gn_quad_angles_fwd_f32")
    return ang


def quad_angles_bwd(g_ang, R, qc, qa, qb, qd, packed=False):
    """g_ang (Q,4) = (dE/dPhi_cab, dE/dTheta_cabd, -, -) -> (Gc, Gb, Gd) (Q,3) each, or packed: Gc and
    Gbd (Q,8) = [Gb xyz, 0, Gd xyz, 0]."""
    require_device(g_ang, R)
    g_ang, R = _f32c(g_ang), _f32c(R)
    Q = qc.shape[0]
    Gc = torch.empty((Q, 3), device=R.device, dtype=torch.float32)
    if packed:
        Gbd = torch.empty((Q, 8), device=R.device, dtype=torch.float32)      # (the kernel writes the two pad lanes itself)
        base = addr(Gbd)
        check(_lib.load().gn_quad_angles_bwd_ld_f32(ptr(g_ang), ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(Gc), 3,
                                                    base, 8, base + 16, 8, Q, stream()), "gn_quad_angles_bwd_ld_f32")
        return Gc, Gbd
    Gb, Gd = torch.empty_like(Gc), torch.empty_like(Gc)
    check(_lib.load().gn_quad_angles_bwd_ld_f32(ptr(g_ang), ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(Gc), 3,
                                                ptr(Gb), 3, ptr(Gd), 3, Q, stream()), "gn_quad_angles_bwd_ld_f32")
    return Gc, Gb, Gd


def quad_angles_jvp(R, tR, qc, qa, qb, qd):
    """tang (Q,4) = (dPhi_cab . tR, dTheta_cabd . tR, 0, 0): the tangents of the two quadruplet angles along the position
    tangent tR (A,3) — the double backward of `quad_angles_bwd` (gn_quad_angles_jvp_f32)."""
    require_device(R, tR, qc, qa, qb, qd)
    R, tR = _f32c(R), _f32c(tR)
    Q = qc.shape[0]
    tang = torch.empty((Q, 4), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_quad_angles_jvp_f32(ptr(R), ptr(tR), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(tang), Q, stream()),
          "gn_quad_angles_jvp_f32")
    return tang


def bil_reduce_project_tan(ang, tang, x, tx, B, tB, Sm, sp, want_P=True):
    """Tangent sweep S3 of the quadruplet bilinear layer in ONE launch (gn_bil_reduce_project_ang_tan_f32):
    Smd[e] = sum_q (dY[q] (x) x[g(q)] + Y[q] (x) tx[g(q)]),  Pd[e] = B[e]^T Smd[e] + tB[e]^T Sm[e];  dY from the angle tangents
    `tang` (Q,4).  tang / tx / tB may be None (at least one of tang, tx) -> (Smd (E,S,C), Pd (E,I,C) | None)."""
    require_device(ang, x, B)
    assert tang is not None or tx is not None
    ang, x, B = _f32c(ang), _f32c(x), _f32c(B)
    tang, tx, tB, Sm = (None if t is None else _f32c(t) for t in (tang, tx, tB, Sm))
    S, C, I = B.shape[1], x.shape[1], B.shape[2]
    assert is_angle_form(ang, S) and (tang is None or tang.shape == ang.shape) and (tx is None or tx.shape == x.shape)
    assert tB is None or (tB.shape == B.shape and Sm is not None and Sm.shape == (sp.n_reduce, S, C))
    Smd = torch.empty((sp.n_reduce, S, C), device=x.device, dtype=torch.float32)
    Pd = torch.empty((sp.n_reduce, I, C), device=x.device, dtype=torch.float32) if want_P else None
    check(_lib.load().gn_bil_reduce_project_ang_tan_f32(ptr(ang), ptr(tang), ptr(x), ptr(tx), ptr(sp.expand.idx32),
                                                        ptr(sp.seg_off), ptr(B), ptr(tB), ptr(Sm), ptr(Smd), ptr(Pd),
                                                        sp.n_reduce, S, C, I, stream()), "gn_bil_reduce_project_ang_tan_f32")
    return Smd, Pd


def bil_reduce_t_tan(ang, tang, D1, D2, sp):
    """x-adjoint of the second adjoint S4 of the quadruplet bilinear layer: dx[j] = sum_{q: g(q) = j} (Y[q] D1[r(q)] + dY[q]
    D2[r(q)]) — per-quadruplet rows by gn_bil_expand_ang_tan_f32, then the CSR sum over the expand rows.  D1 may be None."""
    require_device(ang, tang, D2)
    ang, tang, D2 = _f32c(ang), _f32c(tang), _f32c(D2)
    D1 = None if D1 is None else _f32c(D1)
    S, C = D2.shape[1], D2.shape[2]
    assert is_angle_form(ang, S) and tang.shape == ang.shape and (D1 is None or D1.shape == D2.shape)
    rg = sp.row_grid if (USE_ROW_GRID and C == 32 and sp.ROW_TILE == 64) else None
    if rg is not None:
        # one pass, no per-quadruplet rows in memory (csrc/bilinear_ang.hip: bil_expand_rows_ang_tan_kernel)
        a_perm, a_seg, j_off, qmap, g_off, task_atom, task_row0, n_tasks = rg
        dx = torch.empty((sp.n_expand, C), device=ang.device, dtype=torch.float32)
        check(_lib.load().gn_bil_expand_rows_ang_tan_f32(ptr(ang), ptr(tang), ptr(D1), ptr(D2), ptr(a_perm), ptr(a_seg), ptr(j_off),
                                                         ptr(qmap), ptr(g_off), ptr(task_atom), ptr(task_row0), int(n_tasks),
                                                         ptr(dx), S, C, sp.ROW_TILE, stream()), "gn_bil_expand_rows_ang_tan_f32")
        return dx
    dxt = torch.empty((sp.size, C), device=ang.device, dtype=torch.float32)
    check(_lib.load().gn_bil_expand_ang_tan_f32(ptr(ang), ptr(tang), ptr(D1), ptr(D2), ptr(sp.seg_off), ptr(dxt), sp.n_reduce,
                                                S, C, stream()), "gn_bil_expand_ang_tan_f32")
    permT, segT = sp.expand.csr
    return segsum(dxt, permT, segT, sp.n_expand)


def bil_ang_train_supported(S, C, I):
    """Shapes of the fused twice-differentiable quadruplet bilinear layer in angle form (ops_train._BilinearAng2)."""
    return (S, C, I) == (49, 32, 32)


def bil_dy_multi(dSm_list, x_list, sp, ang=None):
    """dY (T,S) = sum_b sum_c x_b[g(t),c] dSm_b[r(t),s,c] for the blocks b that share one tensor basis (one pass).
    With `ang` (the basis in angle form): the gradient w.r.t. the two angles, (T,4), instead."""
    require_device(*dSm_list, *x_list)
    dSm_list = [_f32c(t) for t in dSm_list]
    x_list = [_f32c(t) for t in x_list]
    nb = len(dSm_list)
    E, S, C = dSm_list[0].shape
    if ang is not None:
        ang = _f32c(ang)
        g_ang = torch.empty((sp.size, 4), device=ang.device, dtype=torch.float32)
        arr = ctypes.c_void_p * nb
        check(_lib.load().gn_bil_dy_multi_ang_f32(arr(*[addr(t) for t in dSm_list]), arr(*[addr(t) for t in x_list]),
                                                  nb, ptr(ang), ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(g_ang), E, S, C,
                                                  GN_ANG_F16 if ANG_F16_MASK & 2 else 0, stream()), "gn_bil_dy_multi_ang_f32")
        return g_ang
    dY = torch.empty((sp.size, S), device=x_list[0].device, dtype=torch.float32)
    arr = ctypes.c_void_p * nb
    check(_lib.load().gn_bil_dy_multi_f32(arr(*[addr(t) for t in dSm_list]), arr(*[addr(t) for t in x_list]),
                                          nb, ptr(sp.expand.idx32), ptr(sp.seg_off), ptr(dY), E, S, C, stream()),
          "gn_bil_dy_multi_f32")
    return dY


def bil_project_bwd(dP, Sm, B, x, sp, dY_accum=None, want_dY=True, gB_accum=None, dSm_accum=None):
    """Fused adjoint of K2 and of K1 w.r.t. Y -> (gB (E,S,I), dSm (E,S,C), dY (T,S)); with `dY_accum` the Y
    gradient is ADDED into that (T,S) buffer (and returned) instead of written to a fresh one; `gB_accum` / `dSm_accum`
    likewise (the cross terms of the training step's second adjoint land on the first-order ones)."""
    require_device(dP, Sm, B, x)
    dP, Sm, B, x = _f32c(dP), _f32c(Sm), _f32c(B), _f32c(x)
    E, S, C = Sm.shape
    I = B.shape[2]
    if gB_accum is not None:
        assert gB_accum.shape == (E, S, I) and gB_accum.is_contiguous() and gB_accum.dtype == torch.float32
    gB = gB_accum if gB_accum is not None else torch.empty((E, S, I), device=x.device, dtype=torch.float32)
    if dSm_accum is not None:
        assert dSm_accum.shape == (E, S, C) and dSm_accum.is_contiguous() and dSm_accum.dtype == torch.float32
        assert (S, C, I) == (7, 64, 16), "dSm accumulation exists for the spherical-basis kernel only"
    dSm = dSm_accum if dSm_accum is not None else torch.empty((E, S, C), device=x.device, dtype=torch.float32)
    if dY_accum is not None:
        assert dY_accum.shape == (sp.size, S) and dY_accum.is_contiguous() and dY_accum.dtype == torch.float32
    if not want_dY:
        dY = None
    else:
        dY = dY_accum if dY_accum is not None else torch.empty((sp.size, S), device=x.device, dtype=torch.float32)
    flags = int(dY_accum is not None) | (2 if gB_accum is not None else 0) | (4 if dSm_accum is not None else 0)
    check(_lib.load().gn_bil_project_bwd_acc_f32(ptr(dP), ptr(Sm), ptr(B), ptr(x), ptr(sp.expand.idx32),
                                                 ptr(sp.seg_off), ptr(gB), ptr(dSm), ptr(dY), E, S, C, I,
                                                 flags, stream()),
          "gn_bil_project_bwd_acc_f32")
    return gB, dSm, dY


def bil_fused_bwd_supported(S, C, I, O):
    return (S, C, I, O) == (7, 64, 16, 64)


def bil_fused_bwd(g, W2, Sm, B, alpha=1.0, gB_accum=None, W2_planes=None):
    """Adjoint of the bilinear tail in one launch -> (gB (E,S,I), dSm (E,S,C)); W2 (I*C, O) = the bilinear weight
    (gn_bil_fused_bwd_f32: dP = alpha g W2^T stays in LDS).  `gB_accum`: running gradient gB is added to (and returned).
    W2_planes: pack_weight_split(W2, fmt=1) — the first product then runs on the fp16 matrix pipe with split operands."""
    require_device(g, W2, Sm, B)
    g, W2, Sm, B = _f32c(g), _f32c(W2), _f32c(Sm), _f32c(B)
    E, S, C = Sm.shape
    I, O = B.shape[2], g.shape[1]
    assert tuple(W2.shape) == (I * C, O) and B.shape[:2] == (E, S) and g.shape[0] == E
    if gB_accum is not None:
        assert gB_accum.shape == (E, S, I) and gB_accum.is_contiguous() and gB_accum.dtype == torch.float32
    gB = gB_accum if gB_accum is not None else torch.empty((E, S, I), device=g.device, dtype=torch.float32)
    dSm = torch.empty((E, S, C), device=g.device, dtype=torch.float32)
    if W2_planes is not None:
        assert getattr(W2_planes, "_gn_fmt", None) == 1 and W2_planes.numel() == 64 * 2 * 2 * 64 * 16
    check(_lib.load().gn_bil_fused_bwd_f32(ptr(g), ptr(W2), ptr(W2_planes) if USE_K3_F16 else None, ptr(Sm), ptr(B), ptr(gB),
                                           ptr(dSm), E, S, C, I, O, float(alpha),
                                           2 if gB_accum is not None else 0, stream()), "gn_bil_fused_bwd_f32")
    return gB, dSm


def quad_basis_fwd(R, qc, qa, qb, qd, S):
    """-> Y (Q, S^2) = Y_lm(Phi_cab, Theta_cabd) from the 4 atoms of every quadruplet (gemnet.py:334-418)."""
    require_device(R, qc, qa, qb, qd)
    R = _f32c(R)
    Q = qc.shape[0]
    Y = torch.empty((Q, S * S), device=R.device, dtype=torch.float32)
    check(_lib.load().gn_quad_basis_fwd_f32(ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(Y), Q, S, stream()),
          "gn_quad_basis_fwd_f32")
    return Y


def quad_basis_bwd(gY, R, qc, qa, qb, qd, S):
    """-> Gc, Gb, Gd (Q,3); dE/dR_a = -(Gc+Gb+Gd)."""
    require_device(gY, R)
    gY, R = _f32c(gY), _f32c(R)
    Q = qc.shape[0]
    Gc, Gb, Gd = (torch.empty((Q, 3), device=R.device, dtype=torch.float32) for _ in range(3))
    check(_lib.load().gn_quad_basis_bwd_f32(ptr(gY), ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(Gc), ptr(Gb),
                                            ptr(Gd), Q, S, stream()), "gn_quad_basis_bwd_f32")
    return Gc, Gb, Gd


def quad_basis_bwd_packed(gY, R, qc, qa, qb, qd, S):
    """-> Gc (Q,3) and Gbd (Q,8) = [Gb xyz, 0, Gd xyz, 0] (one float4 segmented sum reduces both)."""
    require_device(gY, R)
    gY, R = _f32c(gY), _f32c(R)
    Q = qc.shape[0]
    Gc = torch.empty((Q, 3), device=R.device, dtype=torch.float32)
    Gbd = torch.zeros((Q, 8), device=R.device, dtype=torch.float32)
    base = addr(Gbd)
    check(_lib.load().gn_quad_basis_bwd_ld_f32(ptr(gY), ptr(R), ptr(qc), ptr(qa), ptr(qb), ptr(qd), ptr(Gc), 3,
                                               base, 8, base + 16, 8, Q, S, stream()), "gn_quad_basis_bwd_ld_f32")
    return Gc, Gbd
