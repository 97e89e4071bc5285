"""Index construction on the device: `build_indices_device` returns the same dict of index
tensors as `DataContainer.__getitem__` (gemnet/training/data_container.py:156-408; canonical within-segment
order, see include/gemnet_hip.h), built by csrc/index_gpu.hip from positions that already live in HBM — the
MD loop of ase_calculator.py:155-158 rebuilds the graph every step.

    idx = build_indices_device(R, N, cutoff, int_cutoff, triplets_only)      # R (A,3) cuda float32|float64
    inputs = dict(Z=Z, R=R.float(), N=N_dev, **idx)                            # -> GemNet.forward

`DeviceGraphBuilder` keeps the molecule layout (offsets, workspace) for repeated calls on the same system.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_device, stream

KEYS_T = ["batch_seg", "id_undir", "id_swap", "id_c", "id_a", "id3_expand_ba", "id3_reduce_ca", "Kidx3"]
KEYS_Q = ["id4_int_b", "id4_int_a", "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd", "Kidx4",
          "id4_reduce_intm_ca", "id4_expand_intm_db", "id4_reduce_intm_ab", "id4_expand_intm_ab"]


class DeviceGraphBuilder:
    def __init__(self, N, cutoff, int_cutoff, triplets_only=False, device="cuda"):
        N = np.asarray(N, dtype=np.int64).reshape(-1)
        self.B, self.A = int(len(N)), int(N.sum())
        self.nmax = int(N.max()) if len(N) else 0
        self.sum_n2 = int((N * N).sum())
        if self.sum_n2 >= 2 ** 31:
            raise ValueError("sum of squared molecule sizes exceeds int32")
        self.cutoff, self.int_cutoff, self.triplets_only = float(cutoff), float(int_cutoff), bool(triplets_only)
        self.device = torch.device(device)
        self.mol_off = torch.tensor(np.concatenate([[0], np.cumsum(N)]), dtype=torch.int32, device=self.device)
        self.sq_off = torch.tensor(np.concatenate([[0], np.cumsum(N * N)]), dtype=torch.int32, device=self.device)
        self.lib = _lib.load()
        nbytes = int(self.lib.gn_index_gpu_ws_bytes(self.A, self.sum_n2, int(self.triplets_only)))
        self.ws = torch.empty(max(nbytes, 64), dtype=torch.uint8, device=self.device)
        self.cap = max(self.sum_n2 - self.A, 0)  # upper bound of E and of Eint

    def __call__(self, R, dtype=torch.int64):
        """R (A,3) device float32 | float64 -> {key: tensor(dtype)}; dtype int64 (reference) or int32 (no copy)."""
        require_device(R)
        if R.dtype not in (torch.float32, torch.float64):
            raise TypeError("positions must be float32 or float64")
        R = R.contiguous()
        assert R.shape == (self.A, 3)
        dev, i32 = R.device, torch.int32
        quad = not self.triplets_only
        new = lambda n: torch.empty(int(n), dtype=i32, device=dev)
        batch_seg = new(self.A)
        e_arr = {k: new(self.cap) for k in ("id_a", "id_c", "id_undir", "id_swap")}
        i_arr = {k: new(self.cap if quad else 0) for k in ("id4_int_a", "id4_int_b")}
        sizes = (ctypes.c_int64 * 6)()
        check(_lib.load().gn_index_gpu_stage1(
            ptr(R), int(R.dtype == torch.float64), ptr(self.mol_off), ptr(self.sq_off), self.B, self.A, self.nmax,
            self.sum_n2, self.cutoff, self.int_cutoff, int(self.triplets_only), ptr(self.ws), ptr(batch_seg),
            ptr(e_arr["id_a"]), ptr(e_arr["id_c"]), ptr(e_arr["id_undir"]), ptr(e_arr["id_swap"]),
            ptr(i_arr["id4_int_a"]), ptr(i_arr["id4_int_b"]), sizes, stream()), "gn_index_gpu_stage1")
        E, T, Eint, Ica, Idb, Q = (int(v) for v in sizes)
        out = {"batch_seg": batch_seg}
        for k, v in e_arr.items():
            out[k] = v[:E]
        t_arr = {k: new(T) for k in ("id3_reduce_ca", "id3_expand_ba", "Kidx3")}
        q_arr = {}
        if quad:
            for k, v in i_arr.items():
                out[k] = v[:Eint]
            q_arr = {k: new(Q) for k in ("id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd", "Kidx4")}
            q_arr.update(id4_reduce_intm_ca=new(Ica), id4_reduce_intm_ab=new(Ica),
                         id4_expand_intm_db=new(Idb), id4_expand_intm_ab=new(Idb))
        g = lambda d, k: ptr(d[k]) if k in d else None
        check(_lib.load().gn_index_gpu_stage2(
            ptr(self.mol_off), ptr(self.sq_off), self.B, self.A, self.sum_n2, int(self.triplets_only), ptr(self.ws),
            ptr(out["id_a"]), ptr(out["id_c"]), g(out, "id4_int_a"), g(out, "id4_int_b"), E, Eint,
            ptr(t_arr["id3_reduce_ca"]), ptr(t_arr["id3_expand_ba"]), ptr(t_arr["Kidx3"]),
            g(q_arr, "id4_reduce_ca"), g(q_arr, "id4_expand_db"), g(q_arr, "id4_reduce_cab"), g(q_arr, "id4_expand_abd"),
            g(q_arr, "Kidx4"), g(q_arr, "id4_reduce_intm_ca"), g(q_arr, "id4_expand_intm_db"),
            g(q_arr, "id4_reduce_intm_ab"), g(q_arr, "id4_expand_intm_ab"), stream()), "gn_index_gpu_stage2")
        out.update(t_arr)
        out.update(q_arr)
        keys = KEYS_T + ([] if self.triplets_only else KEYS_Q)
        return {k: (out[k] if dtype == torch.int32 else out[k].to(dtype)) for k in keys}


def build_indices_device(R, N, cutoff, int_cutoff, triplets_only=False, dtype=torch.int64):
    return DeviceGraphBuilder(N, cutoff, int_cutoff, triplets_only, device=R.device)(R, dtype=dtype)


def ensure_indices(inputs, cutoff, int_cutoff, triplets_only):
    """`inputs` as `GemNet.forward` takes them; when the index arrays are missing (a `DataContainer(indices="device")` batch:
    Z, R, N only) they are built here from the device-resident positions — the same arrays, in the same canonical order, as the
    host builder's (tests/test_gpu_index.py).  One small read-back (the molecule sizes) + the builder's own size read-backs."""
    if "id_c" in inputs:
        return inputs
    R = inputs["R"]
    require_device(R)      # (no CPU fallback: on the host the arrays come from training.data_container.build_indices)
    N_host = inputs["N"].detach().cpu().numpy()
    idx = build_indices_device(R.detach(), N_host, cutoff, int_cutoff, triplets_only, dtype=torch.int32)
    return dict(inputs, **idx)
