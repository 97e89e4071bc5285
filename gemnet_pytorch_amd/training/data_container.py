"""`DataContainer` — counterpart of the reference's gemnet/training/data_container.py:7-565 (H1 of
SURVEY.md §8): same constructor arguments, same `__getitem__(idx | list | slice) -> dict` keys and
dtypes (int64 indices, float32 R/E/F), same attribute contract used by ase_calculator.Molecule
(`index_keys`, `keys`, `get_dtypes`, `N_cumsum`, `convert_to_tensor`).

The index construction itself (edges, id_swap, triplets, quadruplets, Kidx) runs in the native
host builder csrc/index_build.cpp through the C ABI of include/gemnet_index.h; within-segment order
is canonical (ascending expand edge) where the reference's depends on numpy's unstable argsort.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
INDEX_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libgemnet_index.so")
_ilib = None


def _load_index_lib():
    global _ilib
    if _ilib is None:
        if not os.path.exists(INDEX_LIB_PATH):
            raise RuntimeError(f"{INDEX_LIB_PATH} not found: run __graft_entry__.build() first")
        lib = ctypes.CDLL(INDEX_LIB_PATH)
        lib.gn_index_build.restype = ctypes.c_void_p
        lib.gn_index_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_int]
        lib.gn_index_free.argtypes = [ctypes.c_void_p]
        lib.gn_index_free.restype = None
        lib.gn_index_size.restype = ctypes.c_int64
        lib.gn_index_size.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.gn_index_copy.restype = ctypes.c_int
        lib.gn_index_copy.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
        _ilib = lib
    return _ilib


INDEX_KEYS_T = ["batch_seg", "id_undir", "id_swap", "id_c", "id_a", "id3_expand_ba", "id3_reduce_ca", "Kidx3"]
INDEX_KEYS_Q = ["id4_int_b", "id4_int_a", "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab", "id4_expand_abd",
                "Kidx4", "id4_reduce_intm_ca", "id4_expand_intm_db", "id4_reduce_intm_ab", "id4_expand_intm_ab"]


def build_indices(R, N, cutoff, int_cutoff, triplets_only):
    """R (A,3) float32|float64, N (B,) -> {key: int64 ndarray}; distances are evaluated in R's dtype
    (float32 from a dataset, float64 from ASE positions: ase_calculator.py:155)."""
    R = np.ascontiguousarray(R)
    if R.dtype not in (np.float32, np.float64):
        R = R.astype(np.float32)
    N32 = np.ascontiguousarray(np.asarray(N, dtype=np.int32))
    assert R.shape == (int(N32.sum()), 3)
    lib = _load_index_lib()
    h = lib.gn_index_build(R.ctypes.data_as(ctypes.c_void_p), int(R.dtype == np.float64),
                           N32.ctypes.data_as(ctypes.c_void_p), len(N32), float(cutoff), float(int_cutoff),
                           int(bool(triplets_only)))
    if not h:
        raise RuntimeError("gn_index_build failed")
    try:
        out = {}
        for key in INDEX_KEYS_T + ([] if triplets_only else INDEX_KEYS_Q):
            n = lib.gn_index_size(h, key.encode())
            arr = np.empty(n, dtype=np.int64)
            if lib.gn_index_copy(h, key.encode(), arr.ctypes.data_as(ctypes.c_void_p)) != 0:
                raise RuntimeError(f"gn_index_copy({key}) failed")
            out[key] = arr
        return out
    finally:
        lib.gn_index_free(h)


class DataContainer:
    def __init__(self, path, cutoff, int_cutoff, triplets_only=False, transforms=None, addID=False, indices="host"):
        """`indices` (not in the reference): "host" — every batch carries its index arrays, built by the native host builder as
        data_container.py:244-489 does (the default: a drop-in); "device" — a batch carries Z, R, N and the targets only and the
        model builds the arrays on the GPU when it is called (`index_device.ensure_indices`, csrc/index_gpu.hip: 0.5 ms for a
        32 x 32-atom GemNet-Q batch against 0.9 s on a host core) — for loaders that would otherwise bound a GemNet-Q training
        step of 38 ms."""
        if indices not in ("host", "device"):
            raise ValueError("indices: 'host' or 'device'")
        self.indices = indices
        self.index_keys = list(INDEX_KEYS_T) + ([] if triplets_only else list(INDEX_KEYS_Q))
        self.triplets_only = triplets_only
        self.cutoff = cutoff
        self.int_cutoff = int_cutoff
        self.addID = addID
        self.keys = ["N", "Z", "R", "F", "E"] + (["id"] if addID else [])
        self._load_npz(path, self.keys)
        self.transforms = [] if transforms is None else list(transforms)
        for transform in self.transforms:
            transform(self)
        for k in ("R", "N", "Z", "E", "F"):
            assert getattr(self, k, None) is not None, k
        assert len(self.E) > 0 and len(self.F) > 0
        self.E = self.E[:, None]
        self.N_cumsum = np.concatenate([[0], np.cumsum(self.N)])
        self.dtypes, dtypes2 = self.get_dtypes()
        self.dtypes.update(dtypes2)
        self.targets = ["E", "F"]

    @classmethod
    def from_arrays(cls, data, cutoff, int_cutoff, triplets_only=False, **kw):
        """Build from an in-memory dict with the COLL npz keys (N, Z, R, E, F)."""
        self = cls.__new__(cls)
        self._mem = data
        cls.__init__(self, None, cutoff, int_cutoff, triplets_only=triplets_only, **kw)
        return self

    def _load_npz(self, path, keys):
        src = getattr(self, "_mem", None)
        if src is None:
            src = np.load(path, allow_pickle=True)
        for key in keys:
            if key not in src:
                if key != "F":
                    raise UserWarning(f"Can not find key {key} in the dataset.")
            else:
                setattr(self, key, np.asarray(src[key]))

    def __len__(self):
        return len(self.N)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            idx = [int(idx)]
        if isinstance(idx, tuple):
            idx = list(idx)
        if isinstance(idx, slice):
            idx = np.arange(idx.start or 0, min(idx.stop, len(self)), idx.step)
        idx = [int(i) for i in idx]
        data = {}
        if self.addID:
            data["id"] = self.id[idx]
        data["E"] = self.E[idx]
        data["N"] = self.N[idx]
        rows = np.concatenate([np.arange(self.N_cumsum[i], self.N_cumsum[i + 1]) for i in idx]) \
            if idx else np.zeros(0, dtype=np.int64)
        data["Z"] = self.Z[rows].astype(np.int32)
        R = self.R[rows]
        data["R"] = R.astype(np.float32)
        data["F"] = self.F[rows].astype(np.float32)
        # cutoff membership is decided in the dtype the positions are stored in (float32 for the
        # npz datasets, float64 when an ASE caller assigns float64 positions to .R)
        if getattr(self, "indices", "host") == "host":      # (subclasses with their own __init__: ase_calculator.Molecule)
            data.update(build_indices(R, data["N"], self.cutoff, self.int_cutoff, self.triplets_only))
            return self.convert_to_tensor(data)
        out = self.convert_to_tensor(data)
        # a batch without index arrays names the graph it stands for: (cutoff, int_cutoff) — GemNet.with_indices builds the
        # neighbour lists from THESE (as the host mode and the reference do from the DataContainer's), not from the model's
        out["cutoffs"] = torch.tensor([float(self.cutoff), float(self.int_cutoff)], dtype=torch.float64)
        return out

    def convert_to_tensor(self, data):
        for key in data:
            data[key] = torch.tensor(np.asarray(data[key]), dtype=self.dtypes[key])
        return data

    def get_dtypes(self):
        dtypes_input = {}
        if self.addID:
            dtypes_input["id"] = torch.int64
        dtypes_input["Z"] = torch.int64
        dtypes_input["N"] = torch.int64
        dtypes_input["R"] = torch.float32
        for key in self.index_keys:
            dtypes_input[key] = torch.int64
        return dtypes_input, {"E": torch.float32, "F": torch.float32}
