"""Shared by the BASELINE-size golden tests (tests/golden/fullsize.npz + fullsize_index.json, written by
tests/golden/make_golden.py::golden_fullsize from the REFERENCE in float64): the four workloads and the index digest."""
import ast
import hashlib
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, SCALE_FILE
from oracle import gemnet_oracle as GO
from oracle import index_oracle as IO
from gemnet_pytorch_amd.synthetic import make_dataset, make_molecule

HEAD_KEYS = ("out_energy.weight", "out_forces.weight")


def load_fullsize():
    return dict(np.load(os.path.join(GOLDEN, "fullsize.npz")))


def load_digests():
    with open(os.path.join(GOLDEN, "fullsize_index.json")) as f:
        return json.load(f)


def dataset(tag):
    """The generated inputs of a fixture, from the seeded generator (the npz holds them too: asserted equal by the tests)."""
    if tag in ("t64s", "q64s"):
        m = make_molecule(64, 4000)
        return dict(N=np.array([64], np.int32), Z=m["Z"], R=m["R"], E=np.zeros(1, np.float32), F=np.zeros_like(m["R"]))
    if tag == "tB32" or tag == "idxB32.Q":
        return make_dataset(32, 32, config=2)
    if tag == "qB4":
        return make_dataset(4, 32, config=2)
    if tag in ("idx32.T", "idx32.Q"):
        return make_dataset(1, 32, config=2)
    raise KeyError(tag)


def triplets_only(tag):
    return tag in ("t64s", "tB32", "idx32.T")


def params_of(g, tag, dtype=torch.float32):
    cfg = ast.literal_eval(str(g[f"{tag}.cfg"]))
    params = GO.make_params(cfg, int(g[f"{tag}.seed"]), GO.load_scale_factors(SCALE_FILE), dtype=dtype)
    sc = float(g[f"{tag}.out_scale"])
    return cfg, {k: (v * sc if k.endswith(HEAD_KEYS) else v) for k, v in params.items()}


def digest(idx, to, canonicalize=False):
    """sizes + SHA-256 of the int32 little-endian bytes of the index arrays (make_golden.py::_index_digest hashes the
    CANONICAL form of the reference's arrays).  The product builders and the oracle emit the canonical order themselves, so their
    raw output is hashed as it is (canonicalize=False): a builder that returned another within-segment order would fail here."""
    can = IO.canonicalize({k: np.asarray(v) for k, v in idx.items()}, to) if canonicalize else {k: np.asarray(v) for k, v in idx.items()}
    return {k: dict(n=int(v.shape[0]), sha256=hashlib.sha256(np.ascontiguousarray(v.astype("<i4")).tobytes()).hexdigest())
            for k, v in sorted(can.items())}
