"""Synthetic COLL-shaped molecules (the COLL npz files are absent from the reference mount).

Generator specified in SURVEY.md §8(d): n atoms uniform in a cube of edge L, rejection if
any pair is closer than 0.9 A; Z in {1, 6, 8}; R float32; targets E ~ N(0,1), F ~ N(0,1).
n=32 -> L=7.0 A, n=64 -> L=8.8 A, n=12 -> L=5.0 A.  Molecule i of a batch for config k uses
seed 1000*k + i.  The npz schema (keys N, Z, R, E, F) is the one the reference's
DataContainer reads (gemnet/training/data_container.py:61).
"""
import numpy as np

BOX = {12: 5.0, 32: 7.0, 64: 8.8}


def make_molecule(n_atoms: int, seed: int, box: float = None, min_dist: float = 0.9):
    rs = np.random.RandomState(seed)
    L = box if box is not None else BOX.get(n_atoms, 7.0 * (n_atoms / 32.0) ** (1 / 3))
    R = np.zeros((0, 3))
    while len(R) < n_atoms:
        p = rs.uniform(0, L, size=(1, 3))
        if len(R) == 0 or np.min(np.linalg.norm(R - p, axis=1)) >= min_dist:
            R = np.concatenate([R, p])
    Z = rs.choice(np.array([1, 6, 8]), size=n_atoms)
    E = rs.standard_normal()
    F = rs.standard_normal((n_atoms, 3))
    return dict(N=n_atoms, Z=Z.astype(np.int32), R=R.astype(np.float32),
                E=np.float32(E), F=F.astype(np.float32))


def make_dataset(n_mol: int, n_atoms: int, config: int = 2, first: int = 0, ids=None):
    """-> dict with the COLL npz keys: N (M,), Z (sumN,), R (sumN,3), E (M,), F (sumN,3).
    `ids`: explicit molecule numbers (a rank's shard of a global batch) instead of first .. first + n_mol - 1."""
    if ids is None:
        ids = range(first, first + n_mol)
    mols = [make_molecule(n_atoms, 1000 * config + int(i)) for i in ids]
    return dict(
        N=np.array([m["N"] for m in mols], dtype=np.int32),
        Z=np.concatenate([m["Z"] for m in mols]),
        R=np.concatenate([m["R"] for m in mols]),
        E=np.array([m["E"] for m in mols], dtype=np.float32),
        F=np.concatenate([m["F"] for m in mols]),
    )


def triplet_count(R, cutoff=5.0):
    """Number of triplets (c->a, b->a, b != c) of one molecule: the per-molecule cost GemNet-T's kernels scale with
    (used to balance molecule shards across ranks, training/ddp.py::partition_molecules)."""
    d = np.linalg.norm(R[:, None, :].astype(np.float64) - R[None, :, :].astype(np.float64), axis=-1)
    deg = ((d < cutoff) & (d > 0)).sum(axis=1)
    return int((deg * (deg - 1)).sum())
