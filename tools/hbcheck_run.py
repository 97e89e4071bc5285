"""Happens-before check (gemnet_pytorch_amd/hbcheck.py) of the captured steps, including the three configurations whose
hipGraph replays did not match the eager run in round 3:

   PYTHONPATH=.:tests python tools/hbcheck_run.py [case ...]        # default: all

cases:  T32 T64 Q64 train            the shipped configurations
        T64-rbfout-side              finding 2: output-block radial projection produced on the side stream of the forked head
        Q64-side                     finding 1: quadruplet model with its output blocks on the side stream
        train-overlap                finding 3: force training with the output blocks on the side stream
Every case also replays the captured graph a few times and reports whether the replays reproduce the eager result bit for
bit — the symptom next to the diagnosis."""
import copy
import sys

import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd import hbcheck
from gemnet_pytorch_amd.model import gemnet as G
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.ddp import TrainStep

DEV = "cuda"
FULL = dict(num_spherical=7, num_radial=6, num_blocks=4, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
            emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
            emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2)


def batch(n_mol, n_atoms, triplets_only, keep_targets=False):
    ds = make_dataset(n_mol, n_atoms, config=2)
    dc = DataContainer.from_arrays(dict(ds), 5.0, 10.0, triplets_only=triplets_only)
    b = dc[list(range(n_mol))]
    inputs = {k: v.to(DEV) for k, v in b.items() if k not in ("E", "F")}
    if keep_targets:
        g = torch.Generator().manual_seed(4)
        return inputs, {"E": torch.randn(n_mol, 1, generator=g).to(DEV), "F": torch.randn(n_mol * n_atoms, 3, generator=g).to(DEV)}
    return inputs


def warm(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


def force_case(name, kind, n_mol, n_atoms, flags=()):
    for f in ("_Q_OVERLAP", "_RBF_OUT_SIDE"):
        setattr(G, f, f in flags)
    cfg = dict(FULL, triplets_only=kind == "T")
    torch.manual_seed(11)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    inputs = batch(n_mol, n_atoms, cfg["triplets_only"])
    E0, F0 = model(inputs)
    s = 1.0 / float(F0.abs().mean())
    with torch.no_grad():
        for ob in model.out_blocks:
            ob.out_energy.weight.mul_(s)
    model._wcache.clear()
    E0, F0 = (t.detach().clone() for t in model(inputs))
    warm(lambda: model(inputs))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        with hbcheck.record() as rec:
            Eg, Fg = model(inputs)
    races = rec.races()
    print(f"=== {name}: {len(races)} unordered conflicting pairs")
    print(rec.format(races))
    bad = 0
    for _ in range(10):
        graph.replay()
        torch.cuda.synchronize()
        bad += int(not (torch.equal(Fg, F0) and torch.equal(Eg, E0)))
    print(f"=== {name}: {bad} of 10 replays differ from the eager run (max |dF| of the last {float((Fg - F0).abs().max()):.3e})",
          flush=True)
    return len(races), bad


def train_case(name, overlap):
    G._TRAIN_OVERLAP = overlap
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(9)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    inputs, targets = batch(8, 32, True, keep_targets=True)
    ts = TrainStep(copy.deepcopy(model), fused_optimizer=True)
    ts(inputs, targets, step_optimizer=False)
    torch.cuda.synchronize()
    ref = ts.buf.flat.clone()
    ts.capture(inputs, targets, check=True)
    races = ts.hb.races()
    print(f"=== {name}: {len(races)} unordered conflicting pairs")
    print(ts.hb.format(races))
    devs = []
    for _ in range(6):
        ts(inputs, targets, step_optimizer=False)
        torch.cuda.synchronize()
        devs.append(float((ts.buf.flat - ref).norm() / ref.norm()))
    print(f"=== {name}: flat gradient of 6 replays vs the eager step: {['%.1e' % d for d in devs]}", flush=True)
    return len(races), sum(d != 0 for d in devs)


CASES = {
    "T32": lambda: force_case("T32", "T", 32, 32),
    "T64": lambda: force_case("T64", "T", 8, 64),
    "Q64": lambda: force_case("Q64", "Q", 8, 64),
    "train": lambda: train_case("train", False),
    "T64-rbfout-side": lambda: force_case("T64-rbfout-side", "T", 8, 64, ("_RBF_OUT_SIDE",)),
    "Q64-side": lambda: force_case("Q64-side", "Q", 8, 64, ("_Q_OVERLAP",)),
    "train-overlap": lambda: train_case("train-overlap", True),
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    res = {}
    for n in names:
        try:
            res[n] = CASES[n]()
        except Exception as e:   # keep going: one broken case must not hide the others' reports
            import traceback
            traceback.print_exc()
            res[n] = ("error", repr(e))
    print("summary (unordered pairs, replays differing):", res)
