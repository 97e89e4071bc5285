"""Happens-before check (gemnet_pytorch_amd/hbcheck.py) of the captured steps, including the three configurations whose
hipGraph replays did not match the eager run in round 3 (docs/HISTORY.md section 10):

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


def diff_replays(name, rec, replay, n=6):
    """Where do two replays part?  (recorder with keep=True: every intermediate of the captured step is still there)"""
    replay()
    torch.cuda.synchronize()
    snap = rec.snapshot()
    for k in range(n):
        replay()
        torch.cuda.synchronize()
        cnt, text = rec.replay_diff(snap)
        print(f"=== {name}: replay {k + 1} vs replay 0: {cnt} operations wrote differing tensors (first ones, in issue order):")
        if cnt:
            print(text, flush=True)


def forensics(name, rec, replay, n=4):
    """The aggregation adjoint g_m[e] = s g_out[id_a[e]] (.) (W rbf[e]) has no accumulator and single-writer inputs:
    recompute it from the operands as they are AFTER a replay and look at the elements the replay got wrong."""
    by_op = {}
    for idx, kind, what, t in rec.touched:
        by_op.setdefault(idx, {})[what.split(" ")[0]] = t
    ops = [o for o in rec.ops if o.name == "gn_rbf_aggregate_bwd_f32"]
    for k in range(n):
        replay()
        torch.cuda.synchronize()
        for o in ops:
            t = by_op[o.idx]
            if "g_m" not in t:
                continue
            g_out, rbf, W, id_a, g_m = t["g_out"], t["rbf"], t["W"], t["id_a"].long(), t["g_m"]
            ref = g_out[id_a].double() * (rbf.double() @ W.double().T)
            ok = ref.abs() > 1e-6 * ref.abs().max()
            scale = float((g_m.double()[ok] / ref[ok]).median())
            err = (g_m.double() - scale * ref).abs()
            bad = err > 1e-5 * float(ref.abs().max()) * abs(scale)
            if not bool(bad.any()):
                continue
            rows = bad.any(dim=1).nonzero().flatten()
            atoms = id_a[rows]
            print(f"=== {name}: replay {k}: {o!r}: {int(bad.sum())} wrong elements in {rows.numel()} rows; scale {scale:.4g}")
            print(f"      rows {rows.tolist()[:24]}")
            print(f"      their target atoms {atoms.tolist()[:24]}")
            for a in atoms.unique().tolist()[:3]:
                all_rows = (id_a == a).nonzero().flatten()
                print(f"      atom {a}: in-degree {all_rows.numel()}, wrong rows among them {int(bad[all_rows].any(dim=1).sum())}")
            r = int(rows[0])
            cols = bad[r].nonzero().flatten()
            c0, c1 = int(cols.min()), int(cols.max()) + 1
            a = int(id_a[r])
            wr = (rbf[r].double() @ W.double().T)[c0:c1]
            print(f"      row {r} (atom {a}), wrong columns {c0}..{c1 - 1}:")
            print(f"        got      {[f'{v:.5e}' for v in g_m[r, c0:c1].tolist()]}")
            print(f"        expected {[f'{v:.5e}' for v in (scale * ref[r, c0:c1]).tolist()]}")
            print(f"        implied g_out[a] {[f'{v:.5e}' for v in (g_m[r, c0:c1].double() / (scale * wr)).tolist()]}")
            print(f"        actual  g_out[a] {[f'{v:.5e}' for v in g_out[a, c0:c1].tolist()]}")
            # is the implied g_out row some OTHER row of g_out (or of another block's g_out)?
            imp = (g_m[r, c0:c1].double() / (scale * wr))
            for o2 in ops:
                g2 = by_op[o2.idx].get("g_out")
                if g2 is None:
                    continue
                d = (g2[:, c0:c1].double() - imp[None, :]).abs().max(dim=1).values
                j = int(d.argmin())
                print(f"        nearest row of the g_out of op #{o2.idx}: row {j}, max |d| {float(d[j]):.3e}")
            sys.stdout.flush()
            break


def force_case(name, kind, n_mol, n_atoms, flags=(), keep=False):
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
        with hbcheck.record(keep=keep) as rec:
            Eg, Fg = model(inputs)
    races = rec.races()
    print(f"=== {name}: {len(races)} unordered conflicting pairs")
    print(rec.format(races))
    import os
    if os.environ.get("HB_DUMP"):
        rec.dump(os.path.join(os.environ["HB_DUMP"], name + ".json"))
    bad = 0
    for _ in range(10):
        graph.replay()
        torch.cuda.synchronize()
        bad += int(not (torch.equal(Fg, F0) and torch.equal(Eg, E0)))
    print(f"=== {name}: {bad} of 10 replays differ from the eager run (max |dF| of the last {float((Fg - F0).abs().max()):.3e})",
          flush=True)
    if keep:
        diff_replays(name, rec, graph.replay, n=2)
        forensics(name, rec, graph.replay)
    return len(races), bad


def train_case(name, overlap, keep=False):
    G._TRAIN_OVERLAP = overlap
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(9)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    inputs, targets = batch(8, 32, True, keep_targets=True)
    ts = TrainStep(copy.deepcopy(model), fused_optimizer=True)
    ts(inputs, targets, step_optimizer=False)
    torch.cuda.synchronize()
    ref = ts.buf.flat.clone()
    ts.capture(inputs, targets, check="keep" if keep else True)
    races = ts.hb.races()
    print(f"=== {name}: {len(races)} unordered conflicting pairs")
    print(ts.hb.format(races))
    devs = []
    for _ in range(6):
        ts(inputs, targets, step_optimizer=False)
        torch.cuda.synchronize()
        devs.append(float((ts.buf.flat - ref).norm() / ref.norm()))
    print(f"=== {name}: flat gradient of 6 replays vs the eager step: {['%.1e' % d for d in devs]}", flush=True)
    if keep:
        diff_replays(name, ts.hb, ts._graph.replay)
    return len(races), sum(d != 0 for d in devs)


CASES = {
    "T32": lambda: force_case("T32", "T", 32, 32),
    "T64": lambda: force_case("T64", "T", 8, 64),
    "Q64": lambda: force_case("Q64", "Q", 8, 64),
    "train": lambda: train_case("train", False),
    "T64-rbfout-side": lambda: force_case("T64-rbfout-side", "T", 8, 64, ("_RBF_OUT_SIDE",)),
    "Q64-side": lambda: force_case("Q64-side", "Q", 8, 64, ("_Q_OVERLAP",)),
    "train-overlap": lambda: train_case("train-overlap", True),
    # the same two with every intermediate kept: which operation's output differs first from replay to replay?
    "T64-rbfout-side-diff": lambda: force_case("T64-rbfout-side-diff", "T", 8, 64, ("_RBF_OUT_SIDE",), keep=True),
    "train-overlap-diff": lambda: train_case("train-overlap-diff", True, keep=True),
}

if __name__ == "__main__":
    names = sys.argv[1:] or [n for n in CASES if not n.endswith("-diff")]
    res = {}
    for n in names:
        try:
            res[n] = CASES[n]()
        except Exception as e:   # keep going: one broken case must not hide the others' reports
            import traceback
            traceback.print_exc()
            res[n] = ("error", repr(e))
    print("summary (unordered pairs, replays differing):", res)
