"""-m gpu: batches of changing array sizes replayed from ONE captured hipGraph (gemnet_pytorch_amd/padded.py): the real
molecules of a batch padded to the capacities get the energies and forces of the plain eager run on the unpadded batch,
for several batches with different edge / triplet counts in turn, twice around (the second round replays only)."""
import numpy as np
import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from test_gpu_fullsize import FULL

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("direct", [False, True], ids=["autograd-forces", "direct-forces"])
def test_padded_graph_replay_equals_eager_on_changing_batches(direct):
    cfg = dict(FULL, triplets_only=True, direct_forces=direct)
    torch.manual_seed(5)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    n_mol, n_atoms = 8, 32
    batches = []
    for b in range(4):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
        R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
        Z = torch.tensor(ds["Z"], device=DEV).long()
        N = torch.tensor(ds["N"], device=DEV).long()
        idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
        batches.append((Z, R, N, idx))
    sizes = [(int(i["id_c"].shape[0]), int(i["id3_reduce_ca"].shape[0])) for _, _, _, i in batches]
    assert len(set(sizes)) > 1, sizes                     # the point of the exercise: the shapes differ
    Z, _, N, _ = batches[0]
    assert all(torch.equal(N, b[2]) for b in batches)     # one layout of molecule sizes; atoms and geometry change
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    runner = PaddedGraphRunner(model, Z, N, e_cap, t_cap)
    runner.check = not direct      # the capture under the happens-before checker (the plan is built inside it, on two streams)
    ref = []
    for Zb, R, Nb, idx in batches:
        E, F = model(dict(Z=Zb, R=R.clone(), N=Nb, **idx))
        ref.append((E.detach().clone(), F.detach().clone()))
    worst = 0.0
    for rnd in range(2):
        for (Zb, R, Nb, idx), (E0, F0) in zip(batches, ref):
            E, F = runner(R, idx, Z=Zb)
            torch.cuda.synchronize()
            assert E.shape == E0.shape and F.shape == F0.shape
            worst = max(worst, float((F - F0).abs().max()), float((E - E0).abs().max()))
            scale = float(F0.abs().max())
            assert float((F - F0).abs().max()) <= 1e-5 * scale and float((E - E0).abs().max()) <= 1e-5 * float(E0.abs().max())
    if not direct:
        races, summary = runner.hb.races(), runner.hb.summary()
        print(runner.hb.format(races))
        assert not races
        assert summary["unrecorded_nodes"] == 0 and summary["unresolved_pointers"] == 0
        assert summary["streams"] >= 3     # main, output blocks, the adjoint-only index structures
    # index build + replay in one call, the build on its own stream
    builders = [DeviceGraphBuilder(N.cpu().numpy(), 5.0, 10.0, True, device=DEV) for _ in batches]
    for rnd in range(2):
        for (Zb, R, Nb, idx), (E0, F0), bld in zip(batches, ref, builders):
            E, F = runner.build_and_run(bld, R, Z=Zb, positions_ready=bool(rnd))
            torch.cuda.synchronize()
            assert torch.equal(E, E0) and torch.equal(F, F0)
    print(f"padded replay vs eager over {len(batches)} batches {sizes} at capacities ({runner.e_cap}, {runner.t_cap}), "
          f"{runner.G} dummy groups: max abs deviation {worst:.3e}")
    with pytest.raises(ValueError):
        small = PaddedGraphRunner(model, Z, N, sizes[0][0] - 8, sizes[0][1])
        small(batches[0][1], batches[0][3], Z=batches[0][0])


def test_padded_training_step_replays_one_graph_for_changing_batches():
    """PaddedTrainStep: the training step (forward + force + loss.backward() through the force) of batches with different
    edge / triplet counts from ONE captured hipGraph — loss and parameter gradients of every batch equal those of the
    eager TrainStep on the unpadded batch (to fp32 rounding: the weight-gradient products split their longer contraction
    differently), and three optimizer steps leave both models with the same parameters."""
    import copy
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(7)
    model_a = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    model_b = copy.deepcopy(model_a)
    n_mol, n_atoms = 8, 32
    g = torch.Generator().manual_seed(3)
    batches = []
    for b in range(3):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
        R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
        Z = torch.tensor(ds["Z"], device=DEV).long()
        N = torch.tensor(ds["N"], device=DEV).long()
        idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
        Et = torch.randn(n_mol, 1, generator=g).to(DEV)
        Ft = torch.randn(n_mol * n_atoms, 3, generator=g).to(DEV)
        batches.append((Z, R, N, idx, Et, Ft))
    sizes = [(int(b[3]["id_c"].shape[0]), int(b[3]["id3_reduce_ca"].shape[0])) for b in batches]
    assert len(set(sizes)) > 1
    from gemnet_pytorch_amd.padded import PaddedGraphRunner
    pts = PaddedTrainStep(model_a, batches[0][0], batches[0][2], *PaddedGraphRunner.suggest_capacities(sizes), fused_optimizer=True)
    ets = TrainStep(model_b, fused_optimizer=True)
    worst = 0.0
    for rnd, step_opt in ((0, False), (1, True)):
        for Z, R, N, idx, Et, Ft in batches:
            la = pts.step(R, idx, Et, Ft, Z=Z, step_optimizer=step_opt)
            ga = pts.buf.flat.clone()
            lb = ets(dict(Z=Z, R=R.clone(), N=N, **idx), {"E": Et, "F": Ft}, step_optimizer=step_opt)
            gb = ets.buf.flat.clone()
            torch.cuda.synchronize()
            if step_opt:
                assert abs(float(la) - float(lb)) <= 2e-3 * abs(float(lb)), (float(la), float(lb))
            if not step_opt:      # same parameters on both sides: compare the step itself
                assert abs(float(la) - float(lb)) <= 2e-5 * abs(float(lb))
                rel = float((ga - gb).norm() / gb.norm())
                worst = max(worst, rel)
                assert rel <= 1e-5, rel
    pa = torch.cat([p.detach().reshape(-1) for p in model_a.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in model_b.parameters()])
    drift = float((pa - pb).abs().max())
    print(f"padded training step vs eager over batches {sizes}: largest gradient deviation {worst:.2e} (relative, flat buffer), "
          f"parameters after three optimizer steps differ by at most {drift:.2e}")
    # AdamW moves every entry by ~lr per step whatever the gradient's size: an entry whose gradient is fp32 noise may go the
    # other way on the two sides (2 x 3 steps x lr 1e-3 at most); the trajectories stay together (losses above)
    assert drift <= 6.5e-3


def _batch(n_mol, n_atoms, first, g=None):
    ds = make_dataset(n_mol, n_atoms, config=2, first=first)
    R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
    out = dict(Z=torch.tensor(ds["Z"], device=DEV).long(), R=R, N=torch.tensor(ds["N"], device=DEV).long(),
               idx=DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R))
    if g is not None:
        out["Et"] = torch.randn(n_mol, 1, generator=g).to(DEV)
        out["Ft"] = torch.randn(n_mol * n_atoms, 3, generator=g).to(DEV)
    return out


def test_changing_molecule_sizes_through_one_graph():
    """`a_cap`: batches of 8 molecules with 32, 24 and 28 atoms each (then the first again) through ONE captured graph —
    forward+force bit-identical to the eager run on the unpadded batch; the training step to fp32 rounding."""
    import copy
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(9)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    g = torch.Generator().manual_seed(4)
    batches = [_batch(8, n, 100 * (i + 1), g) for i, n in enumerate((32, 24, 28))]
    sizes = [(int(b["idx"]["id_c"].shape[0]), int(b["idx"]["id3_reduce_ca"].shape[0])) for b in batches]
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    big = batches[0]
    runner = PaddedGraphRunner(model, big["Z"], big["N"], e_cap, t_cap, a_cap=8 * 32, max_in_degree=31,
                               n_groups=max(1, e_cap // (2 * 31)))
    for b in batches + [batches[0]]:
        E0, F0 = model(dict(Z=b["Z"], R=b["R"].clone(), N=b["N"], **b["idx"]))
        E, F = runner(b["R"], b["idx"], Z=b["Z"], N=b["N"])
        torch.cuda.synchronize()
        assert torch.equal(E, E0.detach()) and torch.equal(F, F0.detach()), (float((F - F0).abs().max()),)
    # training
    model_a = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    model_b = copy.deepcopy(model_a)
    pts = PaddedTrainStep(model_a, big["Z"], big["N"], e_cap, t_cap, a_cap=8 * 32, max_in_degree=31,
                          n_groups=max(1, e_cap // (2 * 31)), fused_optimizer=True)
    ets = TrainStep(model_b, fused_optimizer=True)
    worst = 0.0
    for b in batches + [batches[0]]:
        la = pts.step(b["R"], b["idx"], b["Et"], b["Ft"], Z=b["Z"], N=b["N"], step_optimizer=False)
        ga = pts.buf.flat.clone()
        lb = ets(dict(Z=b["Z"], R=b["R"].clone(), N=b["N"], **b["idx"]), {"E": b["Et"], "F": b["Ft"]}, step_optimizer=False)
        gb = ets.buf.flat.clone()
        torch.cuda.synchronize()
        assert abs(float(la) - float(lb)) <= 2e-5 * abs(float(lb)), (float(la), float(lb))
        worst = max(worst, float((ga - gb).norm() / gb.norm()))
    print(f"molecule sizes 32 / 24 / 28 through one graph ({sizes}): forward+force bit-identical, training gradients within {worst:.2e}")
    assert worst <= 1e-5


def test_dynamic_force_field_follows_a_moving_system():
    """runtime.DynamicForceField: positions drift from call to call (the neighbour list changes), the capacities start
    tight so that the loop has to grow them at least once — every call equals the eager run on that call's arrays."""
    from gemnet_pytorch_amd.runtime import DynamicForceField
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(2)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    ds = make_dataset(4, 32, config=2, first=50)
    R0 = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
    Z = torch.tensor(ds["Z"], device=DEV).long()
    N = torch.tensor(ds["N"], device=DEV).long()
    ff = DynamicForceField(model, Z, ds["N"], 5.0, 10.0, margin=0.03)
    g = torch.Generator().manual_seed(0)
    sizes = []
    for step in range(8):
        # breathe: scale the molecules about their centres (denser -> more edges), plus a little noise
        c = R0.view(4, 32, 3).mean(dim=1, keepdim=True)
        s = 1.0 - 0.008 * step
        R = ((R0.view(4, 32, 3) - c) * s + c).reshape(-1, 3) + 0.01 * torch.randn(R0.shape, generator=g).to(DEV)
        E, F = ff(R)
        idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
        E0, F0 = model(dict(Z=Z, R=R.clone(), N=N, **idx))
        torch.cuda.synchronize()
        sizes.append(int(idx["id_c"].shape[0]))
        assert torch.equal(E, E0.detach()) and torch.equal(F, F0.detach()), (step, float((F - F0).abs().max()))
    print(f"moving system: edges per step {sizes}, graphs re-captured {ff.recaptures} time(s)")
    assert 1 <= ff.recaptures < 7 and len(set(sizes)) > 2      # grown at least once, replayed in between


def test_trainer_with_padded_graph_follows_the_plain_trainer():
    """Trainer.enable_padded_graph: `train_on_batch` on batches of 8 molecules with 32 / 24 / 28 / 32 atoms from one
    captured graph against the plain Trainer on the same batches (same initial weights): the first step's loss equal to
    fp32 rounding, the trajectories and the tracked metrics together afterwards."""
    import copy
    from gemnet_pytorch_amd.training.metrics import Metrics
    from gemnet_pytorch_amd.training.trainer import Trainer
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(13)
    model_a = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    model_b = copy.deepcopy(model_a)
    g = torch.Generator().manual_seed(8)
    batches = [_batch(8, n, 300 * (i + 1), g) for i, n in enumerate((32, 24, 28, 32))]

    def it():
        i = 0
        while True:
            b = batches[i % len(batches)]
            i += 1
            inputs = dict(Z=b["Z"], R=b["R"].clone(), N=b["N"], **b["idx"])
            yield inputs, {"E": b["Et"], "F": b["Ft"]}
    sizes = [(int(b["idx"]["id_c"].shape[0]), int(b["idx"]["id3_reduce_ca"].shape[0])) for b in batches]
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    out = {}
    for name, model in (("padded", model_a), ("plain", model_b)):
        tr = Trainer(model, learning_rate=1e-3, loss="rmse", rho_force=0.99, grad_clip_max=10.0)
        tr.dict2device = lambda d, device=None: d
        if name == "padded":
            tr.enable_padded_graph(a_cap=8 * 32, e_cap=e_cap, t_cap=t_cap, max_in_degree=31, n_groups=max(1, e_cap // 62))
        m = Metrics("train", tr.tracked_metrics)
        stream = it()
        losses = [float(tr.train_on_batch(stream, m)) for _ in range(6)]
        torch.cuda.synchronize()
        out[name] = (losses, m.result(append_tag=False))
        if name == "padded":
            assert tr._pstep is not None and tr._pstep._captured
    (lp, mp), (le, me) = out["padded"], out["plain"]
    print("padded trainer losses", [f"{v:.6f}" for v in lp], "plain", [f"{v:.6f}" for v in le])
    assert abs(lp[0] - le[0]) <= 2e-5 * abs(le[0])
    for a, b in zip(lp, le):
        assert abs(a - b) <= 2e-3 * abs(b), (lp, le)
    for k in me:
        assert abs(float(mp[k]) - float(me[k])) <= 2e-3 * abs(float(me[k])), (k, float(mp[k]), float(me[k]))


def test_trainer_padded_step_overflow_is_caught_before_the_optimizers(monkeypatch):
    """`Trainer.train_on_batch` on the captured padded step in the fp16-plane arithmetic: an activation driven past 65 504 AFTER
    the capture (the atom embedding is gathered from the live parameter by every replay) must not reach the optimizers — the
    Trainer reads the replayed graph's range flag before they see the gradients (`_train_on_batch_padded`, all ranks decide
    together: `TrainStep._range_check(agreed=True)`), warns, moves the model to the bf16 planes, captures anew and repeats the
    step: the reported loss is finite and every parameter stays finite."""
    import warnings
    from gemnet_pytorch_amd import kernels as K
    from gemnet_pytorch_amd.training.metrics import Metrics
    from gemnet_pytorch_amd.training.trainer import Trainer
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(13)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    g = torch.Generator().manual_seed(8)
    batches = [_batch(8, n, 300 * (i + 1), g) for i, n in enumerate((32, 24, 28))]

    def it():
        i = 0
        while True:
            b = batches[i % len(batches)]
            i += 1
            yield dict(Z=b["Z"], R=b["R"].clone(), N=b["N"], **b["idx"]), {"E": b["Et"], "F": b["Ft"]}
    sizes = [(int(b["idx"]["id_c"].shape[0]), int(b["idx"]["id3_reduce_ca"].shape[0])) for b in batches]
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    tr = Trainer(model, learning_rate=1e-3, loss="rmse", rho_force=0.99, grad_clip_max=10.0)
    tr.dict2device = lambda d, device=None: d
    tr.enable_padded_graph(a_cap=8 * 32, e_cap=e_cap, t_cap=t_cap, max_in_degree=31, n_groups=max(1, e_cap // 62))
    m = Metrics("train", tr.tracked_metrics)
    stream = it()
    l0 = float(tr.train_on_batch(stream, m))
    assert np.isfinite(l0) and tr._pstep is not None and tr._pstep._captured and model.matmul_precision is None
    with torch.no_grad():
        model.atom_emb.embeddings.weight.mul_(1.0e6)
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        l1 = float(tr.train_on_batch(stream, m))
    torch.cuda.synchronize()
    assert model.matmul_precision == "split6" and np.isfinite(l1)
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        l2 = float(tr.train_on_batch(stream, m))          # and stays quiet afterwards
    assert np.isfinite(l2) and tr._pstep.flag.trips == 1


def test_captured_training_step_is_bit_reproducible_and_equals_eager():
    """The captured training step replayed four times from the same weights: identical flat gradients every time, and
    identical to the eager step (the output blocks run in line during force training: with them on the side stream the
    replays differed by 3-7e-4 of the gradient norm from run to run, tools/exp/train_determinism.py)."""
    import copy
    from gemnet_pytorch_amd.training.ddp import TrainStep
    cfg = dict(FULL, triplets_only=True, num_blocks=2)
    torch.manual_seed(9)
    base = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    g = torch.Generator().manual_seed(4)
    b = _batch(8, 32, 100, g)
    inputs = dict(Z=b["Z"], R=b["R"].clone(), N=b["N"], **b["idx"])
    targets = {"E": b["Et"], "F": b["Ft"]}
    flat = {}
    for kind in ("eager", "captured"):
        ts = TrainStep(copy.deepcopy(base), fused_optimizer=True)
        if kind == "captured":
            ts.capture(inputs, targets)
        runs = []
        for _ in range(4):
            ts(inputs, targets, step_optimizer=False)
            torch.cuda.synchronize()
            runs.append(ts.buf.flat.clone())
        assert all(torch.equal(r, runs[0]) for r in runs[1:]), kind
        flat[kind] = runs[0]
    assert torch.equal(flat["eager"], flat["captured"])


def test_padded_quadruplet_replay_equals_eager_on_changing_batches():
    """GemNet-Q (round 5): interaction edges, intermediate triplets and quadruplets padded as well (groups of four dummy atoms) —
    batches of different sizes through ONE captured graph, the real molecules bit-identical to the eager run on the unpadded
    arrays; the capture passes the happens-before checker."""
    cfg = dict(FULL, triplets_only=False, num_blocks=2)
    torch.manual_seed(5)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV).eval()
    model.requires_grad_(False)
    n_mol, n_atoms = 4, 24
    batches = []
    for b in range(3):
        ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
        R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
        Z = torch.tensor(ds["Z"], device=DEV).long()
        N = torch.tensor(ds["N"], device=DEV).long()
        idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, False, device=DEV)(R)
        batches.append((Z, R, N, idx))
    sizes = [PaddedGraphRunner.sizes_of(i) for _, _, _, i in batches]
    assert len(set(sizes)) > 1 and len(sizes[0]) == 5, sizes
    Z, _, N, _ = batches[0]
    e_cap, t_cap, caps = PaddedGraphRunner.suggest_capacities(sizes)
    runner = PaddedGraphRunner(model, Z, N, e_cap, t_cap, quad_caps=caps)
    runner.check = True
    ref = []
    for Zb, R, Nb, idx in batches:
        E, F = model(dict(Z=Zb, R=R.clone(), N=Nb, **idx))
        ref.append((E.detach().clone(), F.detach().clone()))
    for rnd in range(2):
        for (Zb, R, Nb, idx), (E0, F0) in zip(batches, ref):
            E, F = runner(R, idx, Z=Zb)
            torch.cuda.synchronize()
            assert E.shape == E0.shape and F.shape == F0.shape
            assert torch.equal(E, E0) and torch.equal(F, F0), (float((F - F0).abs().max()), float(F0.abs().max()))
    races, summary = runner.hb.races(), runner.hb.summary()
    print(runner.hb.format(races))
    assert not races and summary["unrecorded_nodes"] == 0 and summary["unresolved_pointers"] == 0
    print(f"padded GemNet-Q replay == eager, bit for bit, over batches {sizes} at capacities {(runner.e_cap, runner.t_cap) + runner.quad_caps}")
    int32_idx = DeviceGraphBuilder(N.cpu().numpy(), 5.0, 10.0, False, device=DEV)
    E, F = runner.build_and_run(int32_idx, batches[1][1], Z=batches[1][0])
    torch.cuda.synchronize()
    assert torch.equal(E, ref[1][0]) and torch.equal(F, ref[1][1])


def test_padded_training_step_at_the_headline_batch_keeps_finite_gradients_in_the_fp16_plane_arithmetic(monkeypatch):
    """The dummy molecule must stay inside the fp16 planes of the default arithmetic.  With 1 A dummy bonds the pad edges that
    carry all pad triplets (24-45 identical ones each) reached 7e4 in the bilinear layer's output at this batch size; their
    zero cotangents times inf made every weight gradient of the padded training step NaN (found by the range flag in round 5).
    The dummy bonds now sit at 0.9 x the cutoff (`padded.dummy_positions`)."""
    from gemnet_pytorch_amd import kernels as K
    from gemnet_pytorch_amd.training.ddp import PaddedTrainStep
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    cfg = dict(FULL, triplets_only=True)
    g = torch.Generator().manual_seed(1)
    ds = make_dataset(32, 32, config=2, first=32)
    R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
    idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
    Z, N = torch.tensor(ds["Z"], device=DEV).long(), torch.tensor(ds["N"], device=DEV).long()
    Et, Ft = torch.randn(32, 1, generator=g).to(DEV), torch.randn(1024, 3, generator=g).to(DEV)
    sizes = [(int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0]))]
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    ts = PaddedTrainStep(model, Z, N, *PaddedGraphRunner.suggest_capacities(sizes), fused_optimizer=True)
    for _ in range(3):
        loss = ts.step(R, idx, Et, Ft, Z=Z)
        torch.cuda.synchronize()
        assert ts.flag.tripped() == 0 and model.matmul_precision is None
        assert bool(torch.isfinite(loss)) and bool(torch.isfinite(ts.fused.flat_p).all())
    assert bool(torch.isfinite(ts.buf.flat).all())
