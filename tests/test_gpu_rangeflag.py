"""-m gpu: the range check of REPLAYED graphs (runtime.RangeFlag, gn_nonfinite_flag_f32).

The default Dense arithmetic ("h3") keeps activations in two fp16 planes; the reference's fp32 has no overflow cliff
(base_layers.py:44-48).  An eager forward reads its outputs back and falls back to the bf16 planes; a captured hipGraph
replays what was valid at capture time.  Every test here captures a graph on a healthy model and THEN drives an activation
past 65 504 (the atom embedding table is gathered from the live parameter by every replay) — the MD loop, the padded
runner and the captured training step must notice without a host read-back inside the graph, warn, fall back and recover."""
import warnings

import numpy as np
import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd import kernels as K
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.md import DeviceMolecule
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.runtime import ForceGraphs, RangeFlag
from gemnet_pytorch_amd.synthetic import make_dataset, make_molecule
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.ddp import TrainStep

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = dict(num_spherical=7, num_radial=6, num_blocks=2, emb_size_atom=128, emb_size_edge=128, emb_size_trip=64,
           emb_size_quad=32, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32, emb_size_bil_trip=64,
           emb_size_bil_quad=32, num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=2, triplets_only=True)
BLOW = 1.0e6


def new_model(seed=3, blown=False, precision=None):
    torch.manual_seed(seed)
    m = GemNet(**CFG, scale_file=SCALE_FILE).to(DEV).eval()
    m.matmul_precision = precision
    if blown:
        blow(m)
    return m


def blow(model):
    with torch.no_grad():
        model.atom_emb.embeddings.weight.mul_(BLOW)       # in place: the captured gather reads this storage at every replay


def test_flag_kernel_and_mirror():
    f = RangeFlag(torch.device(DEV))
    x = torch.randn(100_003, device=DEV)
    f.watch(x)
    torch.cuda.synchronize()
    assert f.tripped() == 0
    x[77_777] = float("inf")
    f.watch(x, bit=RangeFlag.GRAD)
    torch.cuda.synchronize()
    assert f.tripped() == RangeFlag.GRAD
    f.watch(torch.full((5,), float("nan"), device=DEV))
    f.watch(torch.ones(3, device=DEV))                   # sticky: a later finite tensor does not clear it
    torch.cuda.synchronize()
    assert f.tripped() == RangeFlag.GRAD | RangeFlag.OUTPUT
    f.reset()
    torch.cuda.synchronize()
    assert f.tripped() == 0
    # the flag launches and the mirror copy are capturable
    g = torch.cuda.CUDAGraph()
    y = torch.ones(1000, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f.watch(y)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        f.watch(y)
    g.replay()
    torch.cuda.synchronize()
    assert f.tripped() == 0
    y[5] = float("-inf")
    g.replay()
    torch.cuda.synchronize()
    assert f.tripped() == RangeFlag.OUTPUT


def _batch(n_mol=4, n_atoms=24, first=0):
    ds = make_dataset(n_mol, n_atoms, config=2, first=first)
    R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
    Z = torch.tensor(ds["Z"], device=DEV).long()
    N = torch.tensor(ds["N"], device=DEV).long()
    idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
    return ds, Z, R, N, idx


def test_padded_replay_overflow_after_capture_warns_and_falls_back(monkeypatch):
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    _, Z, R, N, idx = _batch()
    sizes = [(int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0]))]
    e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
    model = new_model()
    runner = PaddedGraphRunner(model, Z, N, e_cap, t_cap)
    E0, F0 = runner(R, idx)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(F0).all()) and runner.flag.tripped() == 0 and model.matmul_precision is None
    blow(model)                                                        # AFTER the capture
    E1, F1 = runner(R, idx)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(F1).all()), "the trigger of this test: the replay must overflow the fp16 planes"
    assert runner.flag.tripped() & RangeFlag.OUTPUT                     # seen without any read-back of E / F
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        E2, F2 = runner(R, idx)                                        # the next call recovers: warn, split6, new capture
    torch.cuda.synchronize()
    assert model.matmul_precision == "split6" and runner.flag.trips == 1
    assert bool(torch.isfinite(E2).all()) and bool(torch.isfinite(F2).all())
    ref = new_model(blown=True, precision="split6")
    E_ref, F_ref = ref(dict(Z=Z, R=R.clone(), N=N, **idx))
    scale = float(F_ref.abs().max())
    assert float((F2 - F_ref.detach()).abs().max()) <= 1e-5 * scale, (float((F2 - F_ref.detach()).abs().max()), scale)
    assert float((E2 - E_ref.detach()).abs().max()) <= 1e-5 * float(E_ref.abs().max())
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        runner(R, idx)                                                 # and stays quiet afterwards
        torch.cuda.synchronize()
    assert runner.flag.tripped() == 0


def test_md_step_overflow_in_a_replayed_graph_is_repeated_on_the_bf16_planes(monkeypatch):
    """`GNNCalculator.calculate` (ase_calculator.py:148-170) -> `model.predict(molecule.get())`: the step that overflows
    returns FINITE results (predict reads them on the host anyway: after that copy the flag is exact)."""
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    mol = make_molecule(24, 11)
    R0, Z = mol["R"].astype(np.float32), mol["Z"]
    model = new_model()
    was = [p.requires_grad for p in model.parameters()]
    dm = DeviceMolecule(R0, Z, 5.0, 10.0, triplets_only=True)
    dm.to(DEV)
    E0, F0 = model.predict(dm.get())
    assert bool(torch.isfinite(F0).all())
    assert [p.requires_grad for p in model.parameters()] == was        # predict() does not freeze the caller's model
    blow(model)
    dm.update(R0 + 0.01)
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        E1, F1 = model.predict(dm.get())
    assert model.matmul_precision == "split6"
    assert bool(torch.isfinite(E1).all()) and bool(torch.isfinite(F1).all())
    ref = new_model(blown=True, precision="split6")
    dc = DataContainer.from_arrays(dict(R=R0 + 0.01, Z=Z, N=np.array([len(Z)]), E=np.zeros((1, 1), np.float32),
                                        F=np.zeros((len(Z), 3), np.float32)), 5.0, 10.0, triplets_only=True)
    b = {k: v.to(DEV) for k, v in dc[[0]].items() if k not in ("E", "F")}
    E_ref, F_ref = ref(b)
    np.testing.assert_allclose(F1.numpy(), F_ref.detach().cpu().numpy(), rtol=0, atol=2e-5 * float(F_ref.abs().max()))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        E2, F2 = model.predict(dm.get())
    assert torch.equal(E1, E2) and torch.equal(F1, F2)


def test_force_graphs_overflow_is_noticed_at_the_next_replay(monkeypatch):
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    ds, Z, R, N, idx = _batch(first=8)
    model = new_model()
    fg = ForceGraphs(model, [dict(Z=Z, R=R, N=N, **idx)])
    (E0, F0), = fg.replay()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(F0).all()) and fg.flag.tripped() == 0
    blow(model)
    fg.replay()
    torch.cuda.synchronize()
    assert fg.flag.tripped() & RangeFlag.OUTPUT
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        (E2, F2), = fg.replay()
    torch.cuda.synchronize()
    assert model.matmul_precision == "split6" and bool(torch.isfinite(F2).all())


def test_captured_training_step_overflow_skips_the_update_and_recaptures(monkeypatch):
    monkeypatch.setattr(K, "DEFAULT_CHAIN_MODE", "h3")
    ds, Z, R, N, idx = _batch(n_mol=4, n_atoms=16, first=16)
    model = new_model(seed=9).train()
    inputs = dict(Z=Z, R=R, N=N, **idx)
    targets = {"E": torch.tensor(ds["E"], device=DEV, dtype=torch.float32).reshape(-1, 1),
               "F": torch.tensor(ds["F"], device=DEV, dtype=torch.float32)}
    ts = TrainStep(model, fused_optimizer=True)
    ts(inputs, targets)                       # one eager step (packs, optimizer state)
    ts.capture(inputs, targets)
    l0 = float(ts(inputs, targets))
    torch.cuda.synchronize()
    assert np.isfinite(l0) and ts.flag.tripped() == 0
    blow(model)                               # the captured step repacks the weights at every replay: it sees this
    before = ts.fused.flat_p.clone()
    ts(inputs, targets)
    torch.cuda.synchronize()
    assert ts.flag.tripped() & RangeFlag.GRAD, ts.flag.tripped()          # the optimizer saw a non-finite gradient norm ...
    assert torch.equal(ts.fused.flat_p, before)                             # ... and left parameters, moments and EMA alone
    assert bool(torch.isfinite(ts.fused.flat_p).all())
    with pytest.warns(RuntimeWarning, match="fp16-plane"):
        l2 = float(ts(inputs, targets))       # polled at the next step: warn, bf16 planes, the step captured anew
    torch.cuda.synchronize()
    assert model.matmul_precision == "split6" and np.isfinite(l2)
    assert ts.flag.tripped() == 0 and not torch.equal(ts.fused.flat_p, before)
    assert bool(torch.isfinite(ts.fused.flat_p).all())
