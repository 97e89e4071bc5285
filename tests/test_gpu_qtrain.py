"""-m gpu: GemNet-Q force TRAINING on the fused angle-form twins (ops_train._QuadAngles2 / _BilinearAng2, round 5) inside the
data-parallel training step (training/ddp.py::TrainStep: grouped weight gradients, fused optimizer, positions' second-order
terms off — the configuration of bench.py's extra.gemnet_q.train_step):
  * the flat parameter gradient equals the one of the COMPOSITE closure (the round-4 form: (Q, 49) harmonics materialised,
    every op differentiated twice by autograd) on the same weights, to fp32 rounding;
  * the captured step replays bit-identically to itself and equals the eager step;
  * a few optimizer steps keep the two forms on the same loss trajectory."""
import copy

import pytest
import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd import ops
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.data_container import DataContainer
from gemnet_pytorch_amd.training.ddp import TrainStep
from test_gpu_fullsize import FULL

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(n_mol, n_atoms):
    ds = make_dataset(n_mol, n_atoms, config=2)
    b = DataContainer.from_arrays(dict(ds), 5.0, 10.0, triplets_only=False)[list(range(n_mol))]
    inputs = {k: v.to(DEV) for k, v in b.items() if k not in ("E", "F")}
    g = torch.Generator().manual_seed(2)
    targets = {"E": torch.randn(n_mol, 1, generator=g).to(DEV), "F": torch.randn(n_mol * n_atoms, 3, generator=g).to(DEV)}
    return inputs, targets


@pytest.mark.parametrize("n_mol,n_atoms,blocks", [(4, 24, 2), (8, 32, 4)])
def test_fused_quadruplet_training_step_equals_the_composite_closure(n_mol, n_atoms, blocks, monkeypatch):
    cfg = dict(FULL, triplets_only=False, num_blocks=blocks)
    torch.manual_seed(13)
    base = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
    inputs, targets = _batch(n_mol, n_atoms)
    grads, losses, steps = {}, {}, {}
    for form in ("twins", "composite"):
        monkeypatch.setattr(ops, "USE_TRAIN2_QUAD", form == "twins")
        ts = TrainStep(copy.deepcopy(base), fused_optimizer=True)
        batch = {k: v for k, v in inputs.items()}
        losses[form] = float(ts(batch, targets, step_optimizer=False))
        torch.cuda.synchronize()
        grads[form] = ts.buf.flat.clone()
        assert bool(torch.isfinite(grads[form]).all())
        if form == "twins":
            ts.capture(batch, targets)
            for _ in range(3):
                lg = float(ts(batch, targets, step_optimizer=False))
                torch.cuda.synchronize()
                assert torch.equal(ts.buf.flat, grads[form]) and lg == losses[form]      # replay == eager, bit for bit
        traj = []
        for _ in range(4):
            traj.append(float(ts(batch, targets)))
        torch.cuda.synchronize()
        steps[form] = traj
    rel = float((grads["twins"] - grads["composite"]).norm() / grads["composite"].norm())
    worst = float((grads["twins"] - grads["composite"]).abs().max() / grads["composite"].abs().max())
    print(f"GemNet-Q {n_mol} x {n_atoms}, {blocks} blocks: loss {losses['twins']:.6f} / {losses['composite']:.6f}; flat gradient twins vs "
          f"composite: {rel:.2e} of the norm, worst element {worst:.2e} of the largest; loss trajectories {steps}")
    assert abs(losses["twins"] - losses["composite"]) <= 2e-5 * abs(losses["composite"])
    assert rel <= 2e-3 and worst <= 5e-3
    for a, b in zip(steps["twins"], steps["composite"]):
        assert abs(a - b) <= 2e-2 * abs(b), steps
