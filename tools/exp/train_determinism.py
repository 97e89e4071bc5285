"""Run-to-run reproducibility of the training step's gradients (eager TrainStep, captured TrainStep, PaddedTrainStep) on
one batch: the same step several times from the same weights, flat gradient buffers compared bit for bit.
   PYTHONPATH=.:tests python tools/exp/train_determinism.py"""
import copy
import torch
from conftest import SCALE_FILE
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from gemnet_pytorch_amd.training.ddp import PaddedTrainStep, TrainStep
from test_gpu_fullsize import FULL

DEV = "cuda"
cfg = dict(FULL, triplets_only=True, num_blocks=2)
torch.manual_seed(9)
base = GemNet(**cfg, scale_file=SCALE_FILE).to(DEV)
import os
if os.environ.get("T_NO_OVERLAP"):
    base.overlap_output_blocks = False
ds = make_dataset(8, 32, config=2, first=100)
R = torch.tensor(ds["R"], device=DEV, dtype=torch.float32)
Z = torch.tensor(ds["Z"], device=DEV).long()
N = torch.tensor(ds["N"], device=DEV).long()
idx = DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=DEV)(R)
g = torch.Generator().manual_seed(4)
Et, Ft = torch.randn(8, 1, generator=g).to(DEV), torch.randn(256, 3, generator=g).to(DEV)
E, T = int(idx["id_c"].shape[0]), int(idx["id3_reduce_ca"].shape[0])


def grads(kind, reps=4):
    out = []
    model = copy.deepcopy(base)
    if kind == "padded":
        ts = PaddedTrainStep(model, Z, N, E + 200, T + 4000, fused_optimizer=True)
        run = lambda: ts.step(R, idx, Et, Ft, Z=Z, step_optimizer=False)
    else:
        ts = TrainStep(model, fused_optimizer=True)
        inputs = dict(Z=Z, R=R.clone(), N=N, **idx)
        targets = {"E": Et, "F": Ft}
        if kind == "captured":
            ts.capture(inputs, targets)
        run = lambda: ts(inputs, targets, step_optimizer=False)
    for _ in range(reps):
        run()
        torch.cuda.synchronize()
        out.append(ts.buf.flat.clone())
    return out


ref = None
for kind in ("eager", "captured") + (("padded",) if not os.environ.get("T_SHORT") else ()):
    gs = grads(kind)
    dev = [float((x - gs[0]).norm() / gs[0].norm()) for x in gs[1:]]
    if ref is None:
        ref = gs[0]
    print(f"{kind:9s}: run-to-run relative deviation of the flat gradient {['%.2e' % d for d in dev]}; "
          f"vs the first eager run {float((gs[0] - ref).norm() / ref.norm()):.2e}")
