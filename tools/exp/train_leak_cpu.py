"""Does an EAGER force-training step free everything it allocated?  Runs the fused training path (ops_train.py) on the CPU
emulation of the launchers with the cyclic garbage collector OFF and counts the tensors / sweep records alive after each step.
(Found in round 4: records of the second-order sweeps kept themselves alive through autograd's C++ edges.)

    PYTHONPATH=.:tests python tools/exp/train_leak_cpu.py [tag]
"""
import gc
import sys

import numpy as np
import torch

import cpu_kernels
from oracle import gemnet_oracle as GO
from gemnet_pytorch_amd import ops_train
from test_model_cpu import build
from test_oracle_model import load_case

tag = sys.argv[1] if len(sys.argv) > 1 else "t2s"
g = np.load("tests/golden/model2.npz", allow_pickle=True)
cfg, params, inputs = load_case(g, tag)


def alive():
    objs = gc.get_objects()
    ts = [o for o in objs if isinstance(o, torch.Tensor)]
    return len(ts), sum(t.numel() * t.element_size() for t in ts), sum(isinstance(o, ops_train._Rec) for o in objs)


with cpu_kernels.emulate():
    model = build(cfg, params).train()
    inputs["R"] = inputs["R"].double()
    Et, Ft = torch.tensor(g[f"{tag}.Et"]).double()[:, None], torch.tensor(g[f"{tag}.Ft"]).double()

    def step():
        model.zero_grad(set_to_none=True)
        E, F = model(inputs)
        loss = GO.training_loss(E[:, :1], F[:, 0] if F.dim() == 3 else F, Et, Ft)
        loss.backward()
        return float(loss.detach())

    step()
    gc.collect()
    gc.disable()
    base = alive()
    print("after the first step (tensors, bytes, records):", base)
    for i in range(3):
        step()
        print(f"step {i + 2}:", alive())
