"""GPU diagnosis: second-order parameter gradients of a golden case in the fused training form (train2) and on the
composite closure, each against the reference's probes, and against each other.   python tools/exp/t2_probe_gpu.py q4s"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import SCALE_FILE, grad_probes  # noqa: E402
from oracle import gemnet_oracle as GO  # noqa: E402
from gemnet_pytorch_amd import ops  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from test_oracle_model import load_case  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "q4s"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "model2.npz")))
cfg, params, inputs = load_case(g, tag)
grads = {}
for t2 in (True, False):
    ops.USE_TRAIN2 = t2
    model = GemNet(**cfg, scale_file=SCALE_FILE)
    model.load_state_dict(GO.expand_to_reference_state_dict({k: v.float() for k, v in params.items()}), strict=True)
    model = model.to("cuda").train()
    E, F = model({k: v.to("cuda") for k, v in inputs.items()})
    loss = GO.training_loss(E[:, :1], F, torch.tensor(g[f"{tag}.Et"], device="cuda")[:, None], torch.tensor(g[f"{tag}.Ft"], device="cuda"))
    loss.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in g[f"{tag}.grad_names"]]
    ref_proj, ref_norm = g[f"{tag}.grad_proj"], g[f"{tag}.grad_norms"]
    rows = []
    for i, n in enumerate(names):
        gr = named[n].grad
        v = grad_probes(n, gr.numel()) @ gr.detach().double().cpu().reshape(-1).numpy()
        rows.append((float(np.abs(v - ref_proj[i]).max()) / max(float(ref_norm[i]), 1e-6 * float(ref_norm.max())), n,
                     float(gr.norm()) / float(ref_norm[i]) - 1.0))
    rows.sort(reverse=True)
    print("train2" if t2 else "composite", f"loss {loss.item():.6f} (ref {float(g[f'{tag}.loss']):.6f})")
    for r in rows[:6]:
        print(f"   probe err / ||g_ref|| = {r[0]:.2e}   norm rel err {r[2]:+.2e}   {r[1]}")
    grads[t2] = {n: named[n].grad.detach().clone() for n in names}
worst = max((float((grads[True][n] - grads[False][n]).norm()) / (float(grads[False][n].norm()) + 1e-30), n) for n in grads[True])
print("train2 vs composite: worst relative gradient difference", worst)
