"""The padded-capacity replay loop of bench.py's extra.dynamic_shape alone (a new batch every step: device index build,
padding, ONE hipGraph replayed), for same-box A/Bs of switches that only matter there:

    PYTHONPATH=.:tests GEMNET_PLAN_LATE=0 python tools/exp/padded_ab.py     adjoint-only index structures in line
    PYTHONPATH=.:tests GEMNET_PLAN_LATE=1 python tools/exp/padded_ab.py     ... on a stream of their own (default)
"""
import os
import sys
import time

import torch

from conftest import SCALE_FILE
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset
from test_gpu_fullsize import FULL

dev = torch.device("cuda")
n_mol, n_atoms, n_batches = 32, 32, 4
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
model = GemNet(**dict(FULL, triplets_only=True), scale_file=SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
data = []
for b in range(n_batches):
    ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
    data.append(dict(R=torch.tensor(ds["R"], device=dev), Z=torch.tensor(ds["Z"], device=dev).long(),
                     N=torch.tensor(ds["N"], device=dev).long(), N_host=ds["N"]))
builders = [DeviceGraphBuilder(d["N_host"], 5.0, 10.0, True, device=dev) for d in data]
idxs = [builders[b](data[b]["R"]) for b in range(n_batches)]
sizes = [(int(i["id_c"].shape[0]), int(i["id3_reduce_ca"].shape[0])) for i in idxs]
e_cap, t_cap = PaddedGraphRunner.suggest_capacities(sizes)
runner = PaddedGraphRunner(model, data[0]["Z"], data[0]["N"], e_cap, t_cap)
state = {"i": 0}


def pstep(ready=True):
    b = state["i"] % n_batches
    state["i"] += 1
    return runner.build_and_run(builders[b], data[b]["R"], Z=data[b]["Z"], positions_ready=ready)


for ready in (True, False):
    for _ in range(6):
        pstep(ready)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pstep(ready)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"GEMNET_PLAN_LATE={os.environ.get('GEMNET_PLAN_LATE', '1')} positions_ready={ready}: {el / steps * 1e3:.3f} ms/step, "
          f"{n_mol * steps / el:.0f} molecules/s")
# replay alone (no index build, no padding): what the graph itself costs
for _ in range(5):
    runner.graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    runner.graph.replay()
torch.cuda.synchronize()
print(f"   replay alone: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms")
E0, F0 = model(dict(Z=data[0]["Z"], R=data[0]["R"].clone(), N=data[0]["N"], **idxs[0]))
E1, F1 = runner(data[0]["R"], idxs[0], Z=data[0]["Z"])
torch.cuda.synchronize()
print(f"   max |F - F_eager| {float((F1 - F0).abs().max()):.3e}")
