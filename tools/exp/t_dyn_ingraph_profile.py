"""GemNet-T new-batch loop with the index build inside the graph, for rocprofv3 --kernel-trace.   PYTHONPATH=. python tools/exp/t_dyn_ingraph_profile.py [steps]"""
import sys
import time

import torch

import bench
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
cfg = dict(bench.GEMNET_T)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
data = []
for b in range(3):
    ds = make_dataset(32, 32, config=2, first=(b + 1) * 32)
    data.append(dict(R=torch.tensor(ds["R"], device=dev), Z=torch.tensor(ds["Z"], device=dev).long(),
                     N=torch.tensor(ds["N"], device=dev).long(), N_host=ds["N"]))
builder = DeviceGraphBuilder(data[0]["N_host"], cfg["cutoff"], cfg["int_cutoff"], True, device=dev)
idxs = [builder(d["R"], dtype=torch.int32) for d in data]
caps = PaddedGraphRunner.suggest_capacities([PaddedGraphRunner.sizes_of(i) for i in idxs])
runner = PaddedGraphRunner(model, data[0]["Z"], data[0]["N"], *caps)
runner._fill(data[0]["R"], idxs[0], data[0]["Z"])
runner.attach_builder(builder)
for i in range(6):
    runner.run_positions(data[i % 3]["R"], Z=data[i % 3]["Z"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    runner.run_positions(data[i % 3]["R"], Z=data[i % 3]["Z"])
torch.cuda.synchronize()
print(f"in-graph index loop: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
