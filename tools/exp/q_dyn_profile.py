"""GemNet-Q padded-capacity replay loop, part by part (device time, synchronised between the parts), for
rocprofv3 --kernel-trace --stats.      PYTHONPATH=. python tools/exp/q_dyn_profile.py [steps]"""
import sys
import time

import torch

import bench
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
cfg = dict(bench.GEMNET_T, triplets_only=False)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
n_mol, n_atoms, nb = 32, 32, 3
data, builders = [], []
for b in range(nb):
    ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
    data.append(dict(R=torch.tensor(ds["R"], device=dev), Z=torch.tensor(ds["Z"], device=dev).long(),
                     N=torch.tensor(ds["N"], device=dev).long()))
    builders.append(DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], False, device=dev))
idxs = [builders[b](data[b]["R"]) for b in range(nb)]
sizes = [PaddedGraphRunner.sizes_of(i) for i in idxs]
caps = PaddedGraphRunner.suggest_capacities(sizes)
runner = PaddedGraphRunner(model, data[0]["Z"], data[0]["N"], caps[0], caps[1], quad_caps=caps[2])
for b in range(nb):
    runner(data[b]["R"], idxs[b], Z=data[b]["Z"])
torch.cuda.synchronize()
t = dict(build=0.0, fill=0.0, replay=0.0)
for i in range(steps):
    b = i % nb
    t1 = time.perf_counter()
    idx = builders[b](data[b]["R"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    runner._fill(data[b]["R"], idx, data[b]["Z"])
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    runner.graph.replay()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    t["build"] += t2 - t1
    t["fill"] += t3 - t2
    t["replay"] += t4 - t3
print("GemNet-Q padded loop, parts synchronised: " + ", ".join(f"{k} {v / steps * 1e3:.3f} ms" for k, v in t.items()),
      f"; sizes {sizes[0]}, capacities {caps}")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    b = i % nb
    runner.build_and_run(builders[b], data[b]["R"], Z=data[b]["Z"], positions_ready=True)
torch.cuda.synchronize()
print(f"build_and_run loop: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
# the same loop with the HOST time of each part (no synchronisation added)
h = dict(build=0.0, fill=0.0, replay=0.0)
main = torch.cuda.current_stream()
bs = runner._bstream
t0 = time.perf_counter()
for i in range(steps):
    b = i % nb
    t1 = time.perf_counter()
    with torch.cuda.stream(bs):
        idx = builders[b](data[b]["R"], dtype=runner.index_dtype)
    main.wait_stream(bs)
    for v in idx.values():
        v.record_stream(main)
    t2 = time.perf_counter()
    runner._fill(data[b]["R"], idx, data[b]["Z"])
    t3 = time.perf_counter()
    runner.graph.replay()
    t4 = time.perf_counter()
    h["build"] += t2 - t1
    h["fill"] += t3 - t2
    h["replay"] += t4 - t3
torch.cuda.synchronize()
print(f"host time per part in the free-running loop ({(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step): "
      + ", ".join(f"{k} {v / steps * 1e3:.3f} ms" for k, v in h.items()))
