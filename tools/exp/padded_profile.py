"""The padded-capacity replay loop of bench.py's extra.dynamic_shape on its own (for rocprofv3 --kernel-trace --stats).
   PYTHONPATH=. python tools/exp/padded_profile.py [steps]"""
import sys
import time
import torch
import bench
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.padded import PaddedGraphRunner
from gemnet_pytorch_amd.synthetic import make_dataset

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_fullsize import FULL  # noqa: E402
cfg = dict(FULL, triplets_only=True)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
n_mol, n_atoms, nb = 32, 32, 4
data, builders = [], []
for b in range(nb):
    ds = make_dataset(n_mol, n_atoms, config=2, first=(b + 1) * n_mol)
    data.append(dict(R=torch.tensor(ds["R"], device=dev), Z=torch.tensor(ds["Z"], device=dev).long(),
                     N=torch.tensor(ds["N"], device=dev).long()))
    builders.append(DeviceGraphBuilder(ds["N"], 5.0, 10.0, True, device=dev))
idxs = [builders[b](data[b]["R"]) for b in range(nb)]
sizes = [(int(i["id_c"].shape[0]), int(i["id3_reduce_ca"].shape[0])) for i in idxs]
runner = PaddedGraphRunner(model, data[0]["Z"], data[0]["N"], *PaddedGraphRunner.suggest_capacities(sizes))
for b in range(nb):
    runner(data[b]["R"], idxs[b], Z=data[b]["Z"])
torch.cuda.synchronize()
t_build = t_fill = t_replay = 0.0
t0 = time.perf_counter()
for i in range(steps):
    b = i % nb
    t1 = time.perf_counter()
    idx = builders[b](data[b]["R"])
    t2 = time.perf_counter()
    runner._fill(data[b]["R"], idx, data[b]["Z"])
    t3 = time.perf_counter()
    runner.graph.replay()
    t4 = time.perf_counter()
    t_build += t2 - t1; t_fill += t3 - t2; t_replay += t4 - t3
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"padded loop: {el / steps * 1e3:.3f} ms/step; host time per step: index build {t_build / steps * 1e3:.3f} ms "
      f"(with its read-back), fill {t_fill / steps * 1e3:.3f} ms, replay call {t_replay / steps * 1e3:.3f} ms; sizes {sizes}, "
      f"capacities ({runner.e_cap}, {runner.t_cap}), {3 * runner.G} dummy atoms")
