"""BASELINE.json configs[4], one GPU's shard — GemNet-Q, 64 molecules x 64 atoms (batch 512 over 8 GPUs), forward+force in the
default arithmetic — as a stand-alone workload for the counter passes of tools/gpu_artifacts.sh (`rocprofv3 --pmc ... --
python tools/config4_shard.py`: two eager steps, no hipGraph).  `python tools/config4_shard.py families` instead runs
bench.extra_config4_shard (timing under a captured graph + the launcher families' algorithmic bytes, dumped under
GEMNET_DUMP_FAMILIES)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if "families" in sys.argv:
    print(json.dumps(bench.extra_config4_shard(0)))
    sys.exit(0)
from gemnet_pytorch_amd.graph import GraphPlan  # noqa: E402
from gemnet_pytorch_amd.index_device import DeviceGraphBuilder  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402
from gemnet_pytorch_amd.synthetic import make_dataset  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.GEMNET_T, triplets_only=False)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
ds = make_dataset(64, 64, config=4)
R = torch.tensor(ds["R"], device=dev)
idx = DeviceGraphBuilder(ds["N"], cfg["cutoff"], cfg["int_cutoff"], False, device=dev)(R)
inputs = dict(Z=torch.tensor(ds["Z"], device=dev).long(), R=R, N=torch.tensor(ds["N"], device=dev).long(), **idx)
GraphPlan.from_inputs(inputs, False).warm()
for _ in range(2):
    E, F = model(inputs)
torch.cuda.synchronize()
print("configs[4] shard: 2 eager steps done; mean|F| =", float(F.abs().mean()))
