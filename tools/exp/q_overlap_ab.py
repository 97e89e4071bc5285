"""GPU: GemNet-Q 32 x 32 forward+force (hipGraph replay) with the output blocks on the side stream vs. in line."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.GEMNET_T, triplets_only=False)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
for overlap in (True, False, True, False):
    model.overlap_output_blocks = overlap
    for _ in range(2):
        model(inputs)
    graph, _ = bench.capture(lambda: model(inputs))
    el = bench.time_steps(graph.replay, 10, 3)
    print(f"GemNet-Q 32x32 forward+force, output blocks on the side stream = {overlap}: {el / 10 * 1e3:.3f} ms/step", flush=True)
    del graph
