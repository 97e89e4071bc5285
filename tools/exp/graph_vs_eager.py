import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.graph import GraphPlan
cfg = dict(bench.GEMNET_T)
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
if "nooverlap" in sys.argv:
    model.overlap_output_blocks = False
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
GraphPlan.from_inputs(inputs, True).warm()
step = lambda: model(inputs)
E1, F1 = step(); E2, F2 = step()
torch.cuda.synchronize()
print("eager vs eager: E", float((E1 - E2).abs().max()), "F", float((F1 - F2).abs().max()))
graph, out = bench.capture(step)
E3, F3 = step()
torch.cuda.synchronize()
print("graph vs eager: E", float((out[0] - E3).abs().max()), "F", float((out[1] - F3).abs().max()), "scale E", float(E3.abs().max()), "F", float(F3.abs().max()))
for i in range(3):
    graph.replay(); torch.cuda.synchronize()
    print(" replay", i, "E", float((out[0] - E3).abs().max()), "F", float((out[1] - F3).abs().max()))
