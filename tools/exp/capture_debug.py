import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.training.ddp import TrainStep
which = sys.argv[1]
cfg = dict(bench.GEMNET_T)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev)
inputs, targets = bench.make_batch(cfg, 8, 32, first=0, device=dev)
ts = TrainStep(model, world_size=1, fused_optimizer=(which != "torchopt"))
if which == "noqueue":
    ts.wgrad = None
if which == "eagerfirst":
    ts(inputs, targets); ts(inputs, targets); torch.cuda.synchronize()
print("capturing", which, flush=True)
ts.capture(inputs, targets)
print("captured", flush=True)
for _ in range(3):
    l = ts(inputs, targets)
torch.cuda.synchronize()
print("ok", which, float(l), flush=True)
