"""Experiment: two molecule shards of the batch on two concurrent HIP streams inside one hipGraph."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gemnet_pytorch_amd.graph import GraphPlan
from gemnet_pytorch_amd.model.gemnet import GemNet

dev = torch.device("cuda", 0)
cfg = dict(bench.GEMNET_T)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model.overlap_output_blocks = (len(sys.argv) > 2 and sys.argv[2] == "side")
per = 32 // nstreams
shards = []
for k in range(nstreams):
    si, _ = bench.make_batch(cfg, per, 32, first=k * per, device=dev)
    GraphPlan.from_inputs(si, True).warm()
    shards.append((si, torch.cuda.Stream(device=dev)))


def step():
    main = torch.cuda.current_stream()
    outs = []
    for si, st in shards:
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(model(si))
    for _, st in shards:
        main.wait_stream(st)
    return outs


for _ in range(3):
    step()
torch.cuda.synchronize()
print("eager ok", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"streams={nstreams} side={model.overlap_output_blocks}: {dt * 1e3:.3f} ms/step -> {32 / dt:.0f} mol/s", flush=True)
