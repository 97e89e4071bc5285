"""Which torch (non-HIP-library) kernels run in one forward+force step, by op and input shape (GPU box)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
cfg = dict(bench.GEMNET_T)
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
for _ in range(2):
    model(inputs)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    model(inputs); torch.cuda.synchronize()
tot = 0.0
for ev in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
    if ev.key.startswith("aten::") and ev.self_device_time_total > 0:
        tot += ev.self_device_time_total
        print(f"{ev.count:4d} x {ev.key:24s} {str(ev.input_shapes)[:80]:80s} {ev.self_device_time_total:8.1f} us")
print(f"total torch-op device time: {tot:.1f} us")
