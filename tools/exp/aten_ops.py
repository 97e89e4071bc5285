"""Which torch (aten) kernels run inside one forward+force step, and from which Python line?  GPU, eager, 1 step."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as B
import __graft_entry__ as ge
ge.build()
from gemnet_pytorch_amd.graph import GraphPlan
from gemnet_pytorch_amd.model.gemnet import GemNet
dev = torch.device("cuda", 0)
cfg = dict(B.GEMNET_T, triplets_only=(len(sys.argv) < 2 or sys.argv[1] != "Q"))
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=B.SCALE_FILE).to(dev)
inputs, _ = B.make_batch(cfg, 32, 32, first=0, device=dev)
GraphPlan.from_inputs(inputs, cfg["triplets_only"]).warm()
model.eval(); model.requires_grad_(False)
for _ in range(2): model(inputs)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    model(inputs); torch.cuda.synchronize()
rows = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0: continue
    if any(k.name.startswith("aten::") and k is not ev for k in (ev.cpu_children or [])): continue   # leaf aten ops only
    src = next((s for s in (ev.stack or []) if "gemnet_pytorch_amd" in s), (ev.stack or ["<autograd engine>"])[0] if ev.stack else "<autograd engine>")
    key = (ev.name, str(ev.input_shapes)[:60], src.split("/root/repo/")[-1].split("repo/")[-1][:90])
    rows[key] += 1; tim[key] += ev.device_time_total
tot = 0
for k, c in sorted(rows.items(), key=lambda kv: -tim[kv[0]]):
    print(f"{c:3d} x {tim[k]:8.1f} us  {k[0]:22s} {k[1]:62s} {k[2]}"); tot += tim[k]
print("total aten device time per step: %.1f us in %d launches" % (tot, sum(rows.values())))
