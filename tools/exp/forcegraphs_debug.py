import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for v in ("one", "two", "two_noeager"):
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, "->", (r.stdout.strip().splitlines() or ["<no output>"])[-1], "| rc", r.returncode)
    sys.exit(0)
import torch
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.runtime import ForceGraphs
v = sys.argv[1]
cfg = dict(bench.GEMNET_T)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
if v == "two_nooverlap":
    model.overlap_output_blocks = False
a, _ = bench.make_batch(cfg, 4, 32, first=0, device=dev)
b, _ = bench.make_batch(cfg, 4, 32, first=4, device=dev)
if v not in ("two_noeager",):
    E0, F0 = model(dict(a))
    torch.cuda.synchronize()
if v == "two_sameStream":
    import gemnet_pytorch_amd.runtime as rt
    s = torch.cuda.Stream(device=dev)
    orig = torch.cuda.Stream
    rt.torch.cuda.Stream = lambda device=None: s
runner = ForceGraphs(model, [a] if v == "one" else [a, b])
runner()
torch.cuda.synchronize()
E, F = runner.energies_forces()
print("ok", v, float(F.abs().mean()))
