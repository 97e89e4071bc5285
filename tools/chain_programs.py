"""Per-program time of every chain launch of one GemNet-T forward+force step (GPU box): which stacks / adjoints cost what,
on the f32-MFMA kernel and on the split-operand bf16 kernel."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd import kernels as K
from tools.gemm_bench import timeit
cfg = dict(bench.GEMNET_T)
if "Q" in sys.argv:
    cfg["triplets_only"] = False
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
model(inputs); torch.cuda.synchronize()
progs = []
orig = K.chain
def rec(p, mode=None):
    progs.append(p)
    return orig(p, mode)
K.chain = rec
model(inputs); torch.cuda.synchronize()
K.chain = orig
groups = collections.OrderedDict()
for p in progs:
    sig = (p.M, " ".join({"load": "L", "scale": "S", "gemm": "G", "store": "T"}[o["kind"]] + (str(o["W"].shape[0]) + "x" + str(o["W"].shape[1]) if o["kind"] == "gemm" else "") for o in p.ops))
    groups.setdefault(sig, []).append(p)
tot = {"f32": 0.0, "split6": 0.0}
print(f"{'M':>6s} {'n':>3s} {'f32 us':>8s} {'split6 us':>9s}  program")
for sig, ps in groups.items():
    t = {}
    for mode in tot:
        t[mode] = timeit(lambda: orig(ps[0], mode), iters=100)
        tot[mode] += t[mode] * len(ps)
    print(f"{sig[0]:6d} {len(ps):3d} {t['f32']:8.1f} {t['split6']:9.1f}  {sig[1]}")
print("total per step:", {k: round(v, 1) for k, v in tot.items()}, "us;", len(progs), "launches")
