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
# columns: (label, chain mode, kernel layout, forced tile rows of the wide layout (0 = automatic))
from gemnet_pytorch_amd import _lib
lib = _lib.load()
COLS = [("h3 tall", "h3", "tall", 0), ("h3 wide", "h3", "wide", 0)]
COLS += [(f"wide/{r}", "h3", "wide", int(r)) for r in os.environ.get("TILE_ROWS", "").split(",") if r]
# (label "stg N": the wide layout with the second workgroup of a CU started N x 64 cycles late; negative "rows" encodes it)
COLS += [(f"stg {n}", "h3", "wide", -int(n)) for n in os.environ.get("STAGGER", "").split(",") if n]
tot = {c[0]: 0.0 for c in COLS}
def uses_park(p):
    return any(o.get("slot") == 2 or any(isinstance(o.get(k), int) and o.get(k) == 2 for k in ("mul", "res", "res2")) for o in p.ops)
print(f"{'M':>6s} {'n':>3s} " + " ".join(f"{c[0]:>9s}" for c in COLS) + "  park  program")
for sig, ps in groups.items():
    t = {}
    for label, mode, layout, rows in COLS:
        K.CHAIN_LAYOUT = layout
        K.WIDE_TILE_ROWS, K.WIDE_STAGGER = max(rows, 0), max(-rows, 0)      # bits of this launch's `nprod` (ABI 13)
        t[label] = timeit(lambda: orig(ps[0], mode), iters=100)
        tot[label] += t[label] * len(ps)
    print(f"{sig[0]:6d} {len(ps):3d} " + " ".join(f"{t[c[0]]:9.1f}" for c in COLS) + f"  {'P' if uses_park(ps[0]) else '-':>4s}  {sig[1]}")
K.WIDE_TILE_ROWS = K.WIDE_STAGGER = 0
print("total per step:", {k: round(v, 1) for k, v in tot.items()}, "us;", len(progs), "launches")
