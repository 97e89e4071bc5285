"""Per-program time of every chain launch of one GemNet-T (or, argument Q, GemNet-Q) forward+force step (GPU box), per kernel
layout of the "h3" arithmetic: tall = csrc/chain2.hip, wide = csrc/chain3.hip, row = csrc/chain4.hip (the default).
LAYOUTS=tall,row (default) selects the columns; the model is rebuilt per layout (the packed-weight format differs)."""
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
LAYOUTS = os.environ.get("LAYOUTS", "tall,row").split(",")
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
orig = K.chain


def sig_of(p):
    return (p.M, " ".join({"load": "L", "scale": "S", "gemm": "G", "store": "T"}[o["kind"]] +
                          (str(o["W"].shape[0]) + "x" + str(o["W"].shape[1]) if o["kind"] == "gemm" else "") for o in p.ops))


def uses_park(p):
    return any(o.get("slot") == 2 or any(isinstance(o.get(k), int) and o.get(k) == 2 for k in ("mul", "res", "res2")) for o in p.ops)


def is_adj(p):
    return uses_park(p) or any(o.get("y2", -1) >= 0 or o.get("out2") is not None or o.get("add") is not None
                               or o.get("add2") is not None for o in p.ops)


times, order, count, flags = {}, [], {}, {}
for layout in LAYOUTS:
    K.CHAIN_LAYOUT = layout
    torch.manual_seed(1234)
    model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
    model.requires_grad_(False)
    model(inputs); torch.cuda.synchronize()
    progs = []
    def rec(p, mode=None):
        progs.append(p)
        return orig(p, mode)
    K.chain = rec
    E, F = model(inputs); torch.cuda.synchronize()
    K.chain = orig
    print(f"[{layout}] E[0] = {float(E.flatten()[0]):.6f}  mean|F| = {float(F.abs().mean()):.6e}  launches {len(progs)}")
    groups = collections.OrderedDict()
    for p in progs:
        groups.setdefault(sig_of(p), []).append(p)
    for sig, ps in groups.items():
        if sig not in count:
            order.append(sig); count[sig] = len(ps); flags[sig] = ("P" if uses_park(ps[0]) else ("A" if is_adj(ps[0]) else "-"))
        times.setdefault(sig, {})[layout] = timeit(lambda: orig(ps[0], "h3"), iters=100)
    del model
tot = {l: 0.0 for l in LAYOUTS}
print(f"{'M':>6s} {'n':>3s} " + " ".join(f"{l:>9s}" for l in LAYOUTS) + "  kind  program   (kind: P parking slot, A second outputs / sources, - plain)")
for sig in order:
    for l in LAYOUTS:
        tot[l] += times[sig].get(l, float('nan')) * count[sig]
    print(f"{sig[0]:6d} {count[sig]:3d} " + " ".join(f"{times[sig].get(l, float('nan')):9.1f}" for l in LAYOUTS) + f"  {flags[sig]:>4s}  {sig[1]}")
print("total per step:", {k: round(v, 1) for k, v in tot.items()}, "us;", sum(count.values()), "launches")
