"""Where does autograd itself add gradients in the forward+force path?  Walks the graph of E (eval mode, fused ops,
CPU emulation of the kernels) and lists every (node, input) that receives more than one gradient edge: each extra
edge is one elementwise add launched by the engine (41 per step at 4 blocks, profiles/r2_step_breakdown.txt)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpu_kernels
from test_oracle_model import load_case
from test_model_cpu import build

g = dict(np.load(os.path.join(ROOT, "tests/golden/model.npz"), allow_pickle=True))
tag = sys.argv[1] if len(sys.argv) > 1 else "t2"
cfg, params, inputs = load_case(g, tag)
real_grad = torch.autograd.grad
def spy(outputs, inputs_, *a, **k):
    outs = outputs if isinstance(outputs, (list, tuple)) else [outputs]
    fan = collections.Counter(); names = {}; seen = set(); stack = [o.grad_fn for o in outs if o.grad_fn is not None]
    producers = collections.defaultdict(list)
    while stack:
        n = stack.pop()
        if n in seen: continue
        seen.add(n)
        for (nx, nr) in n.next_functions:
            if nx is None: continue
            fan[(nx, nr)] += 1; producers[(nx, nr)].append(type(n).__name__)
            stack.append(nx)
    # the engine only runs nodes from which an input of the call is reachable
    targets = set()
    for t in (inputs_ if isinstance(inputs_, (list, tuple)) else [inputs_]):
        targets.add(t.grad_fn if t.grad_fn is not None else None)
    leaves = [n for n in seen if type(n).__name__ == "AccumulateGrad" and any(n.variable is t for t in (inputs_ if isinstance(inputs_, (list, tuple)) else [inputs_]))]
    needed = set(leaves) | {t for t in targets if t is not None}
    changed = True
    while changed:
        changed = False
        for n in seen:
            if n not in needed and any(nx in needed for nx, _ in n.next_functions if nx is not None):
                needed.add(n); changed = True
    fan = collections.Counter({k: 0 for k in fan}); prod2 = collections.defaultdict(list)
    for n in needed:
        for (nx, nr) in n.next_functions:
            if nx is not None and nx in needed:
                fan[(nx, nr)] += 1; prod2[(nx, nr)].append(type(n).__name__)
    producers = prod2
    tot = 0
    rows = collections.Counter()
    for (nx, nr), c in fan.items():
        if c > 1:
            tot += c - 1
            shape = None
            try: shape = tuple(nx._input_metadata[nr].shape)
            except Exception: pass
            rows[(type(nx).__name__, nr, shape, tuple(sorted(producers[(nx, nr)])))] += c - 1
    print(f"[{tag}] {len(seen)} nodes; {tot} engine-side adds per backward pass")
    for (name, nr, shape, prod), c in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"  {c:3d} x  grad of output {nr} of {name:28s} {shape}  <- {', '.join(prod)}")
    return real_grad(outputs, inputs_, *a, **k)
with cpu_kernels.emulate():
    model = build(cfg, params); model.eval()
    inputs["R"] = inputs["R"].double()
    torch.autograd.grad = spy
    model(inputs)
