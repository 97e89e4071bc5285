"""Where does the autograd engine itself add gradients in the forward+force pass of the headline model (GPU, fused ops)?
Walks the graph of E at the torch.autograd.grad call of GemNet.forward and lists every (node, input) with more than one
incoming gradient edge (each extra edge = one elementwise add launched by the engine).   PYTHONPATH=. python tools/exp/grad_fanin_gpu.py [T|Q]"""
import collections
import sys

import torch

import bench as B
from gemnet_pytorch_amd.model.gemnet import GemNet

which = sys.argv[1] if len(sys.argv) > 1 else "T"
dev = torch.device("cuda", 0)
cfg = dict(B.GEMNET_T, triplets_only=(which == "T"))
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=B.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = B.make_batch(cfg, 32, 32, first=0, device=dev)
real_grad = torch.autograd.grad


def spy(outputs, inputs_, *a, **k):
    outs = outputs if isinstance(outputs, (list, tuple)) else [outputs]
    ins = inputs_ if isinstance(inputs_, (list, tuple)) else [inputs_]
    seen, stack = set(), [o.grad_fn for o in outs if o.grad_fn is not None]
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        stack.extend(nx for nx, _ in n.next_functions if nx is not None)
    needed = {n for n in seen if type(n).__name__ == "AccumulateGrad" and any(n.variable is t for t in ins)}
    needed |= {t.grad_fn for t in ins if t.grad_fn is not None}
    changed = True
    while changed:
        changed = False
        for n in seen:
            if n not in needed and any(nx in needed for nx, _ in n.next_functions if nx is not None):
                needed.add(n)
                changed = True
    fan, prod = collections.Counter(), collections.defaultdict(list)
    for n in needed:
        for nx, nr in n.next_functions:
            if nx is not None and nx in needed:
                fan[(nx, nr)] += 1
                prod[(nx, nr)].append(type(n).__name__)
    rows = collections.Counter()
    for (nx, nr), c in fan.items():
        if c > 1:
            try:
                shape = tuple(nx._input_metadata[nr].shape)
            except Exception:  # noqa: BLE001
                shape = None
            rows[(type(nx).__name__, nr, shape, tuple(sorted(prod[(nx, nr)])))] += c - 1
    print(f"GemNet-{which}: {len(needed)} nodes run; {sum(rows.values())} engine-side adds per backward pass")
    for (name, nr, shape, pr), c in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"  {c:3d} x  grad of output {nr} of {name:26s} {shape}  <- {', '.join(pr)}")
    return real_grad(outputs, inputs_, *a, **k)


torch.autograd.grad = spy
model(inputs)
