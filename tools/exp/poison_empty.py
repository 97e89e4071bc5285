"""Read-before-write detector (GPU, eager): every torch.empty / empty_like buffer the launchers allocate is filled with
NaN first; a kernel that reads a buffer it (or its producer) did not fully write shows up as NaN in E / F.
    python tools/exp/poison_empty.py Q 8 64"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fullsize as T
from gemnet_pytorch_amd.model.gemnet import GemNet
kind, n_mol, n_atoms = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = dict(T.FULL, triplets_only=kind == "T")
torch.manual_seed(11)
model = GemNet(**cfg, scale_file=T.SCALE_FILE).to("cuda").eval(); model.requires_grad_(False)
ds = T.make_dataset(n_mol, n_atoms, config=2)
inputs = T.batch_of(ds, range(n_mol), cfg["triplets_only"])
E0, F0 = model(inputs); torch.cuda.synchronize()
real_empty, real_like = torch.empty, torch.empty_like
def poison(*a, **k):
    t = real_empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
def poison_like(x, *a, **k):
    t = real_like(x, *a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = poison, poison_like
try:
    E, F = model(inputs); torch.cuda.synchronize()
finally:
    torch.empty, torch.empty_like = real_empty, real_like
print(f"{kind} {n_mol}x{n_atoms}: NaN in E {int(torch.isnan(E).sum())}/{E.numel()}, in F {int(torch.isnan(F).sum())}/{F.numel()}; "
      f"max|dF| vs clean run {float((F - F0).abs().nan_to_num(0).max()):.3e}")
