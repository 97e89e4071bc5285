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
for i in range(3):
    E, F = model(inputs); torch.cuda.synchronize()
    print("eager", i, "dE", float((E - E0).abs().max()), "dF", float((F - F0).abs().max()), "nz", int((F != F0).sum()))
if "ext" in sys.argv:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        Es, Fs = model(inputs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("external stream run: dF", float((Fs - F0).abs().max()), "nz", int((Fs != F0).sum()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    Eg, Fg = model(inputs)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", i, "dE", float((Eg - E0).abs().max()), "dF", float((Fg - F0).abs().max()), "nz", int((Fg != F0).sum()), "scale", float(F0.abs().mean()))

E, F = model(inputs); torch.cuda.synchronize()
print("eager after capture: dF vs first", float((F - F0).abs().max()), "nz", int((F != F0).sum()), "vs graph nz", int((F != Fg).sum()))
