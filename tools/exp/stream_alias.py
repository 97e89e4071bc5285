"""The stream pool of torch is round-robin over 32 streams: force the model's side stream (created inside a capture)
onto the slot of the capture stream itself and check that graph replay == eager, bit for bit (GemNet._side_stream
skips an aliasing stream; before that fix the output blocks joined the gradient sinks of the main stream in the captured
run only, which changed the summation order)."""
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
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):          # creates the process-wide capture stream
    torch.zeros(1, device="cuda")
C = torch.cuda.graph.default_capture_stream
n = 0
while torch.cuda.Stream().cuda_stream != C.cuda_stream and n < 64:
    n += 1
for _ in range(31):
    torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    side = model._side_stream(torch.device("cuda", 0))
    print("next pool stream would have been the capture stream; model side stream aliases it:",
          side.cuda_stream == torch.cuda.current_stream().cuda_stream)
    Eg, Fg = model(inputs)
g.replay(); torch.cuda.synchronize()
print(f"{kind} {n_mol}x{n_atoms}: graph vs eager max|dF| = {float((Fg - F0).abs().max()):.3e} ({int((Fg != F0).sum())} elements differ)")
