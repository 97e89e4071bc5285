"""forward+force of the headline batch as S concurrent sub-batch hipGraphs (gemnet_pytorch_amd.runtime)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.runtime import ForceGraphs

cfg = dict(bench.GEMNET_T)
if len(sys.argv) > 1 and sys.argv[1] == "Q":
    cfg["triplets_only"] = False
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
B, A = 32, 32
full, _ = bench.make_batch(cfg, B, A, first=0, device=dev)
E0, F0 = model(full)
torch.cuda.synchronize()
for S in (1, 2, 4, 8):
    n = B // S
    batches = [bench.make_batch(cfg, n, A, first=i * n, device=dev)[0] for i in range(S)]
    try:
        run = ForceGraphs(model, batches)
        run(); torch.cuda.synchronize()
        E, F = run.energies_forces()
        err = (float((E - E0).abs().max()), float((F - F0).abs().max()))
        for _ in range(10): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"S={S}: {dt*1e3:.3f} ms/step  {B/dt:.0f} mol/s  max|dE|={err[0]:.2e} max|dF|={err[1]:.2e}", flush=True)
    except Exception as ex:
        print(f"S={S}: failed {type(ex).__name__}: {ex}", flush=True)
    del run
