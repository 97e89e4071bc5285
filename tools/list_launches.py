"""List the launches of one forward+force step by family/shape (GPU box)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd import kernels as K
cfg = dict(bench.GEMNET_T)
if "Q" in sys.argv:
    cfg["triplets_only"] = False
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev).eval()
model.requires_grad_(False)
inputs, _ = bench.make_batch(cfg, 32, 32, first=0, device=dev)
model(inputs); torch.cuda.synchronize()
cnt = collections.Counter()
orig = K.gemm
def logged(A, B, trans_a=False, trans_b=False, **kw):
    M, Kd = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    N = B.shape[1] if trans_b else B.shape[0]
    flags = "".join(c for c, k in (("d", "a_dact_pre"), ("m", "mul"), ("r", "res"), ("2", "res2"), ("g", "gadd1")) if kw.get(k) is not None)
    fast = (not trans_a and not trans_b and A.stride(0) % 4 == 0 and B.stride(0) % 4 == 0 and Kd % 4 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)
    cnt[("gemm", int(trans_a), int(trans_b), M, N, Kd, flags + ("A" if kw.get("act") else ""), "fast" if fast else "GENERIC",
         A.stride(0), B.stride(0))] += 1
    return orig(A, B, trans_a, trans_b, **kw)
K.gemm = logged
origc = K.chain
def loggedc(p):
    cnt[("chain", p.M, len(p.ops), sum(o["kind"] == "gemm" for o in p.ops))] += 1
    return origc(p)
K.chain = loggedc
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    model(inputs); torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
