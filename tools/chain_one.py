"""One chain program (5 GEMMs 128x128, optional act/pre) launched `reps` times: target for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemnet_pytorch_amd import kernels as K
M = int(sys.argv[1]); n = int(sys.argv[2]); pre = int(sys.argv[3]); reps = int(sys.argv[4])
x = torch.randn(M, 128, device="cuda")
Ws = [torch.randn(128, 128, device="cuda") / 11 for _ in range(n)]
zs = [torch.empty(M, 128, device="cuda") for _ in range(n)]
y = torch.empty(M, 128, device="cuda")
p = K.ChainProgram(M); p.load(0, x)
cur, oth = 0, 1
for i in range(n):
    p.gemm(Ws[i], a_slot=cur, y_slot=oth, act=bool(pre), pre_out=zs[i] if pre else None, out=y if i == n - 1 else None)
    cur, oth = oth, cur
for _ in range(reps):
    K.chain(p)
torch.cuda.synchronize()
