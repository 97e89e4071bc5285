"""Run one GEMM shape repeatedly (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemnet_pytorch_amd import kernels as K
M, N, Kd = (int(a) for a in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else -1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
A = torch.randn(M, Kd, device="cuda"); W = torch.randn(N, Kd, device="cuda")
for _ in range(reps):
    K.gemm(A, W, cfg=cfg)
torch.cuda.synchronize()
