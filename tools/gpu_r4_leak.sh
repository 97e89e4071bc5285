#!/bin/bash
O=gpurun_out/r4_leak; mkdir -p $O
export PYTHONPATH=.:tests
timeout 200 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_padded.py tests/test_gpu_model.py -x -q -m gpu -k "train or Train or captured" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 100 python - <<'PY' 2>&1 | grep -v amdgpu | tail -4
import time, torch, bench
from gemnet_pytorch_amd.model.gemnet import GemNet
from gemnet_pytorch_amd.training.ddp import TrainStep
cfg = dict(bench.GEMNET_T); dev = torch.device("cuda")
torch.manual_seed(0)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to(dev)
inputs, targets = bench.make_batch(cfg, 32, 32, first=0, device=dev)
ts = TrainStep(model, fused_optimizer=True)
for _ in range(3): ts(inputs, targets)
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for i in range(10): ts(inputs, targets)
torch.cuda.synchronize()
print(f"eager training step: {(time.perf_counter()-t0)/10*1e3:.1f} ms; allocated before/after 10 steps {m0/2**30:.2f} / {torch.cuda.memory_allocated()/2**30:.2f} GiB, peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
PY
