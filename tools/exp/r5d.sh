#!/bin/bash
# round 5, GPU call D: native CSR, GemNet-Q padded replay / MD, fused Q training inside TrainStep, glue inventory
O=gpurun_out/r5d; mkdir -p $O
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_index.py -q -x > $O/index.log 2>&1; echo "index + csr rc=$?" | tee -a $O/summary.txt; tail -2 $O/index.log | tee -a $O/summary.txt
python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py tests/test_gpu_rangeflag.py -q -x -s > $O/padded.log 2>&1; echo "padded + md + rangeflag rc=$?" | tee -a $O/summary.txt; grep -h "padded GemNet-Q\|passed\|failed\|Error" $O/padded.log | tail -6 | tee -a $O/summary.txt
python -m pytest tests/test_gpu_qtrain.py -q -x -s > $O/qtrain.log 2>&1; echo "qtrain rc=$?" | tee -a $O/summary.txt; grep -h "GemNet-Q\|passed\|failed\|Error\|assert" $O/qtrain.log | tail -8 | cut -c1-700 | tee -a $O/summary.txt
timeout 300 python tools/exp/md_bench.py 32 40 2>&1 | grep "GemNet" | tee -a $O/summary.txt
timeout 300 python tools/exp/md_bench.py 64 30 2>&1 | grep "GemNet" | tee -a $O/summary.txt
for v in 1 0; do echo "== padded T loop, GEMNET_NATIVE_CSR=$v" | tee -a $O/summary.txt; PYTHONPATH=$PWD:$PWD/tests GEMNET_NATIVE_CSR=$v timeout 300 python tools/exp/padded_ab.py 60 2>&1 | tail -4 | tee -a $O/summary.txt; done
for v in 1 0; do GEMNET_NATIVE_CSR=$v timeout 600 python - <<PY 2>$O/qdyn_$v.err | tee -a $O/summary.txt
import json, torch, bench
from gemnet_pytorch_amd.model.gemnet import GemNet
cfg = dict(bench.GEMNET_T, triplets_only=False)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to("cuda").eval(); model.requires_grad_(False)
d = bench.extra_dynamic_shape(cfg, model, 32, 32, 0, n_batches=3, steps=6, warmup=2)
print("Q dynamic, native csr $v:", d["ms_per_step"], "ms eager;", d.get("padded_graph", {}).get("ms_per_step"), "ms padded;", d.get("padded_graph", {}).get("error"))
PY
done
timeout 300 python tools/exp/train_glue.py > $O/train_glue.txt 2>&1; tail -45 $O/train_glue.txt | cut -c1-220 | tee -a $O/summary.txt
