#!/bin/bash
O=gpurun_out/r5ac; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_index.py tests/test_gpu_padded.py tests/test_gpu_index_in_graph.py tests/test_gpu_md.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/ab.txt
for rep in 1 2; do for v in 1 0; do
GEMNET_NATIVE_EXPANDED=$v timeout 600 python - <<PY 2>/dev/null | tee -a $O/ab.txt
import json, torch, bench
from gemnet_pytorch_amd.model.gemnet import GemNet
cfg = dict(bench.GEMNET_T)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to("cuda").eval(); model.requires_grad_(False)
d = bench.extra_dynamic_shape(cfg, model, 32, 32, 0, n_batches=3, steps=12, warmup=3)
p = d.get("padded_graph", {})
print("native_expanded=$v T dynamic:", p.get("ms_per_step"), "ms padded;", p.get("index_in_graph", {}).get("ms_per_step"), "in-graph index", p.get("error"))
PY
done; done
for v in 1 0; do GEMNET_NATIVE_EXPANDED=$v timeout 300 python tools/exp/md_bench.py 48 40 2>&1 | grep "GemNet-T" | cut -c1-60 | sed "s/^/native_expanded=$v /" | tee -a $O/ab.txt; done
