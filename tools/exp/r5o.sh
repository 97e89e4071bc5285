#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python - <<PY 2>$O/dyn.err | tee $O/dyn.txt
import json, torch, bench
from gemnet_pytorch_amd.model.gemnet import GemNet
cfg = dict(bench.GEMNET_T)
torch.manual_seed(1234)
model = GemNet(**cfg, scale_file=bench.SCALE_FILE).to("cuda").eval(); model.requires_grad_(False)
for rep in range(2):
    d = bench.extra_dynamic_shape(cfg, model, 32, 32, 0, n_batches=4, steps=12, warmup=4)
    p = d.get("padded_graph", {})
    print("T dynamic:", d["ms_per_step"], "ms eager;", p.get("ms_per_step"), "ms padded;", p.get("index_in_graph", p.get("error")))
PY
tail -3 $O/dyn.err
