#!/bin/bash
# round 5, GPU call C: GemNet-Q padded replay + MD, the captured Q training step, re-run of the tests touched since call A
O=gpurun_out/r5c; mkdir -p $O
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py -q -x -s > $O/padded.log 2>&1; echo "padded + md rc=$?" | tee -a $O/summary.txt; grep -h "padded GemNet-Q\|passed\|failed\|Error" $O/padded.log | tail -8 | tee -a $O/summary.txt
python -m pytest tests/test_gpu_model.py -q -s -k "training_gradients_parity or layer_stacks or native" > $O/model.log 2>&1; echo "model tests rc=$?" | tee -a $O/summary.txt; grep -h "worst\|\[f32\]\|\[h3\]\|passed\|failed" $O/model.log | tail -30 | tee -a $O/summary.txt
timeout 900 python tools/exp/q_train_bench.py > $O/q_bench.json 2> $O/q_bench.err; echo "q bench rc=$?" | tee -a $O/summary.txt
grep -h -A 16 "training step: per-family" $O/q_bench.err | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/q_bench.json").read().strip().splitlines()[-1]); t=d.get("train_step",{})
    print("Q fwd+force ms", d["ms_per_step"], "| train:", {k:t.get(k) for k in ("ms_per_step","molecules_per_s","hipgraph","peak_memory_gib","loss","error","launches_per_step")})
    print("Q dynamic:", json.dumps(d.get("dynamic_shape"))[:1200])
except Exception as e: print("q bench parse failed", e)
PY
GEMNET_TRAIN2_QUAD=0 timeout 900 python tools/exp/q_train_bench.py > $O/q_bench_r4form.json 2> $O/q_bench_r4form.err; echo "q bench (round-4 form) rc=$?" | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/q_bench_r4form.json").read().strip().splitlines()[-1]); t=d.get("train_step",{})
    print("round-4 form train:", {k:t.get(k) for k in ("ms_per_step","molecules_per_s","hipgraph","peak_memory_gib","loss","error")})
except Exception as e: print("q bench parse failed", e)
PY
