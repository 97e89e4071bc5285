#!/bin/bash
# round 5, GPU call B: tangent kernels, GemNet-Q training parity on the fused twins, the captured Q training step
O=gpurun_out/r5b; mkdir -p $O
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_kernels.py -q -x -k "tangent or tensor_basis or quad_angles" > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/kernels.log | tee -a $O/summary.txt
python -m pytest tests/test_gpu_model.py -q -s -k "training_gradients_parity or layer_stacks or native" > $O/model.log 2>&1; echo "model tests rc=$?" | tee -a $O/summary.txt; grep -h "probe\|grad\|\[f32\]\|\[h3\]\|passed\|failed" $O/model.log | tail -25 | tee -a $O/summary.txt
python -m pytest tests/test_gpu_trainer.py -q -k "quad" > $O/trainer.log 2>&1; echo "trainer quad rc=$?" | tee -a $O/summary.txt; tail -3 $O/trainer.log | tee -a $O/summary.txt
timeout 600 python tools/exp/q_train_bench.py > $O/q_bench.json 2> $O/q_bench.err; echo "q bench rc=$?" | tee -a $O/summary.txt
grep -h "training step\|bil_\|quad_\|chain \|gemm " $O/q_bench.err | tail -40 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/q_bench.json").read().strip().splitlines()[-1]); t=d.get("train_step",{})
    print("Q fwd+force ms", d["ms_per_step"], "| train:", {k:t.get(k) for k in ("ms_per_step","molecules_per_s","hipgraph","peak_memory_gib","loss","error","launches_per_step")})
except Exception as e: print("q bench parse failed", e)
PY
GEMNET_TRAIN2_QUAD=0 timeout 600 python tools/exp/q_train_bench.py > $O/q_bench_r4form.json 2> $O/q_bench_r4form.err; echo "q bench (round-4 form) rc=$?" | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/q_bench_r4form.json").read().strip().splitlines()[-1]); t=d.get("train_step",{})
    print("round-4 form train:", {k:t.get(k) for k in ("ms_per_step","molecules_per_s","hipgraph","peak_memory_gib","loss","error")})
except Exception as e: print("q bench parse failed", e)
PY
