#!/bin/bash
O=gpurun_out/r4_dyang; mkdir -p $O
export PYTHONPATH=.:tests
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "angle_form or ang" > $O/tests.log 2>&1; tail -2 $O/tests.log
GEMNET_ANG_F16=7 timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "q4s or q2s or q1" -s 2>&1 | grep -i "force MAE\|passed\|failed" | tail -8
for m in 5 7 5 7; do
  GEMNET_ANG_F16=$m timeout 300 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/q_$m.json 2> $O/q_$m.log
  python -c "import json,sys; d=json.loads(open('$O/q_$m.json').read().strip().splitlines()[-1]); print('mask $m', d['value'], d['ms_per_step'])"; grep "bil_dy_multi\|bil_reduce_t \|bil_reduce_project" $O/q_$m.log | head -3
done
