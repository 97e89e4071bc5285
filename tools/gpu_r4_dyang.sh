#!/bin/bash
O=gpurun_out/r4_exp; mkdir -p $O
export PYTHONPATH=.:tests
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "angle_form or ang" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "q4s or q2s or q1" -s 2>&1 | grep -i "force MAE\|passed\|failed" | tail -4
for i in 1 2; do
  for L in "" $PWD/tools/exp/bin/libgemnet_hip_prev.so; do
  GEMNET_HIP_LIB=$L timeout 300 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/q.json 2> $O/q.log
  python -c "import json,sys; d=json.loads(open('$O/q.json').read().strip().splitlines()[-1]); print('lib [$L]', d['value'], d['ms_per_step'])"; grep "bil_reduce_t " $O/q.log | head -1
  done
done
