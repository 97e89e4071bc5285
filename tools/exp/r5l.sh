#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
python -m pytest tests/test_gpu_hbcheck.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
python -m pytest tests/test_gpu_model.py -q -x -k "parity or replay or graph" 2>&1 | tail -3 | tee -a $O/tests.txt
for rep in 1 2; do for v in 1 0; do GEMNET_LATE_DY=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench late_dy=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
for v in 1 0; do GEMNET_LATE_DY=$v timeout 300 python bench.py --model Q --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench Q late_dy=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
timeout 200 python tools/exp/aten_ops.py > $O/aten_ops.txt 2>&1; tail -30 $O/aten_ops.txt | cut -c1-200
