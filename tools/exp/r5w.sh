#!/bin/bash
O=gpurun_out/r5w; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "ang or expand or bil" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "q4s or q2s or q1" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a $O/ab.txt
for rep in 1 2; do for v in 1 0; do GEMNET_EXPAND_POS=$v timeout 300 python bench.py --model Q --no-extras --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench Q expand_pos=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
