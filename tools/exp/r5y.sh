#!/bin/bash
O=gpurun_out/r5y; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "cbf_project" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -8 | tee $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_padded.py tests/test_gpu_hbcheck.py tests/test_gpu_index_in_graph.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee -a $O/ab.txt
for rep in 1 2; do for v in 1 0; do GEMNET_CBF_PROJECT=$v timeout 300 python bench.py --model Q --no-extras --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench Q cbf_project=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
