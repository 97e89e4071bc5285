#!/bin/bash
O=gpurun_out/r5x; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "force_loss" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -8 | tee $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_padded.py tests/test_gpu_rangeflag.py tests/test_gpu_hbcheck.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "training or train" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a $O/ab.txt
for rep in 1 2; do for v in 1 0; do GEMNET_FUSED_LOSS=$v timeout 300 python bench.py --mode train --no-extras --no-cpu-baseline --no-roofline --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train fused_loss=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
