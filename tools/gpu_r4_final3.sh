#!/bin/bash
TAG=r4final4; OUT=gpurun_out/$TAG; mkdir -p $OUT
PYTHONPATH=.:tests timeout 900 python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py tests/test_gpu_hbcheck.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_T_train.json 2> $OUT/bench_T_train.log; cut -c1-200 $OUT/bench_T_train.json
