#!/bin/bash
O=gpurun_out/r5ad; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee $O/ab.txt
for rep in 1 2; do for v in default nopre; do L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so; GEMNET_HIP_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
