#!/bin/bash
O=gpurun_out/r5z2; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
for rep in 1 2 3; do for v in default k1shfl; do L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so; GEMNET_HIP_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
