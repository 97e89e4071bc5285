#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -2 | tee $O/ab.txt
python -m pytest tests/test_gpu_model.py -q -k "energy_force_parity or hipgraph or replay" 2>&1 | tail -2 | tee -a $O/ab.txt
for rep in 1 2 3; do for v in default widek; do L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so; GEMNET_HIP_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
for v in default widek; do L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so; GEMNET_HIP_LIB=$L timeout 300 python bench.py --model Q --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench Q $v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
