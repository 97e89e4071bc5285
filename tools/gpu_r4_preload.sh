#!/bin/bash
# kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=16 on every source) against the product library, same box
O=gpurun_out/r4_preload; mkdir -p $O
export PYTHONPATH=.:tests
L=$PWD/tools/exp/bin/libgemnet_hip_preload.so
GEMNET_HIP_LIB=$L timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
  timeout 600 python bench.py --no-config4 --no-extras --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_base_$i.log 2>&1; grep -h '^{"metric"' $O/bench_base_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  GEMNET_HIP_LIB=$L timeout 600 python bench.py --no-config4 --no-extras --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_pre_$i.log 2>&1; grep -h '^{"metric"' $O/bench_pre_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('preload', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
GEMNET_HIP_LIB=$L timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | cut -c1-260
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric"' | cut -c1-260
