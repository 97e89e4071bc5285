#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
for v in default aggv1; do L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so; echo "== $v"; GEMNET_HIP_LIB=$L timeout 200 python tools/exp/basis_bench.py 2>&1 | grep -v Warn; done | tee $O/basis.txt
python -m pytest tests/test_gpu_kernels.py -q -k "basis or bessel or radial or ylm" 2>&1 | tail -2 | tee -a $O/basis.txt
for rep in 1 2; do GEMNET_HIP_LIB= timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench default', d['value'], d['ms_per_step'])" | tee -a $O/basis.txt; done
