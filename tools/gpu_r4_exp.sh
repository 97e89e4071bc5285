#!/bin/bash
# same-box A/B of the product library against tools/exp/bin/libgemnet_hip_prev.so (the library built before a change)
O=gpurun_out/r4_acc3; mkdir -p $O
export PYTHONPATH=.:tests
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chain or stack or program" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python tools/chain2_trace.py --quick --modes=h3 2>&1 | grep -v amdgpu.ids > "$O/trace_edge.txt"; grep "wave0 op[1-3]\|wave7 op[1-3]" "$O/trace_edge.txt" | head -6
for i in 1 2; do
  timeout 600 python bench.py --no-config4 --no-extras --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_new_$i.log 2>&1; grep -h '^{"metric"' $O/bench_new_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  GEMNET_HIP_LIB=$PWD/tools/exp/bin/libgemnet_hip_prev.so timeout 600 python bench.py --no-config4 --no-extras --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_prev_$i.log 2>&1; grep -h '^{"metric"' $O/bench_prev_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "t4s or parity" -s 2>&1 | grep -i "t4s.*force MAE\|passed\|failed" | tail -5
