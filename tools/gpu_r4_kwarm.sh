#!/bin/bash
# chain kernel prologue A/B (kernarg warm-up, op table without its own barrier): tests, phase trace, headline bench against
# the library built before the change (tools/exp/bin/libgemnet_hip_nokwarm.so)
O=gpurun_out/r4_kw2; mkdir -p $O
export PYTHONPATH=.:tests
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chain or stack or program" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/chain2_trace.py --quick --modes=h3 > "$O/trace_edge.txt" 2>&1
timeout 300 python tools/chain2_trace.py --quick --small --modes=h3 > "$O/trace_atom.txt" 2>&1
grep -h "prologue" $O/trace_*.txt
for i in 1 2; do
  timeout 600 python bench.py --no-config4 --steps 60 --warmup 10 > $O/bench_kw_$i.log 2>&1; grep -h '^{"metric"' $O/bench_kw_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kwarm', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  GEMNET_HIP_LIB=$PWD/tools/exp/bin/libgemnet_hip_nokwarm.so timeout 600 python bench.py --no-config4 --steps 60 --warmup 10 > $O/bench_nokw_$i.log 2>&1; grep -h '^{"metric"' $O/bench_nokw_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no kwarm', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_hbcheck.py -x -q -m gpu > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
