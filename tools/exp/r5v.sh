#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee $O/summary.txt
grep -E "passed|failed|error" $O/gpu_suite.log | tail -3 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | tee -a $O/summary.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.log; cut -c1-300 $O/bench_default.json | tee -a $O/summary.txt
timeout 600 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_Q_force.json 2> $O/bench_Q_force.log; cut -c1-250 $O/bench_Q_force.json | tee -a $O/summary.txt
timeout 600 python bench.py --model Q --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_Q_train.json 2> $O/bench_Q_train.log; cut -c1-250 $O/bench_Q_train.json | tee -a $O/summary.txt
timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_T_train.json 2> $O/bench_T_train.log; cut -c1-250 $O/bench_T_train.json | tee -a $O/summary.txt
