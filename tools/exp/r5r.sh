#!/bin/bash
O=gpurun_out/r5r; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee $O/summary.txt
grep -E "passed|failed|error" $O/gpu_suite.log | tail -3 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | tee -a $O/summary.txt
