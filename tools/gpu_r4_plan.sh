#!/bin/bash
O=gpurun_out/r4_plan; mkdir -p $O
export PYTHONPATH=.:tests
timeout 900 python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  GEMNET_PLAN_LATE=0 timeout 300 python tools/exp/padded_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  GEMNET_PLAN_LATE=1 timeout 300 python tools/exp/padded_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests; cd /tmp && export TMPDIR=/tmp && GEMNET_PLAN_LATE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python $GRAFT_REPO_ROOT/tools/exp/padded_ab.py 30 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
