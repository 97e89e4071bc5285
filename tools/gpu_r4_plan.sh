#!/bin/bash
O=gpurun_out/r4_plan2; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
timeout 900 python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py tests/test_gpu_hbcheck.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for q in "" 2; do
  DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 300 python tools/exp/padded_ab.py 2>&1 | grep "ms" | sed "s/^/queues=[$q] /" | tee -a $O/ab.txt
done
timeout 900 python bench.py --no-config4 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('bench', d['value'], d['ms_per_step'], 'padded', e['dynamic_shape']['padded_graph']['ms_per_step'], 'train', e['train_step']['ms_per_step'], 'train padded', e['train_step_dynamic']['padded_graph']['ms_per_step'])"
