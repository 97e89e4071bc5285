#!/bin/bash
O=gpurun_out/r4_plan3; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
for i in 1 2; do
  timeout 300 python tools/exp/padded_ab.py 2>&1 | grep "ms" | sed "s/^/side-stream late plan: /" | tee -a $O/ab.txt
  GEMNET_PLAN_LATE=0 timeout 300 python tools/exp/padded_ab.py 2>&1 | grep "ms" | sed "s/^/in line: /" | tee -a $O/ab.txt
done
timeout 900 python bench.py --no-config4 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('bench', d['value'], d['ms_per_step'], 'padded', e['dynamic_shape']['padded_graph']['ms_per_step'], 'train', e['train_step']['ms_per_step'], 'train padded', e['train_step_dynamic']['padded_graph']['ms_per_step'])"
