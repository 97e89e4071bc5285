#!/bin/bash
# headline / training step / GemNet-Q under DEBUG_HIP_FORCE_GRAPH_QUEUES (hardware queues a replayed hipGraph may use), one box
O=gpurun_out/r4_envperf; mkdir -p $O
run() { # label, bench args..., then env after --
  l=$1; shift
  timeout 300 python bench.py "$@" --no-config4 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'])" | tee -a $O/env2.txt
}
for i in 1 2; do
  run "T base   " --steps 80 --warmup 10
  DEBUG_HIP_FORCE_GRAPH_QUEUES=2 run "T queues=2" --steps 80 --warmup 10
done
run "train base   " --mode train --steps 20 --warmup 5
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 run "train queues=2" --mode train --steps 20 --warmup 5
run "Q base   " --model Q --steps 10 --warmup 3
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 run "Q queues=2" --model Q --steps 10 --warmup 3
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 PYTHONPATH=.:tests timeout 600 python -m pytest tests/test_gpu_hbcheck.py tests/test_gpu_padded.py -x -q -m gpu 2>&1 | tail -1
