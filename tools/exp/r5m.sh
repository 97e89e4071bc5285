#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 200 python tools/exp/grad_fanin_gpu.py T 2>&1 | grep -v Warn | tail -14 | cut -c1-250 | tee $O/fanin.txt
for rep in 1 2 3 4; do for v in 1 0; do GEMNET_LATE_DY=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench late_dy=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done; done
