#!/bin/bash
# Same-box A/B of switches on the training step: tools/gpu_ab_train.sh <out> VAR=a VAR=b ...
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for rep in 1 2; do
for kv in "$@"; do
  v=$(env $kv python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$kv  train ms_per_step molecules/s: $v" | tee -a $OUT
done
done
