#!/bin/bash
# Same-box A/B of switches on the headline (forward+force, hipGraph): tools/gpu_ab.sh <out> VAR=a VAR=b ...
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for rep in 1 2; do
for kv in "$@"; do
  v=$(env $kv python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$kv  ms_per_step molecules/s: $v" | tee -a $OUT
done
done
