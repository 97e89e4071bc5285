#!/bin/bash
TAG=${1:-s20}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest all gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
echo "== bench train"; timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.log; cut -c1-300 $OUT/train.json; grep -A12 "per-family" $OUT/train.log
echo "== bench force"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
echo "== done"
