#!/bin/bash
TAG=${1:-s19}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest all gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== gemm bench"; timeout 300 python tools/gemm_bench.py 2>&1 | grep -A8 "shape (18122, 128, 128)" | head -12
echo "== bench"; timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json; grep -A14 "per-family" $OUT/bench.log
echo "== bench train"; timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.log; cut -c1-300 $OUT/train.json; grep -A12 "per-family" $OUT/train.log
echo "== done"
