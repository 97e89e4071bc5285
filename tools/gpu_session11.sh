#!/bin/bash
TAG=${1:-s18}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest all gpu (stacks off)"; timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== pytest model (stacks on)"; GEMNET_STACKS=1 timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x --tb=short -p no:cacheprovider -s 2>&1 | grep "force MAE\|passed\|failed" | tail -8
echo "== chain bench"; timeout 300 python tools/chain_bench.py 2>&1 | grep -A12 "^M=18122"
echo "== bench stacks off"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> $OUT/b0.log | cut -c1-260
echo "== bench stacks on"; GEMNET_STACKS=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> $OUT/b1.log | cut -c1-260; grep -A8 "per-family" $OUT/b1.log
echo "== done"
