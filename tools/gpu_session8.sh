#!/bin/bash
TAG=${1:-s15}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest chain"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "chain or stack" 2>&1 | tail -5
echo "== chain bench"; timeout 300 python tools/chain_bench.py > $OUT/chain_bench.txt 2>&1; cat $OUT/chain_bench.txt
echo "== bench stacks off"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> $OUT/b0.log | cut -c1-260
echo "== bench stacks on"; GEMNET_STACKS=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> $OUT/b1.log | cut -c1-260; grep -A14 "per-family" $OUT/b1.log
echo "== pytest model (stacks on)"; GEMNET_STACKS=1 timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== done"
