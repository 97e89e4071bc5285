#!/bin/bash
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
echo "== bench graph"; timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json; grep -v Warn $OUT/bench.log | tail -16
echo "== rocprof eager"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -30
find $OUT/prof -name "*.db" -delete
echo "== done"
