#!/bin/bash
TAG=${1:-s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench graph"; timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json; grep -v Warn $OUT/bench.log | tail -5
echo "== bench eager"; timeout 900 python bench.py --steps 20 --warmup 5 --no-graph > $OUT/bench_eager.json 2> $OUT/bench_eager.log; cat $OUT/bench_eager.json; grep -v Warn $OUT/bench_eager.log | tail -20
echo "== rocprof eager"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
find $OUT/prof -name "*.db" -delete
echo "== done"
