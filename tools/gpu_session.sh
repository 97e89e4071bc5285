#!/bin/bash
# One GPU-box session: GPU tests, smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
# usage: tools/gpu_session.sh [tag]
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; rocm-smi --showproductname 2>&1 | head -20 >> $OUT/env.log
nproc >> $OUT/env.log; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/env.log
echo "== build"; timeout 600 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json; tail -25 $OUT/bench.log
echo "== rocprof"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
ls -R $OUT/prof | head -20
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
# keep the merged-back output small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"
