#!/bin/bash
# Round-3 GPU-box session: GPU tests, smoke, the driver's bench line, rocprofv3 kernel stats of the TRAINING step.
# usage: tools/gpu_session3.sh <tag> [pytest-args...]        outputs under gpurun_out/<tag>/
TAG=${1:-r3}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== build"; timeout 600 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
echo "== pytest -m gpu $*"; timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider "$@" > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo "== bench"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.log; cut -c1-600 $OUT/bench.json; grep -E "extra\.|train" $OUT/bench.log | cut -c1-700 | head -12
echo "== rocprof training step"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1 )
f=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -22 "$f" | cut -c1-180
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== done"
