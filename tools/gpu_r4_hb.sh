#!/bin/bash
# Round-4 GPU session A: the happens-before checker over the captured steps (shipped configurations and the three
# round-3 findings), then the driver's bench line as the round's starting point.   outputs under gpurun_out/<tag>/
TAG=${1:-r4_hb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== build"; timeout 900 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
echo "== hbcheck"; PYTHONPATH=.:tests timeout 1500 python tools/hbcheck_run.py > $OUT/hbcheck.txt 2> $OUT/hbcheck.err; grep -E "^===|^summary" $OUT/hbcheck.txt; tail -3 $OUT/hbcheck.err
echo "== bench"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.log; cut -c1-400 $OUT/bench.json
echo "== done"
