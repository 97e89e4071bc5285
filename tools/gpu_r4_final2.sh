#!/bin/bash
# closing session: the whole -m gpu suite, smoke(), the default bench line, kernel stats of the headline command
TAG=r4final2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof.log 2>&1 )
python tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) --list > $OUT/timeline.txt 2>&1; head -4 $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== done"
