#!/bin/bash
O=gpurun_out/r5k; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $R/$O/rocprof.log 2>&1 )
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) --list > $O/timeline.txt 2>&1; head -4 $O/timeline.txt
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
