#!/bin/bash
# Kernel stats of the headline command per Dense-stack arithmetic (same box): bash tools/gpu_prof_modes.sh <tag> mode...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in "$@"; do
  ( cd /tmp && GEMNET_CHAIN_MODE=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$m -o trace -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof_$m.log 2>&1 )
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m"; grep -E "chain_split|pack_weight" $f | awk -F'","' '{printf "%-62s calls %6s  avg %8.1f us  total %8.2f ms\n", substr($1,2,62), $2, $4/1000, $3/1e6}'
  cp $f $OUT/kernel_stats_$m.csv
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
