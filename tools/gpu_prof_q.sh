#!/bin/bash
OUT=gpurun_out/${1:-pq}; mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --model Q --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
python - <<PY
import pandas as pd, glob
f = glob.glob('$OUT/prof/*kernel_stats.csv')[0]
df = pd.read_csv(f)
n = 5 + 2 + 2 + 3  # timed + warmup + eager warm + capture warm-ups
df['ms_per_step'] = df['TotalDurationNs'] / n / 1e6
df['calls_per_step'] = df['Calls'] / n
df['name'] = df['Name'].str.replace(r'\(anonymous namespace\)::','',regex=True).str.replace('void ','').str.replace('at::native::','').str.slice(0,70)
print(df[['name','calls_per_step','ms_per_step','AverageNs']].head(24).round(2).to_string())
print('total', df['ms_per_step'].sum(), df['calls_per_step'].sum())
PY
tail -2 $OUT/rocprof.log | cut -c1-200
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -size +30M -delete
