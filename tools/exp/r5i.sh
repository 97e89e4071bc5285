#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests TMPDIR=/tmp
R=$PWD
timeout 300 python tools/exp/q_dyn_profile.py 12 2>&1 | grep -v Warn | tail -3 | tee $O/qdyn.txt
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/tools/exp/q_dyn_profile.py 12 > $R/$O/rocprof.log 2>&1 )
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
python - <<PY | tee -a $O/qdyn.txt
import glob, pandas as pd
f = glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)[0]
d = pd.read_csv(f); d["Name"] = d["Name"].str.replace("(anonymous namespace)::", "", regex=False).str.slice(0, 90)
print(d[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].head(45).to_string())
PY
