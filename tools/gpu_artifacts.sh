#!/bin/bash
# Round artifacts: bench lines, rocprofv3 kernel stats of the same command, PMC traffic of the dominant kernel.
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench T force"; timeout 900 python bench.py > $OUT/bench_T_force.json 2> $OUT/bench_T_force.log; cat $OUT/bench_T_force.json
echo "== bench T train"; timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_T_train.json 2> $OUT/bench_T_train.log; cut -c1-200 $OUT/bench_T_train.json
echo "== bench Q force"; timeout 900 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-200 $OUT/bench_Q_force.json
echo "== rocprof kernel stats (same command as the bench line, hipGraph replay)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
echo "== PMC traffic (separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1 )
done
python - <<PY
import glob, pandas as pd
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs: print("no csv for", c); continue
    df = pd.read_csv(fs[0])
    df["k"] = df["Kernel_Name"].str.replace(r"\(anonymous namespace\)::", "", regex=True).str.slice(0, 40)
    g = df[df["Counter_Name"] == c].groupby("k")["Counter_Value"].agg(["mean", "count"]).sort_values("mean", ascending=False)
    print(c, "(raw counter units per dispatch)"); print(g.head(12).to_string())
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete
echo "== done"
