#!/bin/bash
# Round-4 artifacts (GPU box): the driver-contract bench line, rocprofv3 kernel stats of the same command and of the
# training step / GemNet-Q, PMC traffic + SQ counters per workload (separate passes, no trace domains), same-box A/Bs,
# the configs[4] shard.      bash tools/gpu_artifacts3.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r4art}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench (default command of the driver)"; timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-300 $OUT/bench_default.json
echo "== bench train / Q"; timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_T_train.json 2> $OUT/bench_T_train.log; cut -c1-200 $OUT/bench_T_train.json
timeout 600 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-200 $OUT/bench_Q_force.json
echo "== rocprof kernel stats (same command as the bench line, hipGraph replay)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof.log 2>&1 )
python tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) --list > $OUT/timeline.txt 2>&1; head -4 $OUT/timeline.txt
echo "== rocprof kernel stats, training step / GemNet-Q"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_train -o trace -- python $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_train.log 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_Q -o trace -- python $R/bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof_Q.log 2>&1 )
echo "== PMC passes (separate, --pmc only)"
pmc() {  # mode, bench args...
  m=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${m}_$c -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_$c.log 2>&1 )
  done
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/pmc_${m}_sq -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_sq.log 2>&1 )
}
pmc T
pmc train --mode train
pmc Q --model Q
echo "== padded-capacity replay loop (tools/exp/padded_ab.py) + its kernel stats"
export PYTHONPATH=$R:$R/tests
timeout 300 python tools/exp/padded_ab.py > $OUT/padded_ab.txt 2>&1; grep "ms" $OUT/padded_ab.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_padded -o trace -- python $R/tools/exp/padded_ab.py 30 > $R/$OUT/rocprof_padded.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
echo "== done"
