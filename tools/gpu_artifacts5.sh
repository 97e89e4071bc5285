#!/bin/bash
# Round-5 artifacts (GPU box): rocprofv3 kernel stats of the bench command / the training step / GemNet-Q / GemNet-Q training,
# PMC traffic + SQ counters per workload (separate passes, no trace domains) summarised into profiles/r5_* ON the box, THEN
# the driver-contract bench lines (which read those summaries).      bash tools/gpu_artifacts5.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r5art}
OUT=gpurun_out/$TAG
mkdir -p $OUT/profiles
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R:$R/tests
echo "== rocprof kernel stats (same command as the bench line, hipGraph replay)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof.log 2>&1 )
python tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) --list > $OUT/timeline.txt 2>&1; head -4 $OUT/timeline.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_train -o trace -- python $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_train.log 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_Q -o trace -- python $R/bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof_Q.log 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_Qtrain -o trace -- python $R/bench.py --model Q --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_Qtrain.log 2>&1 )
echo "== PMC passes (separate, --pmc only)"
pmc() {  # mode, bench args...
  m=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${m}_$c -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_$c.log 2>&1 )
  done
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/pmc_${m}_sq -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_sq.log 2>&1 )
  python tools/pmc_summary.py $OUT $m r5 2>&1 | tail -2
}
pmc T
pmc train --mode train
pmc Q --model Q
pmc Qtrain --model Q --mode train
for f in prof prof_train prof_Q prof_Qtrain; do cp $(find $OUT/$f -name "*kernel_stats.csv" | head -1) $OUT/profiles/r5_${f}_kernel_stats.csv 2>/dev/null; done
cp profiles/r5_pmc_* profiles/r5_traffic_* profiles/r5_mfma_busy.json $OUT/profiles/ 2>/dev/null
cp $OUT/timeline.txt $OUT/profiles/r5_timeline.txt
echo "== bench (default command of the driver; reads profiles/r5_traffic_* written above)"
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-300 $OUT/bench_default.json
echo "== bench train / Q"
timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_T_train.json 2> $OUT/bench_T_train.log; cut -c1-200 $OUT/bench_T_train.json
timeout 600 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-200 $OUT/bench_Q_force.json
timeout 600 python bench.py --model Q --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_Q_train.json 2> $OUT/bench_Q_train.log; cut -c1-200 $OUT/bench_Q_train.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
echo "== done"
