#!/bin/bash
# The round's evidence on the GPU box (one parameterised script since round 6; it replaces gpu_artifacts{,3,4,5}.sh, the
# gpu_r4_*.sh sessions and the 31 one-off tools/exp/r5*.sh):
#     bash tools/gpu_artifacts.sh <round, e.g. r6> [tag] [parts: all | stats,pmc,bench,config4]
#   stats   rocprofv3 --kernel-trace --stats of the bench command (hipGraph replay), the training step, GemNet-Q, GemNet-Q training
#   pmc     separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters; no trace domains) per workload, summarised by
#           tools/pmc_summary.py into profiles/<round>_traffic_<mode>.json (+ the ALGORITHMIC bytes per launch of every launcher
#           family, dumped by bench.py under GEMNET_DUMP_FAMILIES), <round>_pmc_*.txt, <round>_mfma_busy.json  ON the box
#   config4 the same for ONE GPU's shard of BASELINE configs[4] (64 x 64-atom GemNet-Q)
#   bench   then the driver-contract bench lines (which read those summaries)
# Everything lands in gpurun_out/<tag>/ (profiles/ copies under gpurun_out/<tag>/profiles/: copy what is to be judged into profiles/).
RND=${1:?round (r6)}
TAG=${2:-${RND}art}
PARTS=${3:-all}
has() { [[ "$PARTS" == all || ",$PARTS," == *",$1,"* ]]; }
OUT=gpurun_out/$TAG
mkdir -p $OUT/profiles
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R:$R/tests
B="--no-cpu-baseline --no-roofline"
stats() {  # name, bench args...
  n=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$n -o trace -- python $R/bench.py "$@" $B > $R/$OUT/rocprof_$n.log 2>&1 )
  cp $(find $OUT/$n -name "*kernel_stats.csv" | head -1) $OUT/profiles/${RND}_${n}_kernel_stats.csv 2>/dev/null
}
if has stats; then
  echo "== rocprof kernel stats"
  stats prof --steps 50 --warmup 10 --no-extras
  python tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) --list > $OUT/profiles/${RND}_timeline.txt 2>&1; head -4 $OUT/profiles/${RND}_timeline.txt
  stats prof_train --mode train --steps 20 --warmup 5
  stats prof_Q --model Q --steps 10 --warmup 3 --no-extras
  stats prof_Qtrain --model Q --mode train --steps 5 --warmup 2
fi
pmc() {  # mode, family-dump title, bench args...
  m=$1; title=$2; shift; shift
  # algorithmic bytes / flops per launch of every launcher family of this workload (bench.py's own accounting)
  GEMNET_DUMP_FAMILIES=$R/$OUT/fam_$m timeout 900 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fam_$m.json 2> $OUT/fam_$m.log
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${m}_$c -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 $B --no-extras > $R/$OUT/pmc_${m}_$c.log 2>&1 )
  done
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/pmc_${m}_sq -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 $B --no-extras > $R/$OUT/pmc_${m}_sq.log 2>&1 )
  python tools/pmc_summary.py $OUT $m $RND "$(ls $OUT/fam_${m}_*${title}*.json 2>/dev/null | head -1)" 2>&1 | tail -2
}
if has pmc; then
  echo "== PMC passes (separate, --pmc only)"
  pmc T forward_force
  pmc train training --mode train
  pmc Q forward_force --model Q
  pmc Qtrain training --model Q --mode train
fi
if has config4; then
  echo "== configs[4] shard: family dump + PMC passes"
  GEMNET_DUMP_FAMILIES=$R/$OUT/fam_config4 timeout 900 python tools/config4_shard.py families > $OUT/fam_config4.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_config4_$c -o p -- python $R/tools/config4_shard.py > $R/$OUT/pmc_config4_$c.log 2>&1 )
  done
  python tools/pmc_summary.py $OUT config4 $RND "$(ls $OUT/fam_config4_*.json 2>/dev/null | head -1)" 2>&1 | tail -2
fi
cp profiles/${RND}_pmc_* profiles/${RND}_traffic_* profiles/${RND}_mfma_busy.json $OUT/profiles/ 2>/dev/null
if has bench; then
  echo "== bench (default command of the driver; reads profiles/${RND}_traffic_* written above)"
  timeout 1800 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-300 $OUT/bench_default.json
  timeout 600 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_T_train.json 2> $OUT/bench_T_train.log; cut -c1-200 $OUT/bench_T_train.json
  timeout 600 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-200 $OUT/bench_Q_force.json
  timeout 600 python bench.py --model Q --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_Q_train.json 2> $OUT/bench_Q_train.log; cut -c1-200 $OUT/bench_Q_train.json
  for f in default T_train Q_force Q_train; do cp $OUT/bench_$f.json $OUT/profiles/${RND}_bench_$f.json; done
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
echo "== done"
