#!/bin/bash
# last session of the round: the whole -m gpu suite, smoke(), the GemNet-Q artifacts (its kernels changed after gpu_artifacts4.sh ran)
TAG=r4final; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-200 $OUT/bench_Q_force.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_Q -o trace -- python $R/bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/rocprof_Q.log 2>&1 )
pmc() {
  m=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/pmc_${m}_$c -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_$c.log 2>&1 )
  done
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/pmc_${m}_sq -o p -- python $R/bench.py "$@" --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/$OUT/pmc_${m}_sq.log 2>&1 )
}
pmc Q --model Q
PYTHONPATH=$R:$R/tests timeout 300 python tools/exp/padded_ab.py > $OUT/padded_ab.txt 2>&1; grep "ms" $OUT/padded_ab.txt
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
echo "== done"
