#!/bin/bash
# Round artifacts (GPU box): the driver-contract bench line, rocprofv3 kernel stats of the same command, PMC traffic
# (separate passes, no trace domains), the GemNet-Q and training-step profiles, the chain micro-benchmarks and the
# BASELINE configs[4] shard.   bash tools/gpu_artifacts.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench (default command of the driver)"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-400 $OUT/bench_default.json
echo "== bench Q force"; timeout 900 python bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_Q_force.json 2> $OUT/bench_Q_force.log; cut -c1-300 $OUT/bench_Q_force.json
echo "== bench f32-MFMA chain (A/B of the Dense-stack arithmetic)"; timeout 900 python bench.py --chain-mode f32 --no-cpu-baseline --no-extras > $OUT/bench_T_chain_f32.json 2> $OUT/bench_T_chain_f32.log; cut -c1-300 $OUT/bench_T_chain_f32.json
echo "== rocprof kernel stats (same command as the bench line, hipGraph replay)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
echo "== timeline of one hipGraph replay (queues, gaps, exclusive time per kernel)"
python tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) --list > $OUT/timeline.txt 2>&1; head -5 $OUT/timeline.txt
echo "== rocprof kernel stats, training step"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1 )
echo "== rocprof kernel stats, GemNet-Q"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_Q -o trace -- python $GRAFT_REPO_ROOT/bench.py --model Q --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $GRAFT_REPO_ROOT/$OUT/rocprof_Q.log 2>&1 )
echo "== PMC traffic (separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1 )
done
echo "== PMC: MFMA / wait counters of the chain kernel"
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1 )
echo "== chain micro-benchmarks"
timeout 300 python tools/chain_bench.py > $OUT/chain_bench.txt 2>&1
timeout 300 python tools/chain_programs.py > $OUT/chain_programs.txt 2>&1; tail -13 $OUT/chain_programs.txt
timeout 300 python tools/chain2_trace.py > $OUT/chain2_trace.txt 2>&1
timeout 100 python tools/exp/agg_bench.py 2>&1 | grep " us" > $OUT/agg_bench.txt
./tools/exp/bin/lds_read_bench > $OUT/lds_read_bench.txt 2>&1
./tools/exp/bin/wfetch_bench > $OUT/wfetch.txt 2>&1
echo "== BASELINE configs[4] shard (64 molecules x 64 atoms, GemNet-Q, bf16 operands vs default)"
timeout 900 python tools/config4_shard.py 64 64 > $OUT/config4_shard.txt 2>&1; tail -2 $OUT/config4_shard.txt | cut -c1-900
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== done"
