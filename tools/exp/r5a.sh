#!/bin/bash
# round 5, GPU call A: new correctness features + packed-FP32 repro / policy A/B + the full GPU suite
O=gpurun_out/r5a; mkdir -p $O
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_rangeflag.py tests/test_gpu_modes.py -q -x -s > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
timeout 600 sh tools/exp/pk_corun.sh $O/pk 200 > $O/pk_corun.log 2>&1; echo "pk_corun rc=$?" | tee -a $O/summary.txt
grep -h "RESULT\|replays, aggregation\|device" $O/pk/run_*.txt | tee -a $O/summary.txt
for rep in 1 2; do
  for v in default none legacy; do
    L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_pk_$v.so
    GEMNET_HIP_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1]); print("bench $v $rep:", d["value"], "mol/s", d["ms_per_step"], "ms; chain us", d["roofline"]["avg_launch_us"])
except Exception as e: print("bench $v $rep failed", e)
PY
  done
done
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -5 $O/gpu_tests.log | tee -a $O/summary.txt
