#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 300 python tools/exp/agg_v2_bench.py 2>&1 | grep -v Warning | tee $O/agg_v2.txt
for rep in 1 2; do for v in default aggv1; do
  L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so
  GEMNET_HIP_LIB=$L timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1]); print("bench $v $rep:", d["value"], "mol/s", d["ms_per_step"], "ms")
except Exception as e: print("bench $v $rep failed", e)
PY
done; done
for v in default aggv1; do
  L=""; [ $v != default ] && L=$PWD/tools/exp/bin/libgemnet_hip_$v.so
  GEMNET_HIP_LIB=$L timeout 400 python bench.py --mode train --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/train_${v}.json 2> $O/train_${v}.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/train_${v}.json").read().strip().splitlines()[-1]); print("train $v:", d["value"], d["unit"], d["ms_per_step"], "ms")
except Exception as e: print("train $v failed", e)
PY
done
timeout 600 sh tools/exp/pk_corun.sh $O/pk 200 > $O/pk_corun.log 2>&1; grep -h "RESULT\|replays, aggregation\|victim" $O/pk/run_*.txt | tee -a $O/summary.txt
python -m pytest tests/test_gpu_hbcheck.py tests/test_gpu_kernels.py -q -k "hbcheck or aggregate or replays or corun or stay_exact or conflict" > $O/tests.log 2>&1; tail -3 $O/tests.log | tee -a $O/summary.txt
