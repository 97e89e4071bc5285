#!/bin/bash
TAG=${1:-s13}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
echo "== tn bench"; timeout 300 python tools/tn_bench.py > $OUT/tn_bench.txt 2>&1; cat $OUT/tn_bench.txt
echo "== bench force"; timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json
echo "== bench train"; timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.log; cat $OUT/train.json; tail -12 $OUT/train.log
bash tools/gpu_prof_train.sh $TAG/ptrain 2>&1 | tail -45
echo "== done"
