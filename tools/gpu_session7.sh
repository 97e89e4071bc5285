#!/bin/bash
TAG=${1:-s14}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== shards T"; timeout 600 python tools/shards_bench.py > $OUT/shards_T.txt 2> $OUT/shards_T.log; cat $OUT/shards_T.txt; tail -3 $OUT/shards_T.log
echo "== shards Q"; timeout 600 python tools/shards_bench.py Q > $OUT/shards_Q.txt 2> $OUT/shards_Q.log; cat $OUT/shards_Q.txt; tail -3 $OUT/shards_Q.log
echo "== bench force"; timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json; grep -A14 "per-family" $OUT/bench.log
echo "== bench train"; timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.log; cat $OUT/train.json; grep -A16 "per-family" $OUT/train.log
echo "== done"
