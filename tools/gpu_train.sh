#!/bin/bash
OUT=gpurun_out/${1:-train}; mkdir -p $OUT
echo "== T train graph"; timeout 900 python bench.py --mode train --steps 10 --warmup 3 --no-roofline > $OUT/train.json 2> $OUT/train.log; cat $OUT/train.json | cut -c1-330; grep -v Warn $OUT/train.log | grep -B2 -A12 "Error\|Traceback\|unavailable" | tail -30
echo "== T train eager"; timeout 900 python bench.py --mode train --no-graph --steps 5 --warmup 2 --no-roofline > $OUT/train_e.json 2> $OUT/train_e.log; cat $OUT/train_e.json | cut -c1-330
