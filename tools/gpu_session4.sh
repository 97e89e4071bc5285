#!/bin/bash
OUT=gpurun_out/${1:-s}; mkdir -p $OUT
echo "== T force"; timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.log; cat $OUT/bench.json | cut -c1-300; grep "cpu\|Error" $OUT/bench.log | tail -3
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['cpu_baseline'])"
echo "== T train"; timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-roofline > $OUT/train.json 2> $OUT/train.log; cat $OUT/train.json | cut -c1-330; grep -v Warn $OUT/train.log | grep -B2 -A12 "Error\|Traceback" | tail -30
echo "== Q force"; timeout 900 python bench.py --model Q --steps 5 --warmup 2 --no-cpu-baseline > $OUT/q.json 2> $OUT/q.log; cat $OUT/q.json | cut -c1-700; grep -v Warn $OUT/q.log | tail -16
echo "== dist launcher, 1 rank"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>$OUT/dist.log | cut -c1-200; grep -i "error" $OUT/dist.log | tail -3
