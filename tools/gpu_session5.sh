#!/bin/bash
OUT=gpurun_out/${1:-s}; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log | cut -c1-300
echo "== T force"; timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.log; cut -c1-260 $OUT/bench.json
echo "== Q force"; timeout 900 python bench.py --model Q --steps 5 --warmup 2 --no-cpu-baseline > $OUT/q.json 2> $OUT/q.log; cut -c1-260 $OUT/q.json; grep -v Warn $OUT/q.log | grep "ms  " | head -8
