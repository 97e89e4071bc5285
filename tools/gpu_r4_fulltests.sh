#!/bin/bash
# the whole -m gpu suite + one headline bench line
O=gpurun_out/r4_full2; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python bench.py --no-config4 --steps 60 --warmup 10 > $O/bench.log 2>&1; grep -h '^{"metric"' $O/bench.log | cut -c1-400
