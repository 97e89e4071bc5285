#!/bin/bash
TAG=r4final5; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
