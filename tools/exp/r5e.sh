#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 300 python tools/exp/agg_v2_bench.py 2>&1 | grep -v Warning | tee $O/agg_v2.txt
timeout 300 python tools/exp/train_glue.py > $O/train_glue.txt 2>&1; tail -50 $O/train_glue.txt | cut -c1-230
