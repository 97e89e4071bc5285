#!/bin/bash
O=gpurun_out/r5u; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
for v in 1 0; do echo "GEMNET_INDEX_IN_GRAPH=$v" | tee -a $O/md.txt; for a in 64 48; do GEMNET_INDEX_IN_GRAPH=$v timeout 300 python tools/exp/md_bench.py $a 40 2>&1 | grep "GemNet-" | tee -a $O/md.txt; done; done
