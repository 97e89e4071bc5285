#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_index_in_graph.py -x -q 2>&1 | tail -30 | tee $O/tests.txt
timeout 600 python -m pytest tests/test_gpu_md.py tests/test_gpu_padded.py tests/test_gpu_rangeflag.py tests/test_gpu_index.py -x -q 2>&1 | tail -5 | tee -a $O/tests.txt
for v in 1 0; do echo "GEMNET_INDEX_IN_GRAPH=$v" | tee -a $O/md.txt; GEMNET_INDEX_IN_GRAPH=$v timeout 300 python tools/exp/md_bench.py 32 60 2>&1 | grep "GemNet-T" | tee -a $O/md.txt; GEMNET_INDEX_IN_GRAPH=$v timeout 300 python tools/exp/md_bench.py 64 40 2>&1 | grep "GemNet-T" | tee -a $O/md.txt; done
