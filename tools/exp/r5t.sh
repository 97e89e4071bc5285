#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_index_in_graph.py -x -q -p no:cacheprovider 2>&1 | grep -v "Warning\|warn" | tail -25 | cut -c1-300 | tee $O/tests.txt
timeout 600 python -m pytest tests/test_gpu_md.py tests/test_gpu_padded.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -5 | tee -a $O/tests.txt
for v in 1 0; do echo "GEMNET_INDEX_IN_GRAPH=$v" | tee -a $O/md.txt; GEMNET_INDEX_IN_GRAPH=$v timeout 300 python tools/exp/md_bench.py 32 60 2>&1 | grep "GemNet-" | tee -a $O/md.txt; done
