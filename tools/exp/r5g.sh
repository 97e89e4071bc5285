#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 300 python tools/chain2_trace.py --quick --small --modes=h3 2>&1 | grep -v Warning | tee $O/trace_small.txt
timeout 300 python tools/chain2_trace.py --quick --modes=h3 2>&1 | grep -v Warning | tee $O/trace_edge.txt
