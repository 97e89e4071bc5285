#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
for w in T Q; do
 for env in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; do
   echo "== $w env: $env"; env $env timeout 200 python tools/exp/replay_host_cost.py $w 2>&1 | grep "GemNet-"
 done
done | tee $O/replay_host.txt
