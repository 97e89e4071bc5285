#!/bin/bash
# The two misbehaving capture configurations under runtime switches of the HIP graph path: which one removes the symptom?
OUT=gpurun_out/${1:-r4_env}
mkdir -p $OUT
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; PYTHONPATH=.:tests timeout 600 python tools/hbcheck_run.py T64-rbfout-side train-overlap > $OUT/$name.txt 2> $OUT/$name.err )
  echo "--- $name: $(grep '^summary' $OUT/$name.txt)"
}
run base X=1
run nocapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run hwq1 GPU_MAX_HW_QUEUES=1
run hwq2 GPU_MAX_HW_QUEUES=2
