#!/bin/bash
TAG=${1:-s17}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest all gpu (stacks off)"; timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
echo "== pytest model (stacks on)"; GEMNET_STACKS=1 timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== done"
