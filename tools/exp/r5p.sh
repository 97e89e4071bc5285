#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tests
run() { echo "== $*"; env "$@" timeout 300 python tools/exp/padded_train_debug.py 2>&1 | grep "^step" | head -3; }
run A=1
run GEMNET_K3_F16=0
run GEMNET_NATIVE_CSR=0
run GEMNET_CHAIN_MODE=split6
run GEMNET_AGGREGATE=0
run GEMNET_TRAIN2_BILINEAR=0
run GEMNET_TRAIN_OVERLAP=0
