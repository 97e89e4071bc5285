#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests
timeout 300 python tools/exp/padded_train_debug.py 2>&1 | grep "^step" | head -6 | tee $O/dbg.txt
timeout 900 python -m pytest tests/test_gpu_padded.py tests/test_gpu_md.py tests/test_gpu_index_in_graph.py tests/test_gpu_rangeflag.py tests/test_gpu_trainer.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python - <<PY 2>/dev/null | tee $O/dyn.txt
import json, torch, bench
cfg = dict(bench.GEMNET_T)
print(json.dumps(bench.extra_train_dynamic(cfg, 1234, 32, 32, 0)))
PY
