#!/bin/bash
# round 6: first run of the row-resident chain layout (csrc/chain4.hip)
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "test_chain_kernel_vs_interpreter and h3" -s 2>&1 | tail -25 > gpurun_out/r6_chain4_test.txt
timeout 300 python tools/chain_programs.py > gpurun_out/r6_chain4_programs.txt 2>&1
timeout 600 python tools/exp/q64_diag.py q64s > gpurun_out/r6_q64_diag.txt 2>&1
tail -5 gpurun_out/r6_chain4_test.txt; tail -16 gpurun_out/r6_chain4_programs.txt
