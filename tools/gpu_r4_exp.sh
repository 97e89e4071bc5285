#!/bin/bash
O=gpurun_out/r4_mfma; mkdir -p $O
export PYTHONPATH=.:tests
timeout 600 python -m pytest tests/test_gpu_padded.py -x -q -m gpu 2>&1 | tail -1
for v in "" "-DGN_EXP=1" "-DGN_EXP=2"; do
  GN_TRACE_DEFS="$v" timeout 300 python tools/chain2_trace.py --quick --modes=h3 2>&1 | grep -v amdgpu.ids > "$O/trace_edge${v}.txt"
  echo "== [$v]"; grep "wave0 op[1-3]\|wave7 op[1-3]" "$O/trace_edge${v}.txt" | head -6
done
