#!/bin/bash
O=gpurun_out/r5ab; mkdir -p $O
export PYTHONPATH=$PWD:$PWD/tests TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/tools/exp/t_dyn_ingraph_profile.py 30 > $R/$O/rocprof.log 2>&1 )
grep "ms/step" $O/rocprof.log
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) --list --marker=idx_atom_mol_kernel > $O/timeline.txt 2>&1; head -4 $O/timeline.txt
find $O -name "*.db" -delete
