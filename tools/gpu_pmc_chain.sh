#!/bin/bash
OUT=gpurun_out/pmc_chain
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pre in 0 1; do
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" ; do
  tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p${pre}_$tag -o p -- python $GRAFT_REPO_ROOT/tools/chain_one.py 18122 5 $pre 20 > $GRAFT_REPO_ROOT/$OUT/log_${pre}_$tag.txt 2>&1
done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, pandas as pd
for f in sorted(glob.glob('gpurun_out/pmc_chain/*/*counter_collection.csv')):
    df = pd.read_csv(f)
    df = df[df['Kernel_Name'].str.contains('chain')]
    g = df.groupby('Counter_Name')['Counter_Value'].mean()
    print(f.split('/')[2][:14], dict(g.round(0)))
PY
tail -3 gpurun_out/pmc_chain/log_0_SQ_WAVES*.txt
