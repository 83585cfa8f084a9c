#!/bin/bash
# SQ / TCC counter passes of one configuration (rocprofv3 --pmc runs, one per counter set), per-kernel averages:
#   bash tools/pmc_sq.sh <out-name> [case] [iters]      (env such as GPUNTT_TWO_SWEEP_BIG is inherited; cases: tools/run_case.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASE=${2:-c2}
IT=${3:-10}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/$1/p$i -o pmc -- python $R/tools/run_case.py $CASE $IT > $R/gpurun_out/$1.p$i.log 2>&1 || true
done
cd $R && python tools/pmc_dump.py gpurun_out/$1 | grep -v prep_tw
