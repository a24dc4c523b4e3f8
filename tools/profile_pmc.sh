#!/bin/bash
# SQ counter passes for one tools/prof_kernel.py case (each --pmc group in its own run):
#   tools/profile_pmc.sh gen4k   -> gpurun_out/pmc_gen4k/{pmc1,pmc2}
CASE=${1:-gen4k}
OUT=$PWD/gpurun_out/pmc_$CASE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- python $R/tools/prof_kernel.py --case $CASE --iters 3 > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed" >> $OUT/errors.log
done
