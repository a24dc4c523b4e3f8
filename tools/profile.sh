#!/bin/bash
# rocprofv3 recipe for one kernel case (run on the GPU box through gpurun):
#   tools/profile.sh 8kA            -> gpurun_out/prof_8kA/{trace,pmc1..4}
# kernel-trace + stats in one run; each PMC group in its OWN run (never combined with API traces).
CASE=${1:-8kA}
OUT=$PWD/gpurun_out/prof_$CASE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/prof_kernel.py --case $CASE --iters 8 > $OUT/trace.log 2>&1
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" \
         "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- python $R/tools/prof_kernel.py --case $CASE --iters 4 > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed" >> $OUT/errors.log
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
