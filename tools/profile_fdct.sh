#!/bin/bash
# Round 3: the MFMA FDCT experiment (tools/fdct_mfma.hip) against the shipping butterfly kernel under rocprofv3:
# kernel-trace durations, then instruction / busy counters, one --pmc group per pass.   -> gpurun_out/fdct_mfma/summary.txt
OUT=$PWD/gpurun_out/${PROF_DIR:-fdct_mfma}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/fdct_mfma 2000 > $OUT/unprofiled.txt 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $R/tools/fdct_mfma 300 > $OUT/trace.log 2>&1
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- $R/tools/fdct_mfma 20 > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed" >> $OUT/errors.log
done
{ echo "# unprofiled (2000 launches each after a 2000-launch clock ramp)"; cat $OUT/unprofiled.txt; python $R/tools/read_prof.py $OUT; } > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
