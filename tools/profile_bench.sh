#!/bin/bash
# rocprofv3 evidence for bench.py's roofline object (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the SAME bench command  -> average duration of the dominant kernel
#   2. FETCH_SIZE / WRITE_SIZE in their own --pmc passes -> HBM traffic per launch
# Outputs under gpurun_out/prof_bench/, summarised by tools/read_prof.py into profiles/.
OUT=$PWD/gpurun_out/prof_bench
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 10 --warmup 2 --no-extra --no-cpu $@"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o p -- python $R/bench.py $ARGS > $OUT/pmc1.log 2>&1 || echo "FETCH_SIZE pass failed" >> $OUT/errors.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc2 -o p -- python $R/bench.py $ARGS > $OUT/pmc2.log 2>&1 || echo "WRITE_SIZE pass failed" >> $OUT/errors.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc3 -o p -- python $R/bench.py $ARGS > $OUT/pmc3.log 2>&1 || echo "SQ pass failed" >> $OUT/errors.log
tail -1 $OUT/trace.log
