#!/bin/bash
# HEAD-state rocprofv3 evidence for every kernel bench.py reports (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of ONE process that launches each kernel a few times (tools/qbench.py)
#   2. FETCH_SIZE, WRITE_SIZE and two SQ groups, each --pmc group in its own pass (never combined with API traces)
# Outputs under gpurun_out/prof_all/; tools/read_prof.py turns them into the text committed under profiles/.
# databases stay in /tmp on the GPU box (gpurun copies back at most 64 MiB): only the summary and the logs go to gpurun_out/
KEEP=$PWD/gpurun_out/${PROF_DIR:-prof_all}
OUT=/tmp/${PROF_DIR:-prof_all}
rm -rf $OUT; mkdir -p $OUT $KEEP
R=${GRAFT_REPO_ROOT:-/root/repo}
CASES="${@:-8kC 8kB 8kA 4kAhlg 4kApq b32hlg tm4k gen4k gen4k1 tm8k api0f api1f api1f8k fdct4k idct4k cvt4k huff4k}"
export QB_REPS=1 QB_ITERS=4 QB_NO_SERIAL=1
LIMIT=${LIMIT:-200}
limited() {  # run "$@" in its own session; SIGKILL the whole group after $LIMIT seconds (a profiler that does not come back must not eat the box's time)
  setsid "$@" &
  local pid=$!
  ( sleep $LIMIT; kill -KILL -- -$pid 2>/dev/null ) &
  local wd=$!
  wait $pid
  local rc=$?
  kill $wd 2>/dev/null
  return $rc
}
cd /tmp && export TMPDIR=/tmp
limited rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/qbench.py $CASES > $OUT/trace.log 2>&1
i=0
[ "${PMC:-1}" = "0" ] || for G in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  limited rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- python $R/tools/qbench.py $CASES > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed" >> $OUT/errors.log
done
python $R/tools/read_prof.py $OUT > $OUT/summary.txt 2>&1
cp $OUT/summary.txt $OUT/*.log $KEEP/ 2>/dev/null
tail -3 $OUT/trace.log
