#!/bin/bash
# rocprofv3 evidence for the headline launch of bench.py (run on the GPU box through gpurun): one kernel-trace pass and
# one pass per PMC group (never combined with API traces), each in its own process group with a hard time limit so that
# a profiler that does not come back cannot eat the box's time.  Outputs under gpurun_out/prof_bench/;
#   python tools/read_prof.py gpurun_out/prof_bench > profiles/rNN_bench_mapC_batch16_rocprofv3.txt
# databases stay in /tmp on the GPU box (gpurun copies back at most 64 MiB): only the summary and the logs go to gpurun_out/
KEEP=$PWD/gpurun_out/prof_bench
OUT=/tmp/prof_bench
rm -rf $OUT $KEEP && mkdir -p $OUT $KEEP
R=${GRAFT_REPO_ROOT:-/root/repo}
LIMIT=${LIMIT:-150}
limited() {  # run "$@" in its own session; SIGKILL the whole group after $LIMIT seconds
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
export UHDR_BENCH_HEADLINE_ONLY=1  # no 8K north-star launches (same kernel name, other grid): the statistics below are the headline launch's
CMD="python $R/bench.py --steps 10 --warmup 2 --no-extra --no-cpu --no-config4"
limited rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || echo "trace pass failed / timed out" >> $OUT/errors.log
i=0
export UHDR_BENCH_CLOCK_RAMP_S=0.05  # the byte counters do not depend on the clock; 2000 ramp launches would only bloat the databases
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  limited rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed / timed out" >> $OUT/errors.log
done
python $R/tools/read_prof.py $OUT > $OUT/summary.txt 2>&1
cp $OUT/summary.txt $OUT/*.log $KEEP/ 2>/dev/null
cat $OUT/summary.txt | cut -c1-200 | head -20
