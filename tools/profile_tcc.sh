#!/bin/bash
# Round 3: L2 / fabric / texture-path counters of the 8K applyGainMap launches.
# Target = tools/kb_base (the shipping apply_gainmap.hip in a C++ harness, 20 launches back to back, two rotating buffer
# sets).  Every --pmc group runs in its own pass, together with --kernel-trace only, so each dispatch has its duration
# next to its counters (the 60..83 us spread of map A can be correlated with them).
#   tools/profile_tcc.sh [BIN]      -> gpurun_out/tcc/{A,C}/pmcN ; summary by tools/read_tcc.py
OUT=$PWD/gpurun_out/${PROF_DIR:-tcc}
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=${1:-$R/tools/kb_base}
mkdir -p $OUT
LIMIT=${LIMIT:-90}
limited() {
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
export KB_N=${KB_N:-20}
GROUPS_=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_STREAMING_REQ_sum TCC_WRITEBACK_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
)
for M in ${MAPS:-A C}; do
  mkdir -p $OUT/$M
  limited rocprofv3 --kernel-trace --stats -d $OUT/$M/trace -o t -- $BIN $M gauss > $OUT/$M/trace.log 2>&1
  i=0
  for G in "${GROUPS_[@]}"; do
    i=$((i+1))
    limited rocprofv3 --kernel-trace --pmc $G -d $OUT/$M/pmc$i -o p -- $BIN $M gauss > $OUT/$M/pmc$i.log 2>&1 || echo "map $M pmc group $i failed" >> $OUT/errors.log
  done
done
python $R/tools/read_tcc.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
