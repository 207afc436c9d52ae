#!/bin/bash
# round 4, config 5 (clXEngine 64 x 1024 x 1024 IChar): phase stamps, phase-removal timings, access-pattern ubench, TCP/TCC/TA/SQ counters
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_NINT=1
{
echo "== stamps"; MI355_XE_TS=1 PROBE_IT=3 python $R/tools/xe_batch_probe.py 2>&1 | tail -12
echo "== plain"; PROBE_IT=200 python $R/tools/xe_batch_probe.py
for d in 1 2 3 4 6 7; do echo "== MI355_XE_DBG=$d"; MI355_XE_DBG=$d PROBE_IT=200 python $R/tools/xe_batch_probe.py; done
echo "== second kernel instead of the in-launch reduction"; MI355_XE_INKERNEL_REDUCE=0 PROBE_IT=200 python $R/tools/xe_batch_probe.py
echo "== lockstep schedule instead of ping-pong"; MI355_XE_NO_PINGPONG=1 PROBE_IT=200 python $R/tools/xe_batch_probe.py
echo "== slice_read"; timeout 120 $R/tools/ubench/slice_read 2>&1 | grep -E "^---|W=32|dma"
echo "== slice_tl (finish time per 128-byte line index of the rows)"; timeout 120 $R/tools/ubench/slice_tl 2>&1 | grep -E "^W=32 stride 2048 |base \+ 128|stride 2304|prefetch 1"
} > $O/r04_xe_phases.txt 2>&1
rocprofv3 -L > $O/counters_avail.txt 2>&1
export PROBE_IT=20
i=0
: > $O/r04_xe_counters_raw.txt
while IFS= read -r g; do
  [ -z "$g" ] && continue
  out=$O/r04xe_p$i
  timeout 300 rocprofv3 --pmc $g -d $out -o r -- python $R/tools/xe_batch_probe.py > $out.log 2>&1
  echo "## pass $i: $g" >> $O/r04_xe_counters_raw.txt
  python $R/tools/prof_summary.py $out/r_results.db 2>&1 | grep -E "k_xe|counter" | sed -E 's/ {3,}/  /g' >> $O/r04_xe_counters_raw.txt
  tail -3 $out.log >> $O/r04_xe_counters_raw.txt
  rm -rf $out
  i=$((i+1))
done <<'GROUPS'
TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
TA_BUSY_avr TA_TA_BUSY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
GROUPS
