#!/bin/bash
# counters of the channelizer ring kernel k_pfbw<64,32> (BASELINE config 4 shape, 2^26 samples per launch): four --pmc passes, never with a trace
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  PROBE_IT=10 timeout 300 rocprofv3 --pmc $g -d $O/pfbc_$i -o r -- python $R/tools/pfb_probe.py > $O/pfbc_$i.log 2>&1
  python $R/tools/prof_summary.py $O/pfbc_$i/r_results.db | grep -E "k_pfbw|counter" | cut -c1-30,88-170
  rm -rf $O/pfbc_$i
  i=$((i+1))
done
PROBE_IT=400 python $R/tools/pfb_probe.py 2>&1 | tail -1
