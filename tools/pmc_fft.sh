#!/bin/bash
# usage: tools/pmc_fft.sh <outdir-under-gpurun_out> <python script + args>
# Two SQ counter passes + GRBM for one command; summaries via tools/prof_summary.py
out=/root/repo/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d $out/p1 -o r -- "$@" > $out.p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $out/p2 -o r -- "$@" > $out.p2.log 2>&1
python /root/repo/tools/prof_summary.py $out/p1/r_results.db $out/p2/r_results.db | grep -E "k_fft|k_ols|k_pfb|k_xe|counter" | cut -c1-40,92-170
