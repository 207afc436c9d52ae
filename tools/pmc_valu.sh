#!/bin/bash
# usage: tools/pmc_valu.sh <name> <python script + args>   (VALU issue / wait counters, each pass under timeout)
out=/root/repo/gpurun_out/$1; shift
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU -d $out/p1 -o r -- "$@" > $out.p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA -d $out/p2 -o r -- "$@" > $out.p2.log 2>&1
python /root/repo/tools/prof_summary.py $out/p1/r_results.db $out/p2/r_results.db | grep -E "k_fft|k_ols|k_pfb|k_xe|k_fir" | cut -c1-34,92-150
