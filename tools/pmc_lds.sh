#!/bin/bash
# LDS bank-conflict survey over the whole bench: tools/pmc_lds.sh <name>
out=/root/repo/gpurun_out/$1
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out -o r -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu > $out.log 2>&1
python /root/repo/tools/prof_summary.py $out/r_results.db | grep -E "^k_|^void k_" | grep -E "LDS_BANK|LDS_IDX" | cut -c1-46,92-150
