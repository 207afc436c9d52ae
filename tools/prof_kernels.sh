#!/bin/bash
# usage: tools/prof_kernels.sh <name> <python script + args>: rocprofv3 kernel trace + per-kernel summary
out=/root/repo/gpurun_out/$1; shift
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out -o r -- "$@" > $out.log 2>&1
python /root/repo/tools/prof_summary.py $out/r_results.db | grep -E "^k_|^void k_" | cut -c1-60,88-140
