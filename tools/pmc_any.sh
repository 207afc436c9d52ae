#!/bin/bash
# usage: tools/pmc_any.sh <name> "<counters of pass 1>;<counters of pass 2>;..." <python script + args>
# one rocprofv3 --pmc pass per ';'-separated counter group (never combined with a trace), per-kernel averages printed
name=$1; groups=$2; shift 2
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra G <<< "$groups"
i=0
for g in "${G[@]}"; do
  out=/root/repo/gpurun_out/${name}_p$i
  timeout 300 rocprofv3 --pmc $g -d $out -o r -- "$@" > $out.log 2>&1
  python /root/repo/tools/prof_summary.py $out/r_results.db | grep -E "k_|counter" | cut -c1-50,93-170
  i=$((i+1))
done
