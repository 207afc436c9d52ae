#!/bin/bash
# rocprofv3 evidence per BASELINE config, ONE launch size per command (tools/baseline_cfg.py):
#   tools/make_profiles_cfg.sh <tag> [configs...]  -> gpurun_out/<tag>_cfg<k>_{run.json,stats.txt,fetch.txt,write.txt}
#   then: python tools/collect_profiles_cfg.py <tag>   -> profiles/<tag>_baseline_configs.txt + profiles/baseline_configs_pmc.json
# Kernel trace and the two PMC passes are separate runs (MI355X_MICROARCH.md, HBM section: one counter family per run).
tag=${1:-r06}
shift
cfgs=${@:-2 3 4 5 5b}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in $cfgs; do
  CMD="python $R/tools/baseline_cfg.py $k"
  timeout 300 $CMD 100 > $O/${tag}_cfg${k}_run.json 2> $O/${tag}_cfg${k}_run.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_cfg${k}_stats -o r -- $CMD 100 > $O/${tag}_cfg${k}_stats.log 2>&1
  BASELINE_CFG_WARM_S=0.2 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_cfg${k}_fetch -o r -- $CMD 10 > $O/${tag}_cfg${k}_fetch.log 2>&1
  BASELINE_CFG_WARM_S=0.2 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_cfg${k}_write -o r -- $CMD 10 > $O/${tag}_cfg${k}_write.log 2>&1
  for p in stats fetch write; do
    python $R/tools/prof_summary.py $O/${tag}_cfg${k}_$p/r_results.db > $O/${tag}_cfg${k}_$p.txt 2>&1
    rm -rf $O/${tag}_cfg${k}_$p
  done
  tail -1 $O/${tag}_cfg${k}_run.json
  grep -E "^k_" $O/${tag}_cfg${k}_stats.txt | head -3 | cut -c1-60,88-140
done
