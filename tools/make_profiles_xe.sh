#!/bin/bash
# rocprofv3 evidence for the X-engine IChar path alone (device resident, back-to-back launches), two command sets profiled separately
# (kernels of different geometries share a name, so each set gets its own passes):
#   tools/make_profiles_xe.sh <tag>  -> gpurun_out/<tag>_xe_*  (tools/probe.py rates xengine: BASELINE config 5 and its siblings)
#                                       gpurun_out/<tag>_xl_*  (tools/xe_large.py: rows > 64 = corner turn + k_xe_corr_sb; the per-rank
#                                                               128-channel problem one window per launch and batched)
#   then: python tools/collect_profiles_xe.py <tag>
tag=${1:-r03}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in xe xl; do
  if [ $set = xe ]; then CMD="python $R/tools/probe.py rates xengine"; else CMD="python $R/tools/xe_large.py"; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_${set}_stats -o r -- $CMD > $O/${tag}_${set}_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_${set}_fetch -o r -- $CMD > $O/${tag}_${set}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_${set}_write -o r -- $CMD > $O/${tag}_${set}_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${tag}_${set}_mfma -o r -- $CMD > $O/${tag}_${set}_mfma.log 2>&1
  for k in stats fetch write mfma; do python $R/tools/prof_summary.py $O/${tag}_${set}_$k/r_results.db > $O/${tag}_${set}_$k.txt 2>&1; done
  for k in stats fetch write mfma; do rm -rf $O/${tag}_${set}_$k; done
  grep -h "clXEngine\|per-rank\|single GPU" $O/${tag}_${set}_stats.log
  grep -E "^k_xe" $O/${tag}_${set}_stats.txt | cut -c1-70,88-140
done
