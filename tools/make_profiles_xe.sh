#!/bin/bash
# rocprofv3 evidence for the X-engine IChar path alone (BASELINE config 5, device resident, back-to-back launches):
#   tools/make_profiles_xe.sh <tag>  -> gpurun_out/<tag>_xe_{stats,fetch,write}.txt   (copy into profiles/)
tag=${1:-r02}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/probe.py rates xengine"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_xe_stats -o r -- $CMD > $O/${tag}_xe_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_xe_fetch -o r -- $CMD > $O/${tag}_xe_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_xe_write -o r -- $CMD > $O/${tag}_xe_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${tag}_xe_mfma -o r -- $CMD > $O/${tag}_xe_mfma.log 2>&1
for k in stats fetch write mfma; do python $R/tools/prof_summary.py $O/${tag}_xe_$k/r_results.db > $O/${tag}_xe_$k.txt 2>&1; done
grep -h "clXEngine" $O/${tag}_xe_stats.log
grep -E "^k_xe" $O/${tag}_xe_stats.txt | cut -c1-70,88-140
grep -E "^k_xe_i8" $O/${tag}_xe_fetch.txt $O/${tag}_xe_write.txt $O/${tag}_xe_mfma.txt | grep -E "FETCH|WRITE|MFMA|BUSY|GUI" | cut -c1-200
