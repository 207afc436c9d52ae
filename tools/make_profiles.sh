#!/bin/bash
# Regenerates the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun):
#   tools/make_profiles.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write}.txt + <tag>_bench.json
# Pass 1: kernel trace + stats of the default bench.  Passes 2/3: FETCH_SIZE and WRITE_SIZE in separate --pmc runs
# (MI355X_MICROARCH.md, HBM section: one counter family per run; FETCH_SIZE counts 128-B requests as 64 B on gfx950).
tag=${1:-r01}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 3 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --sustain-s 0.5 > $O/${tag}_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-sustained > $O/${tag}_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_write -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-sustained > $O/${tag}_write.log 2>&1
python $R/tools/prof_summary.py $O/${tag}_stats/r_results.db > $O/${tag}_stats.txt 2>&1
python $R/tools/prof_summary.py $O/${tag}_fetch/r_results.db > $O/${tag}_fetch.txt 2>&1
python $R/tools/prof_summary.py $O/${tag}_write/r_results.db > $O/${tag}_write.txt 2>&1
rm -rf $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write  # the raw databases exceed what gpurun copies back; the summaries are the evidence
tail -1 $O/${tag}_bench.json | cut -c1-400
grep -E "^k_|^void k_" $O/${tag}_stats.txt | head -20
