#!/bin/bash
# per-workgroup phase stamps of the fused X-engine kernel at config 5, three launches (the third one's dump is kept)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MI355_XE_TS=1 MI355_XE_TS_FILE=$O/r04_xe_stamps.txt PROBE_NINT=1 PROBE_IT=${1:-8} python $R/tools/xe_batch_probe.py 2>&1 | tail -9
