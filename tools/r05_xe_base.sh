#!/bin/bash
# round 5 baseline: config 5 from HBM, single and batched, with per-workgroup phase stamps
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== single window, 4 inputs in rotation"
PROBE_NBUF=4 python $R/tools/xe_rotate_probe.py 2>&1 | tail -1
echo "== 8 windows per launch, 2 inputs in rotation"
PROBE_NINT=8 PROBE_NBUF=2 PROBE_IT=20 python $R/tools/xe_rotate_probe.py 2>&1 | tail -1
echo "== 4 windows per launch, 2 inputs in rotation"
PROBE_NINT=4 PROBE_NBUF=2 PROBE_IT=20 python $R/tools/xe_rotate_probe.py 2>&1 | tail -1
echo "== stamps single"
MI355_XE_TS=1 MI355_XE_TS_FILE=$O/r05_stamps_single.txt PROBE_NBUF=4 PROBE_IT=1 python $R/tools/xe_rotate_probe.py 2>&1 | tail -12
echo "== stamps 8 windows"
MI355_XE_TS=1 MI355_XE_TS_FILE=$O/r05_stamps_n8.txt PROBE_NINT=8 PROBE_NBUF=2 PROBE_IT=1 python $R/tools/xe_rotate_probe.py 2>&1 | tail -12
echo "== stamps 4 windows"
MI355_XE_TS=1 MI355_XE_TS_FILE=$O/r05_stamps_n4.txt PROBE_NINT=4 PROBE_NBUF=2 PROBE_IT=1 python $R/tools/xe_rotate_probe.py 2>&1 | tail -12
