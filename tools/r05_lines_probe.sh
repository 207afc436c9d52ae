#!/bin/bash
# round 5: the whole-line X-engine kernel at config 5, 8 windows per launch from HBM: phase removal (MI355_XE_DBG) and per-workgroup stamps
cd /root/repo
export PROBE_NINT=${PROBE_NINT:-8} PROBE_NBUF=2 PROBE_IT=10
for d in ${DBGS:-0 2 4 6}; do echo -n "dbg $d: "; MI355_XE_DBG=$d timeout 120 python tools/xe_rotate_probe.py 2>&1 | tail -1; done
echo -n "old kernel: "; MI355_XE_NO_LINES=1 timeout 120 python tools/xe_rotate_probe.py 2>&1 | tail -1
MI355_XE_TS=1 PROBE_IT=1 timeout 120 python tools/xe_rotate_probe.py 2>&1 | grep "lines stamps" | tail -1
