cd /tmp
for i in 1 2 3; do for env in "X=1" "MI355_XE_DBG=2048"; do echo -n "$env: "; env $env PROBE_NINT=1 PROBE_IT=400 python /root/repo/tools/xe_batch_probe.py 2>&1 | tail -1; done; done
