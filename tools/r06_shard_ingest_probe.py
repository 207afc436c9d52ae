import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, bench, __graft_entry__ as e
pkg = e.load_package()
print(json.dumps(bench.sharded_host_ingest(pkg, 0), indent=1))
