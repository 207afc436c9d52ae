#!/bin/bash
# Run stage of ci/gnuradio.Dockerfile (needs an MI355X): every generated flowgraph runs against the real scheduler, then the GPU suite.
set -e
for f in /tmp/flowgraphs/*.py; do
    echo "== $f"
    timeout 60 python3 "$f" < /dev/null   # head blocks end the graph; "run" mode returns when the scheduler is done
done
python3 __graft_entry__.py
python3 -m pytest tests -m gpu -x -q
