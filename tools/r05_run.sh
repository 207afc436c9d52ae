#!/bin/bash
# usage: tools/r05_run.sh <script-or-command...>   (runs from /tmp with TMPDIR set, as rocprofv3 wants)
cd /tmp && export TMPDIR=/tmp
exec "$@"
