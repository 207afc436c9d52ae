#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh <rounds> <command...>   -- runs the command with the tree's library ("new") and with tools/_lib_old.so
# ("old", a build of an earlier state left there by hand) alternately: same box, same process environment, box-to-box spread removed
L=/root/repo/gr-clenabled_amd/libmi355_clenabled.so
n=$1; shift
cp $L /tmp/_new.so
for r in $(seq $n); do
  cp /tmp/_new.so $L; echo "== new"; "$@" 2>&1 | grep -v amdgpu.ids
  cp /root/repo/tools/_lib_old.so $L; echo "== old"; "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/_new.so $L
