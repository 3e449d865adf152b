#!/bin/bash
# Developer tool: run a command on the GPU box from a FROZEN copy of the working tree, so that the tree can be edited while a call is
# queued (gpurun snapshots /root/repo when it gets a box, not when the call is made).
# usage: tools/gpu.sh <tag> <timeout_s> '<command run inside the frozen copy>'      (output: gpurun_out/<tag>/, also $RR_OUT on the box)
set -e
tag=$1; tmo=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $root/build/snap_*   # one call at a time: earlier snapshots are on their boxes already
snap=$root/build/snap_$tag
rm -rf $snap; mkdir -p $snap
tar -C $root --exclude=./.git --exclude=./gpurun_out --exclude='./build/snap_*' --exclude=__pycache__ --exclude=.pytest_cache -cf - . | tar -C $snap -xf -
exec gpurun --timeout $tmo -- "export RR_CODE_ROOT=\$GRAFT_REPO_ROOT/build/snap_$tag RR_OUT=\$GRAFT_REPO_ROOT/gpurun_out/$tag && mkdir -p \$RR_OUT && cd \$RR_CODE_ROOT && $*"
