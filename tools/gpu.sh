#!/bin/bash
# Developer tool: run a command on the GPU box from a FROZEN copy of the working tree, so that the tree can be edited while a call is
# queued (gpurun snapshots /root/repo when it gets a box, not when the call is made).
# usage: tools/gpu.sh <tag> <timeout_s> '<command run inside the frozen copy>'      (output: gpurun_out/<tag>/, also $RR_OUT on the box)
# Variant libraries (tools/variant.sh -> build/variants/lib_<name>.so, listed in .gpurunignore) travel only when the command names
# them as build/lib_<name>.so: they are copied into the frozen copy's build/.  The frozen copy is removed when ITS call has returned
# (never another call's: a second tools/gpu.sh made while the first is queued leaves the first one's copy alone).
set -e
tag=$1; tmo=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
snap=$root/build/snap_$tag
rm -rf $snap; mkdir -p $snap
tar -C $root --exclude=./.git --exclude=./gpurun_out --exclude=./build --exclude=__pycache__ --exclude=.pytest_cache --exclude=.hypothesis -cf - . | tar -C $snap -xf -
mkdir -p $snap/build
for lib in $(echo "$*" | grep -o 'build/lib_[A-Za-z0-9_]*\.so' | sort -u); do
  cp $root/build/variants/$(basename $lib) $snap/build/ || { echo "no such variant: $lib (tools/variant.sh builds build/variants/)"; rm -rf $snap; exit 1; }
done
[ -x $root/build/valu_issue ] && cp $root/build/valu_issue $snap/build/ || true
rc=0
gpurun --timeout $tmo -- "export RR_CODE_ROOT=\$GRAFT_REPO_ROOT/build/snap_$tag RR_OUT=\$GRAFT_REPO_ROOT/gpurun_out/$tag && mkdir -p \$RR_OUT && cd \$RR_CODE_ROOT && $*" || rc=$?
rm -rf $snap
exit $rc
