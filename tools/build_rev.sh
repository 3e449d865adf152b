#!/bin/bash
# builds the library as of git revision $1 into build/variants/lib_$2.so (developer A/B on one GPU box)
set -e
rev=$1; name=$2; shift 2
d=$(mktemp -d)
mkdir -p $d/rustray_amd/csrc $d/include build/variants
for f in rr_api.hip rr_kernels.hip rr_bvh.cpp rr_bvh.h rr_device.h rr_math.h; do git show $rev:rustray_amd/csrc/$f > $d/rustray_amd/csrc/$f; done
git show $rev:include/rustray_hip.h > $d/include/rustray_hip.h
(cd $d/rustray_amd/csrc && hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 "$@" -shared -o $OLDPWD/build/variants/lib_$name.so rr_api.hip rr_bvh.cpp)
rm -rf $d
