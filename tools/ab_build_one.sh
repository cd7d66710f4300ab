#!/bin/bash
# alternative library that differs from the shipped one in ONE source file:  tools/ab_build_one.sh <name> <source> <extra flags...>
# (the file is compiled with build.py's flags for it + the extra ones, the other objects come from im2im_uq_amd/build/)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
base="-O3 -std=c++17 -fPIC -ffp-contract=off"
case $src in conv_wgrad.hip) base="$base -mllvm -amdgpu-sched-strategy=max-ilp";; esac
obj=/tmp/ab1_${name}_$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $base "$@" -c im2im_uq_amd/csrc/$src -o $obj 2>/dev/null
objs=$obj
for o in im2im_uq_amd/build/*.o; do case $o in */$src.o) ;; *) objs="$objs $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o im2im_uq_amd/lib/libim2im_uq_$name.so $objs
echo built im2im_uq_amd/lib/libim2im_uq_$name.so
