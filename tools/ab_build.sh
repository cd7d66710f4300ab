#!/bin/bash
# build an alternative libim2im_uq with extra -D flags for within-run A/B:  tools/ab_build.sh <name> <flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=im2im_uq_amd/lib/libim2im_uq_$name.so
objs=""
for f in common.cpp hb_bound.cpp rcps.hip conv_mfma.hip conv_wgrad.hip elementwise.hip smallconv.hip fastmri.hip conv_fp8.hip; do
  x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc $x --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c im2im_uq_amd/csrc/$f -o /tmp/ab_${name}_$f.o 2>/dev/null &
  objs="$objs /tmp/ab_${name}_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
echo built $out
