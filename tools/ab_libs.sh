#!/bin/bash
# whole-library variants (tools/ab_build.sh <name> <flags>) against the shipped build on the BASELINE layer shapes, batch 78:
#   tools/ab_libs.sh <tag> <name> [<name> ...]   ->  gpurun_out/<tag>_libs.txt   (forward + data-gradient list, weight-gradient list)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_libs.txt
: > $out
cd $root
for rep in 1 2; do
  for name in "" "$@"; do
    suf=${name:+_$name}
    echo "== lib '${name:-shipped}' round $rep" >> $out
    IM2IM_LIB=$root/im2im_uq_amd/lib/libim2im_uq$suf.so python tools/bench_conv_ab.py 78 5 0 conv_splitk 2>/dev/null | tail -1 >> $out
    IM2IM_LIB=$root/im2im_uq_amd/lib/libim2im_uq$suf.so python tools/bench_wgrad_ab.py 78 5 1 wgrad_co128 2>/dev/null | tail -1 >> $out
  done
done
cat $out
