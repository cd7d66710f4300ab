#!/bin/bash
# weight-gradient layer list (batch 78) per library variant, two interleaved rounds:  tools/ab_wgrad_libs.sh <tag> <name> [<name> ...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_wgrad_libs.txt
mkdir -p $root/gpurun_out; : > $out
cd $root
for rep in 1 2; do
  for name in "" "$@"; do
    suf=${name:+_$name}
    echo "== lib '${name:-shipped}' round $rep" >> $out
    IM2IM_LIB=$root/im2im_uq_amd/lib/libim2im_uq$suf.so python tools/bench_wgrad_ab.py ${BATCH:-78} 5 1 wgrad_co128 2>/dev/null | awk '{print $2, $3, $5, $6, $7, $8}' >> $out
  done
done
cat $out
