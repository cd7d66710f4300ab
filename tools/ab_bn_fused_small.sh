#!/bin/bash
# one-launch BatchNorm sums for few partial rows (im2im_set_option "bn_fused_small"): batch-10 step and the 32x32 config, interleaved
tag=${1:-ab}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_bn_fused_small.txt
: > $out
cd $root
run() {  # label, bench args, env...
  label=$1; args=$2; shift; shift
  env "$@" python bench.py --legs train $args --no-fp32 --no-roofline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')" >> $out
}
for rep in 1 2 3; do
  run "batch10 two-stage " "--batch 10 --steps 40 --warmup 8" IM2IM_BN_FUSED_SMALL=0
  run "batch10 fused     " "--batch 10 --steps 40 --warmup 8" IM2IM_BN_FUSED_SMALL=1
  run "denoise32 two-stage" "--config denoise32 --steps 200 --warmup 20" IM2IM_BN_FUSED_SMALL=0
  run "denoise32 fused    " "--config denoise32 --steps 200 --warmup 20" IM2IM_BN_FUSED_SMALL=1
done
cat $out
