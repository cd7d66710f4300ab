#!/bin/bash
# round-6 launch-count switches, alternating on ONE box: train step at batch 78 and at the per-GPU batch of 10
#   tools/ab_r6.sh <tag>  ->  gpurun_out/<tag>_ab_r6.txt
tag=${1:-r6}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_ab_r6.txt
cd $root
run() {  # label, batch, steps, env...
  label=$1; b=$2; st=$3; shift; shift; shift
  env "$@" python bench.py --legs train --batch $b --no-fp32 --no-roofline --steps $st --warmup 8 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b  $label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step, host', round(d['host_enqueue_ms_per_step'],3))" >> $out
}
: > $out
for rep in 1 2 3; do
  for cfg in "78 20" "10 60"; do
    set -- $cfg
    run "r5 sequence (two-launch sums, per-layer reduce)" $1 $2 IM2IM_BN_ONELAUNCH=0 IM2IM_WGRAD_DEFER_REDUCE=0
    run "one-launch sums only                           " $1 $2 IM2IM_BN_ONELAUNCH=1 IM2IM_WGRAD_DEFER_REDUCE=0
    run "deferred reduce only                           " $1 $2 IM2IM_BN_ONELAUNCH=0 IM2IM_WGRAD_DEFER_REDUCE=1
    run "r6 default (both)                              " $1 $2 IM2IM_BN_ONELAUNCH=1 IM2IM_WGRAD_DEFER_REDUCE=1
  done
done
sort $out
