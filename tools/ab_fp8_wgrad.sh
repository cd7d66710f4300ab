#!/bin/bash
# configs[4] (512x512, 2 input channels, batch 16): bf16 vs fp8 without / with the fp8 weight gradient, interleaved
tag=${1:-ab}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_fp8_wgrad.txt
: > $out
cd $root
run() {  # label, extra bench args, env
  label=$1; extra=$2; shift; shift
  env "$@" python bench.py --legs train --config bsbcm512 $extra --no-fp32 --no-roofline --steps 12 --warmup 4 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')" >> $out
}
for rep in 1 2; do
  run "bf16            " "--dtype bf16" X=1
  run "fp8 wgrad bf16  " "" IM2IM_FP8_WGRAD=0
  run "fp8 wgrad fp8   " "" IM2IM_FP8_WGRAD=1
done
cat $out
