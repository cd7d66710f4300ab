#!/bin/bash
# configs[4] in the fp8 mode: per-layer fp8 weight packs vs one launch per kind (IM2IM_FP8_BATCH_PACK), interleaved
tag=${1:-ab}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_fp8_pack.txt
: > $out
cd $root
run() {  # label, env
  label=$1; shift
  env "$@" python bench.py --legs train --config bsbcm512 --no-fp32 --no-roofline --steps 12 --warmup 4 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step')" >> $out
}
for rep in 1 2 3; do
  run "fp8 per-layer packs " IM2IM_FP8_BATCH_PACK=0
  run "fp8 batched packs   " IM2IM_FP8_BATCH_PACK=1
done
cat $out
