#!/bin/bash
# batch-10 step (the per-GPU share of the reference's global batch of 78 on 8 GPUs): eager vs HIP graph, split-K targets.
#   tools/ab_batch10.sh <tag>  ->  gpurun_out/<tag>_ab_batch10.txt
tag=${1:-ab}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_ab_batch10.txt
cd $root
run() {  # label, env...
  label=$1; shift
  env "$@" python bench.py --legs train --batch 10 --no-fp32 --no-roofline --steps 40 --warmup 8 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step, host', round(d['host_enqueue_ms_per_step'],3))" >> $out
}
: > $out
for rep in 1 2; do
  run "eager splitk=3" IM2IM_HIP_GRAPH=0
  run "graph splitk=3" IM2IM_HIP_GRAPH=1
  run "eager splitk=0" IM2IM_HIP_GRAPH=0 IM2IM_CONV_SPLITK=0
  run "graph splitk=0" IM2IM_HIP_GRAPH=1 IM2IM_CONV_SPLITK=0
  run "graph splitk=2" IM2IM_HIP_GRAPH=1 IM2IM_CONV_SPLITK=2
  run "graph splitk=4" IM2IM_HIP_GRAPH=1 IM2IM_CONV_SPLITK=4
  run "graph splitk=3 no-side-stream" IM2IM_HIP_GRAPH=1 IM2IM_WGRAD_STREAM=0
done
cat $out
