#!/bin/bash
# same-box comparison of the round-3 tree (_r3_tree/, git archive of 88a67d2 built in place) and this tree: default train leg and
# the batch-10 leg, interleaved.   tools/ab_r3_vs_r4.sh <tag>  ->  gpurun_out/<tag>_r3_vs_r4.txt
# prepare first:  mkdir _r3_tree && git archive 88a67d2 | tar -x -C _r3_tree && (cd _r3_tree && python -m im2im_uq_amd.build)
# (the directory is scratch: delete it afterwards, it is not part of the tree)
tag=${1:-ab}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${tag}_r3_vs_r4.txt
: > $out
run() {  # label, dir, args
  label=$1; dir=$2; shift; shift
  (cd $dir && python bench.py --legs train --no-fp32 "$@" 2>/dev/null) | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; w=d.get('roofline_wgrad') or {}; print('$label', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step  conv', round(r.get('frac',0),4), 'wgrad', round(w.get('frac',0),4))" >> $out
}
for rep in 1 2; do
  run "r3 batch78" $root/_r3_tree --steps 20 --warmup 5
  run "r4 batch78" $root --steps 20 --warmup 5
  run "r3 batch10" $root/_r3_tree --batch 10 --steps 40 --warmup 8
  run "r4 batch10" $root --batch 10 --steps 40 --warmup 8
done
cat $out
