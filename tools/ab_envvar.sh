#!/bin/bash
# A/B of one environment variable on the train leg at batch 78 and 10, alternating on one box:  tools/ab_envvar.sh VAR v1 v2 [tag]
var=$1; v1=$2; v2=$3; tag=${4:-envab}
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/${tag}_${var}.txt; cd $root; : > $out
run() { env $var=$1 python bench.py --legs train --batch $2 --no-fp32 --no-roofline --steps $3 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $2  $var=$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms/step, host', round(d['host_enqueue_ms_per_step'],3))" >> $out; }
for rep in 1 2 3; do for cfg in "78 20" "10 60"; do set -- $cfg; run $v1 $1 $2; run $v2 $1 $2; done; done
sort $out
