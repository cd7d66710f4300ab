#!/bin/bash
# within-run A/B of an environment switch on the training leg:  tools/ab_env.sh VAR val1 val2 [extra bench args...]  (two alternating rounds)
var=$1; v1=$2; v2=$3; shift; shift; shift
run() { env $var=$1 python bench.py --legs train --no-fp32 --no-extras --no-live-pmc --no-cpu-baseline --steps 20 --warmup 5 "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d.get('roofline',{}).get('per_kernel') or d.get('per_kernel') or {}
print('$var=$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  conv', round(d['roofline']['achieved'],1), 'TF  wgrad', round(d.get('roofline_wgrad',{}).get('achieved',0),1), {k.replace('conv_igemm_kernel','').replace('conv_roll64_kernel','roll'):round(v['tflops']) for k,v in pk.items() if 'taps=9' in k or 'roll' in k})"; }
for i in 1 2; do run $v1 "$@"; run $v2 "$@"; done
