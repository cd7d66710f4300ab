# within-run A/B of library builds: tools/ab_lib.sh <suffix|new> ...   (new = the default library)
run() { python bench.py --legs train --no-fp32 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d.get('per_kernel') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(d['roofline']['achieved'],1), {k.replace('conv_igemm_kernel',''):round(v['tflops']) for k,v in pk.items() if 'igemm' in k and 'taps=9' in k})" >> gpurun_out/ab_lib.log; }
: > gpurun_out/ab_lib.log
for i in 1 2; do for v in "$@"; do if [ "$v" = "new" ]; then run new; else IM2IM_LIB=$PWD/im2im_uq_amd/lib/libim2im_uq_$v.so run $v; fi; done; done
