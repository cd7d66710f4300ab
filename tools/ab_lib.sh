# within-run A/B of two library builds: default vs im2im_uq_amd/lib/libim2im_uq_old.so
run() { python bench.py --legs train --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d.get('per_kernel') or {}
print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(d['roofline']['achieved'],1), 'fp32', round((d.get('fp32') or {}).get('value',0),1), {k.replace('conv_igemm_kernel',''):round(v['tflops']) for k,v in pk.items() if 'igemm' in k and 'taps=9' in k})" >> gpurun_out/ab_lib.log; }
: > gpurun_out/ab_lib.log
for i in 1 2; do IM2IM_LIB=$PWD/im2im_uq_amd/lib/libim2im_uq_old.so run old; run new; done
