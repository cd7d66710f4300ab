for rep in 1 2; do
for v in base fp8noconv fp8half; do
  if [ $v = base ]; then unset IM2IM_LIB; else export IM2IM_LIB=$PWD/im2im_uq_amd/lib/libim2im_uq_$v.so; fi
  echo "== $v"; python tools/bench_fp8_conv.py 2>&1 | grep -v amdgpu
  python bench.py --config bsbcm512 --legs train --no-fp32 --no-extras --no-live-pmc --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bsbcm512 fp8 step', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done; done
unset IM2IM_LIB
python bench.py --config bsbcm512 --dtype bf16 --legs train --no-fp32 --no-extras --no-live-pmc --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 bsbcm512 step', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
