bash tools/profile_round.sh r02 > gpurun_out/r02_profile_round.log 2>&1
bash tools/profile_isolated.sh r02 > /dev/null 2>&1
for c in temca1024 denoise32; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r02_bench_$c.json 2>/dev/null; done
python bench.py --config bsbcm512 --no-cpu-baseline > gpurun_out/r02_bench_bsbcm512_fp8.json 2>/dev/null
python bench.py --config bsbcm512 --dtype bf16 --no-cpu-baseline > gpurun_out/r02_bench_bsbcm512_bf16.json 2>/dev/null
python bench.py --batch 10 --no-cpu-baseline > gpurun_out/r02_bench_batch10.json 2>/dev/null
