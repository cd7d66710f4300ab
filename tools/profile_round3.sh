#!/bin/bash
# Round-3 evidence in one gpurun call: tools/profile_round3.sh r03
#   default bench line; rocprofv3 kernel stats of the train leg (overlapped = as timed, and isolated = weight-gradient stream
#   off), of the calibration leg, of the fp32 / temca1024 / bsbcm512 train legs; PMC FETCH_SIZE / WRITE_SIZE passes (each alone
#   with --kernel-trace, as MI355X_MICROARCH.md prescribes) for the calibration and conv kernels.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err
prof() {  # name, env, bench args...
  name=$1; shift; envs=$1; shift
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$name -o $tag -- python bench.py "$@" > $out/${tag}_${name}_leg_bench_line.json 2> $out/${tag}_${name}_leg.err
  find $out/${tag}_prof_$name -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${name}_kernel_stats.csv \;
}
prof train "X=1" --legs train --no-fp32 --steps 15 --warmup 3
prof train_isolated "IM2IM_WGRAD_STREAM=0" --legs train --no-fp32 --no-roofline --steps 10 --warmup 3
prof calib "X=1" --legs calib --no-cpu-baseline --steps 10 --warmup 1
prof fp32 "IM2IM_WGRAD_STREAM=0" --legs train --dtype fp32 --no-fp32 --no-roofline --steps 3 --warmup 1
prof temca1024 "IM2IM_WGRAD_STREAM=0" --legs train --config temca1024 --no-fp32 --no-roofline --steps 4 --warmup 2
prof bsbcm512 "IM2IM_WGRAD_STREAM=0" --legs train --config bsbcm512 --no-fp32 --no-roofline --steps 6 --warmup 2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_calib_$c -o pmc -- python bench.py --legs calib --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  IM2IM_WGRAD_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_train_$c -o pmc -- python bench.py --legs train --no-fp32 --no-roofline --steps 2 --warmup 1 > /dev/null 2>&1
done
ls $out | grep $tag | head -40
