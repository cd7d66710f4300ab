#!/bin/bash
# The measured evidence of a round, in one gpurun call:  tools/profile_round.sh r02
#   bench line (default command), rocprofv3 kernel stats of the train and calibration legs (same commands with --legs),
#   PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace, as MI355X_MICROARCH.md prescribes).
# Everything lands in gpurun_out/<tag>_*; tools/summarise_profiles.py turns it into the files committed under profiles/.
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_train -o ${tag} -- python bench.py --legs train --no-fp32 --steps 15 --warmup 3 > $out/${tag}_train_leg_bench_line.json 2> $out/${tag}_train_leg.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_calib -o ${tag} -- python bench.py --legs calib --no-cpu-baseline --steps 10 --warmup 1 > $out/${tag}_calib_leg_bench_line.json 2> $out/${tag}_calib_leg.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_calib_$c -o pmc -- python bench.py --legs calib --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  IM2IM_WGRAD_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_train_$c -o pmc -- python bench.py --legs train --no-fp32 --no-roofline --steps 2 --warmup 1 > /dev/null 2>&1
done
ls $out | grep $tag
