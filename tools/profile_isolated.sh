#!/bin/bash
# per-kernel durations with every kernel ALONE on the chip (weight-gradient stream off): tools/profile_isolated.sh <tag>
tag=${1:-iso}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
IM2IM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_iso -o ${tag} -- python bench.py --legs train --no-fp32 --no-roofline --steps 10 --warmup 3 > $out/${tag}_iso_line.json 2> $out/${tag}_iso.err
find $out/${tag}_prof_iso -name "*kernel_stats.csv" -exec cp {} $out/${tag}_iso_kernel_stats.csv \;
