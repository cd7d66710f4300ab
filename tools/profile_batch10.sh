#!/bin/bash
# per-kernel durations of the config-faithful per-GPU batch (78/8 = 10): tools/profile_batch10.sh <tag>
#   isolated (weight-gradient stream off: every kernel alone on the chip) and overlapped (as the step is timed)
tag=${1:-b10}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
for mode in iso ovl; do
  ws=1; [ $mode = iso ] && ws=0
  IM2IM_WGRAD_STREAM=$ws IM2IM_HIP_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_b10_$mode -o ${tag} -- \
    python bench.py --legs train --batch 10 --no-fp32 --no-roofline --steps 20 --warmup 5 > $out/${tag}_b10_${mode}_line.json 2> $out/${tag}_b10_${mode}.err
  find $out/${tag}_prof_b10_$mode -name "*kernel_stats.csv" -exec cp {} $out/${tag}_b10_${mode}_kernel_stats.csv \;
  find $out/${tag}_prof_b10_$mode -name "*kernel_trace.csv" -exec cp {} $out/${tag}_b10_${mode}_kernel_trace.csv \;
  rm -rf $out/${tag}_prof_b10_$mode
done
