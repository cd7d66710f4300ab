#!/bin/bash
# Round-6 evidence in one gpurun call: tools/profile_round6.sh r06   (then tools/summarise_profiles.py r05 + tools/pmc_busy_summary.py r05)
#   default bench line; rocprofv3 kernel stats of the train leg (overlapped = as timed, and isolated = weight-gradient stream
#   off), of the calibration leg, of the batch-10 step (isolated); PMC FETCH_SIZE / WRITE_SIZE passes (each alone with
#   --kernel-trace) for the calibration and conv kernels; SQ MFMA-busy / LDS counters of the conv kernels on two layer shapes.
tag=${1:-r06}   # (evidence files are named per round; a re-run inside the round overwrites them)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err
prof() {  # name, env, bench args...
  name=$1; shift; envs=$1; shift
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$name -o $tag -- python bench.py "$@" > $out/${tag}_${name}_leg_bench_line.json 2> $out/${tag}_${name}_leg.err
  find $out/${tag}_prof_$name -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${name}_kernel_stats.csv \;
  rm -rf $out/${tag}_prof_$name
}
prof train "X=1" --legs train --no-fp32 --steps 15 --warmup 3
prof train_isolated "IM2IM_WGRAD_STREAM=0" --legs train --no-fp32 --no-roofline --steps 10 --warmup 3
prof calib "X=1" --legs calib --no-cpu-baseline --steps 10 --warmup 1
prof batch10_isolated "IM2IM_WGRAD_STREAM=0 IM2IM_HIP_GRAPH=0" --legs train --batch 10 --no-fp32 --no-roofline --steps 20 --warmup 5
prof bsbcm512 "IM2IM_WGRAD_STREAM=0" --legs train --config bsbcm512 --no-fp32 --no-roofline --steps 6 --warmup 2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_calib_$c -o pmc -- python bench.py --legs calib --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  IM2IM_WGRAD_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_train_$c -o pmc -- python bench.py --legs train --no-fp32 --no-roofline --steps 2 --warmup 1 > /dev/null 2>&1
done
# SQ counters of the conv kernels (tools/bench_conv.py: forward with statistics + weight gradient of one layer shape)
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  for shape in 78,40,40,512,512 78,160,160,128,128 78,320,320,64,64; do
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/${tag}_pmc_sq/p${i}_$shape -o pmc -- python tools/bench_conv.py $shape 5 > /dev/null 2>&1
  done
done
python tools/pmc_busy_summary.py $tag > $out/${tag}_pmc_conv_sq_counters.txt 2>&1
ls $out | grep $tag | head -60
