#!/bin/bash
# one SQ counter pass (cycles, waits, MFMA busy) + kernel durations per library variant:  tools/pmc_variants.sh shape kind variant...
shape=$1; kind=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  unset IM2IM_LIB IM2IM_CONV_ROLL
  case $v in new) ;; old) export IM2IM_CONV_ROLL=0;; one) export IM2IM_CONV_ROLL=2;; *) export IM2IM_LIB=$root/im2im_uq_amd/lib/libim2im_uq_$v.so;; esac
  rm -rf $root/gpurun_out/pmcv_$v
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmcv_$v -o pmc -- python $root/tools/conv_one.py $shape $kind 6 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/pmcv_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] or "conv_roll64" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob("$root/gpurun_out/pmcv_$v/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] or "conv_roll64" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
c = {n: sum(v) / len(v) for n, v in d.items()}
el = c["GRBM_GUI_ACTIVE"] / 8
md = sorted(dur)[len(dur) // 2]
print(f"{'$v':12s} cycles {el/1e3:7.0f}k  median {md:6.1f} us (first {dur[0]:.0f}, last {dur[-1]:.0f})  clock {el/md/1e3:.2f} GHz  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/(el*1024):.3f}  "
      f"wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f}  wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f}  wait_lds {c['SQ_WAIT_INST_LDS']/c['SQ_WAVE_CYCLES']:.3f}")
PY
done
