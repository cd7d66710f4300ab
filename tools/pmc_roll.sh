#!/bin/bash
# SQ counter passes over tools/conv_one.py:  tools/pmc_roll.sh "B,H,W,Ci,Co" kind tag   (both kernels: IM2IM_CONV_ROLL=0 and 1)
shape=${1:-78,320,320,64,64}; kind=${2:-dgrad}; tag=${3:-pmcroll}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for roll in 0 1; do
  export IM2IM_CONV_ROLL=$roll
  python $root/tools/conv_one.py $shape $kind 10
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_r$roll/p$i -o pmc -- python $root/tools/conv_one.py $shape $kind 3 > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, collections
for roll in (0, 1):
    d = collections.defaultdict(list)
    for f in glob.glob("$root/gpurun_out/${tag}_r%d/p*/**/*counter_collection.csv" % roll, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_igemm" in k or "conv_roll64" in k:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {n: sum(v) / len(v) for n, v in d.items()}
    print("== conv_roll =", roll, "$shape $kind")
    for n in sorted(c):
        print(f"  {n:28s} {c[n]:16.0f}  (n={len(d[n])})")
    if "GRBM_GUI_ACTIVE" in c:
        el = c["GRBM_GUI_ACTIVE"] / 8
        print(f"  -> elapsed cycles {el:.0f}  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (el * 1024):.3f}  wave-cycles(quad)/MFMA {c['SQ_WAVE_CYCLES'] / c.get('SQ_INSTS_MFMA', 1):.1f}"
              f"  wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}  wait_inst_any {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
PY
