#!/bin/bash
# PMC passes over tools/bench_conv.py for one conv shape:  tools/pmc_conv.sh "B,H,W,Ci,Co" [lib-suffix]
# (run on the GPU box through gpurun; counters only with --kernel-trace, one group per pass)
shape=${1:-78,160,160,128,128}; suf=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
export IM2IM_LIB=$root/im2im_uq_amd/lib/libim2im_uq$suf.so
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/gpurun_out/pmc_conv$suf/p$i -o pmc -- python $root/tools/bench_conv.py $shape 5 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$root/gpurun_out/pmc_conv$suf/p*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm" in k or "conv_wgrad_pipe" in k:
            d["igemm" if "igemm" in k else "wgrad"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    print("==", k)
    for n, v in sorted(c.items()):
        print(f"  {n:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
