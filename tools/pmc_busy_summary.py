"""gpurun_out/<tag>_pmc_sq/p*_<shape>/ (tools/profile_round4.sh) -> a counter table on stdout and profiles/<tag>_pmc_mfma_busy.json:
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per conv kernel and layer shape -- utilisation of the MFMA pipe
by the hardware's own counter at the clock the chip actually ran (bench.py copies the conv_igemm values into
roofline.mfma_busy_cycle_frac_pmc), plus LDS instruction / bank-conflict counts per launch."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_sq", "p*", "**", "*counter_collection.csv"), recursive=True):
    shape = re.search(r"p\d+_([\d,]+)", f).group(1)
    b, h, w, ci, co = shape.split(",")
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "conv_igemm" if "conv_igemm" in k else "conv_wgrad_roll" if "conv_wgrad_roll" in k else "conv_wgrad_pipe" if "conv_wgrad_pipe" in k else None
        if fam:
            d[f"{fam} {ci}->{co} @{h}x{w} B{b}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc passes of tools/profile_round4.sh over tools/bench_conv.py (--kernel-trace only, one counter group per pass): MFMA "
               "pipe busy cycles over all SIMDs / (elapsed GPU cycles x 1024 SIMDs). Counter-based utilisation at the clock the chip "
               "actually ran, NOT the FLOP fraction of the nominal-clock peak (roofline.frac)."}
for k in sorted(d):
    c = {n: sum(v) / len(v) for n, v in d[k].items()}
    print("==", k)
    for n in sorted(c):
        print(f"  {n:28s} {c[n]:16.0f}  (n={len(d[k][n])})")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        elapsed = c["GRBM_GUI_ACTIVE"] / 8.0
        rec = {"SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE_sum_over_8_xcds": c["GRBM_GUI_ACTIVE"],
               "SQ_INSTS_MFMA": c.get("SQ_INSTS_MFMA"), "elapsed_cycles": elapsed, "simd_cycles": elapsed * 1024,
               "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (elapsed * 1024)}
        for n in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if n in c:
                rec[n] = c[n]
        out[k] = rec
        print(f"  -> mfma_busy_frac {rec['mfma_busy_frac']:.3f}")
if len(out) > 1:
    json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_mfma_busy.json"), "w"), indent=1)
