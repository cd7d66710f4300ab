"""gpurun_out/<tag>_* (tools/profile_round.sh) -> profiles/<tag>_*: kernel-stats CSVs, bench lines, pmc_traffic.json."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


for leg in ("train", "calib", "train_isolated", "fp32", "temca1024", "bsbcm512"):
    src = first(os.path.join(go, f"{tag}_prof_{leg}", "**", "*kernel_stats.csv"))
    if src:
        shutil.copy(src, os.path.join(pr, f"{tag}_{leg}_kernel_stats.csv"))
    for name in (f"{tag}_{leg}_leg_bench_line.json",):
        p = os.path.join(go, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(pr, name))
p = os.path.join(go, f"{tag}_bench_line.json")
if os.path.exists(p) and os.path.getsize(p):
    shutil.copy(p, os.path.join(pr, f"{tag}_bench_line.json"))


def pmc_mean(leg, counter, match):
    f = first(os.path.join(go, f"{tag}_pmc_{leg}_{counter}", "**", "*counter_collection.csv"))
    if not f:
        return None, 0
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals) if vals else None), len(vals)


rec = {}
fk, n = pmc_mean("calib", "FETCH_SIZE", "rcps_hist_kernel")
wk, _ = pmc_mean("calib", "WRITE_SIZE", "rcps_hist_kernel")
line = json.load(open(os.path.join(go, f"{tag}_calib_leg_bench_line.json"))) if os.path.exists(os.path.join(go, f"{tag}_calib_leg_bench_line.json")) else None
if fk is not None and wk is not None:
    rec["rcps_hist_kernel"] = {"images": line["calib"]["images_per_gpu"] if line else None, "hw": 320, "fetch_kb_raw": fk, "write_kb": wk,
                               "launches_profiled": n, "traffic_bytes_per_launch": (2 * fk + wk) * 1024,
                               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream); separate --pmc "
                                       "passes for FETCH_SIZE and WRITE_SIZE over `bench.py --legs calib`"}
fk, n = pmc_mean("train", "FETCH_SIZE", "conv_igemm_kernel")
wk, _ = pmc_mean("train", "WRITE_SIZE", "conv_igemm_kernel")
if fk is not None and wk is not None:
    rec["conv_igemm_kernel"] = {"per_gpu_batch": 78, "hw": 320, "launches_profiled": n, "traffic_bytes_per_launch": (2 * fk + wk) * 1024,
                                "note": "launch-weighted mean over all conv_igemm variants of 2*FETCH_SIZE + WRITE_SIZE (separate --pmc passes over "
                                        "`bench.py --legs train --steps 2 --warmup 1`); FETCH doubled as for wide streaming reads -- an upper estimate "
                                        "for the halo gathers"}
if rec:
    json.dump(rec, open(os.path.join(pr, "pmc_traffic.json"), "w"), indent=1)
    json.dump(rec, open(os.path.join(pr, f"{tag}_pmc_fetch_write.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
