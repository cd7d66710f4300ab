cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
IM2IM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_prof -o q -- python bench.py --legs train --no-fp32 --no-roofline --no-extras --no-live-pmc --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
find gpurun_out/q_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/q_kernel_stats.csv \;
rm -rf gpurun_out/q_prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/q_kernel_stats.csv')))
n=[int(r['Calls']) for r in rows if 'smallconv_s2l_kernel' in r['Name']][0]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    if any(k in r['Name'] for k in ('pool_bwd','bn_relu_bwd','maxpool','up2x','smallconv','reduce_rows','bn_')):
        print(f"{float(r['TotalDurationNs'])/n/1e6:7.3f} ms {int(r['Calls'])/n:5.1f}x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
