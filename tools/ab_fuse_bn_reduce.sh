cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
IM2IM_FUSE_BN_REDUCE=$v python bench.py --legs train --no-fp32 --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_bn_reduce $v', round(d['value'],1), round(d['ms_per_step'],3))"
done
for v in 0 1; do
IM2IM_WGRAD_STREAM=$v python bench.py --legs train --no-fp32 --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_stream $v', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
