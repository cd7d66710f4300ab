#!/bin/bash
# per-GPU batch 10 train step under single runtime environment settings, two rounds interleaved with the baseline
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
run() { env "$@" python bench.py --legs train --batch 10 --no-fp32 --no-roofline --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms  host', round(d['host_enqueue_ms_per_step'],3))" 2>/dev/null || echo FAILED; }
for rep in 1 2; do
  for v in X=1 HSA_NO_SCRATCH_RECLAIM=1 HSA_ALLOCATE_QUEUE_DEV_MEM=1 ROC_SKIP_KERNEL_ARG_COPY=1 DEBUG_HIP_KERNARG_COPY_OPT=1 ROC_USE_FGS_KERNARG=0 GPU_STREAMOPS_CP_WAIT=1 HSA_ENABLE_MWAITX=1 AMD_CPU_AFFINITY=1 ROC_ACTIVE_WAIT_TIMEOUT=100; do
    echo "$v: $(run $v)"
  done
done
