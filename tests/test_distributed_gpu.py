"""The N > 1 code path of bench.py (one process per rank: rendezvous, flat gradient all-reduce, sharded calibration with
the gathered loss table, max-over-ranks timing) on real GPU tensors.  A test box has one GPU and RCCL refuses two ranks on
the same device, so the two ranks share it through the gloo backend (IM2IM_DIST_BACKEND, see bench.py); the measured
configuration is nccl (= RCCL over xGMI), one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, IM2IM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--size", "64", "--calib-images", "8", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["calib"]["value"] > 0 and d["scaling"] == "weak"


def test_bench_strong_scaling_splits_the_global_batch_unevenly():
    """--scaling strong: the GLOBAL batch (here 5 = 3 + 2 over two ranks) and the calibration split are divided over the
    ranks; gradients are summed with (local / global) loss weights."""
    env = dict(os.environ, IM2IM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "5",
           "--size", "64", "--calib-images", "8", "--no-roofline", "--no-cpu-baseline", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 5 and d["config"]["per_gpu_batch"] == 3 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["calib"]["value"] > 0
