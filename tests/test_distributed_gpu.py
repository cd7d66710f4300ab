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


def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver types it -- no torch.distributed.run around it: the script spawns the
    two ranks itself (they share the one GPU over gloo here; on an N-GPU node the default backend is nccl = RCCL) and the
    line proves it: n_gpus, world_size, the all-reduced bytes, and the strong-scaling companion."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["IM2IM_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "5",
           "--size", "64", "--calib-images", "8", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["world_size"] == 2 and d["distributed"]["backend"] == "gloo"
    assert d["distributed"]["allreduce_bytes_per_step"] == 17269123 * 4
    assert d["config"]["global_batch"] == 10 and d["scaling"] == "weak"
    assert d["strong"]["global_batch"] == 5 and d["strong"]["per_gpu_batch_rank0"] == 3 and d["strong"]["value"] > 0
    # [r4] the line says which device every rank ran on, times one standalone all-reduce of the gradient buffer and the part of
    # the exchange a step does not hide (here: two ranks on ONE device over gloo -- distinct_devices says so)
    ds = d["distributed"]
    assert [r["rank"] for r in ds["ranks"]] == [0, 1] and ds["ranks"][0]["pid"] != ds["ranks"][1]["pid"]
    assert ds["distinct_devices"] == 1 and all(r["local_device_index"] == 0 for r in ds["ranks"])
    assert ds["allreduce_ms"] > 0 and ds["allreduce_busbw_gbs"] > 0
    assert ds["step_ms_with_grad_exchange"] > 0 and ds["step_ms_without_grad_exchange"] > 0 and "grad_exchange_exposed_ms" in ds
    # [r5] ... and repeats the scaling run's first-look numbers at the top level
    assert d["strong_ms_per_step"] == d["strong"]["ms_per_step"] and d["strong_imgs_per_s"] == d["strong"]["value"]
    assert d["grad_exchange_exposed_ms"] == ds["grad_exchange_exposed_ms"] and d["allreduce_ms"] == ds["allreduce_ms"]
    assert d["distinct_devices"] == 1
    assert "other_configs" not in d and "fastmri_pipeline" not in d and d["fp32"] is None and d["cpu_baseline"] is None   # N > 1 runs only the scaling legs


def test_bench_gpus_2_over_rccl_needs_two_gpus():
    """default backend (nccl = RCCL): two ranks on a one-GPU box must be a non-zero exit that says why, never n_gpus 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs: the RCCL path itself runs (driver's scaling bench)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IM2IM_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode != 0 and "need 2 GPUs" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _verify_worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as dist
    from im2im_uq_amd import launch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    out = {}
    try:
        ranks = launch.verify_world(dist, rank, world, dev, "gloo")          # functional-test backend: a shared GPU is allowed
        out["ok"] = [r["rank"] for r in ranks] == [0, 1] and launch.distinct_devices(ranks) == 1
        try:
            launch.verify_world(dist, rank, world, dev, "nccl")              # the measured backend's rule on the same (shared-GPU) world
            out["refused"] = None
        except SystemExit as e:
            out["refused"] = str(e)
    finally:
        dist.destroy_process_group()
    with open(os.path.join(tmpdir, f"verify_{rank}.json"), "w") as f:
        json.dump(out, f)


def test_verify_world_refuses_n_rccl_ranks_on_fewer_devices(tmp_path):
    """[r5] launch.verify_world (called by init_distributed right after the rendezvous, so by `bench.py --gpus N` before any leg):
    every rank learns every rank's device over the job's own process group; two ranks that would be RCCL ranks on ONE device are a
    SystemExit naming the devices on BOTH ranks, the rank count is checked by a one-element all-reduce.  (Two ranks, gloo, one GPU.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_verify_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        out = json.load(open(tmp_path / f"verify_{rank}.json"))
        assert out["ok"] is True
        assert out["refused"] and "2 RCCL ranks sit on 1 distinct device" in out["refused"]
