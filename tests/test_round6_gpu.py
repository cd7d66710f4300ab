"""Round 6 (GPU):
  * host datasets through the two-deep pinned prefetcher (im2im_uq_amd/prefetch.py): the batches, train_net's losses / weights and
    calibrate_model's table / lambda-hat are bit-identical to the in-line upload loop of the reference
    (core/scripts/train.py:147-149, core/calibration/calibrate_model.py:118-123);
  * the reference's real 8-way split, functionally: 8 ranks over gloo sharing the box's one GPU -- global batch 78 ->
    10,10,10,10,10,10,9,9 through GradSync and through train_net, the 3,474-row calibration table in contiguous shards -> identical
    lambda-hat and [N, L] row order, eval_set_metrics' miss-map all-reduce; `bench.py --gpus 8` over gloo
    (core/scripts/train.py:22-27,112-115, experiments/fastmri_test/config.yml:44-45).
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader, TensorDataset

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=50, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              device=DEV, dataset="synthetic", batch_size=8, lr=1e-3, input_normalization="standard",
              output_normalization="min-max", num_validation_images=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------------ prefetcher
@pytest.mark.parametrize("thread", [True, False])
def test_prefetcher_yields_the_loaders_batches_bit_for_bit(thread):
    """every batch of a shuffled DataLoader (short last batch, a uint8 and a float64 member, a pinned member) arrives on the device
    with the host values, in the loader's order; non-tensor leaves and None pass through; the ring is reused (more batches than
    slots); abandoning the iteration early leaves no thread behind."""
    import threading
    from im2im_uq_amd.prefetch import DevicePrefetcher
    g = torch.Generator().manual_seed(3)
    n = 37
    x = torch.randn(n, 2, 24, 20, generator=g)
    y = torch.rand(n, 1, 24, 20, generator=g).double()
    z = torch.randint(0, 255, (n, 5), generator=g, dtype=torch.uint8)
    ds = TensorDataset(x, y, z)
    torch.manual_seed(11)
    want = [tuple(t.clone() for t in b) for b in DataLoader(ds, batch_size=5, shuffle=True, num_workers=0)]
    torch.manual_seed(11)
    got = []
    for b in DevicePrefetcher(DataLoader(ds, batch_size=5, shuffle=True, num_workers=0), DEV, thread=thread):
        assert all(t.is_cuda for t in b)
        got.append(tuple(t.cpu() for t in b))                 # the views are only valid until the next batch is requested
    assert len(got) == len(want) == 8 and got[-1][0].shape[0] == 2
    for a, b in zip(got, want):
        for s, t in zip(a, b):
            assert s.dtype == t.dtype and torch.equal(s, t)
    # nested structure with pass-through leaves (what train_net's multi-rank generator yields) and a pinned source
    items = [([x[:3], y[:3]], 78), (None, 5), ([x[3:4].pin_memory(), y[3:4]], 1)]
    out = list(DevicePrefetcher(iter(items), DEV, thread=thread))
    assert out[1] == (None, 5) and out[0][1] == 78 and out[2][1] == 1
    assert torch.equal(out[2][0][0].cpu(), x[3:4]) and torch.equal(out[2][0][1].cpu(), y[3:4])
    # early exit
    before = threading.active_count()
    it = iter(DevicePrefetcher(DataLoader(ds, batch_size=2, num_workers=0), DEV, thread=thread))
    first = next(it)
    assert torch.equal(first[0].cpu(), x[:2])
    it.close()
    time.sleep(0.2)
    assert threading.active_count() <= before
    # an exception inside the dataset surfaces in the consumer
    class Bad(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            if i == 3:
                raise RuntimeError("broken sample")
            return x[i], y[i]
    with pytest.raises(RuntimeError, match="broken sample"):
        list(DevicePrefetcher(DataLoader(Bad(), batch_size=1, num_workers=0), DEV, thread=thread))


def _train_once(prefetch_on, thread, dtype, ds, val, graph=False):
    from im2im_uq_amd import nn_ops, prefetch
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts import train as tr
    nn_ops.set_compute_dtype(dtype)
    was = prefetch.ENABLED, prefetch.THREAD
    prefetch.ENABLED, prefetch.THREAD = prefetch_on, thread
    logged = []
    orig_log = tr.wandb.log
    tr.wandb.log = lambda d, *a, **k: logged.append(dict(d))
    try:
        torch.manual_seed(5)
        net = add_uncertainty(UNet(1, 1, depth=2, base=32), dict(PARAMS))
        torch.manual_seed(6)                                  # the DataLoader's shuffle
        cfg = dict(PARAMS, batch_size=6, hip_graph=graph)
        net = tr.train_net(net, ds, val, DEV, 2, 6, 1e-3, False, None, 100, 100, cfg)
        torch.cuda.synchronize()
        return ([d["train_loss"] for d in logged if "train_loss" in d], [d["val_loss"] for d in logged if "val_loss" in d],
                {k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    finally:
        tr.wandb.log = orig_log
        prefetch.ENABLED, prefetch.THREAD = was
        nn_ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("dtype,graph", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_train_net_with_prefetcher_is_bit_identical_to_the_inline_upload_loop(dtype, graph):
    """train_net on a HOST TensorDataset (the reference's data contract, train.py:104,147-149): two epochs of shuffled batches (the
    last one short) + validation; epoch losses, validation losses and every state_dict tensor are the in-line loop's bits, with
    the producer thread and without -- eager steps and the HIP-graph step (which copies the ring's views into its static inputs)."""
    g = torch.Generator().manual_seed(8)
    x, y = torch.randn(20, 1, 32, 32, generator=g), torch.rand(20, 1, 32, 32, generator=g)
    ds, val = TensorDataset(x, y), TensorDataset(x[:4], y[:4])
    ref = _train_once(False, False, dtype, ds, val, graph)
    assert len(ref[0]) == 2 and all(np.isfinite(ref[0]))
    for thread in (True, False):
        got = _train_once(True, thread, dtype, ds, val, graph)
        assert got[0] == ref[0] and got[1] == ref[1], (got[:2], ref[:2])
        for k in ref[2]:
            assert torch.equal(got[2][k], ref[2][k]), k


def test_calibrate_model_and_loss_table_from_a_host_dataset_match_the_inline_loop_and_the_resident_path():
    """calibrate_model / get_loss_table / eval_net over a host dataset: prefetched == in-line == HBM-resident dataset, bit for bit
    (table, lambda-hat, validation loss)."""
    from im2im_uq_amd import nn_ops, prefetch
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.eval import eval_net, get_loss_table
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(2)
    model = add_uncertainty(UNet(1, 1, depth=2, base=32), dict(PARAMS)).to(DEV)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(45, 1, 32, 32, generator=g)
    with torch.no_grad():
        model.train(); model(x[:8].to(DEV)); model.eval()
        o = model(x.to(DEV)).cpu()
    z = torch.randn(o[:, 1].shape, generator=g)
    y = o[:, 1] + 1.5 * z * torch.where(z > 0, (o[:, 2] - o[:, 1]).clamp_min(1e-6), (o[:, 1] - o[:, 0]).clamp_min(1e-6))
    cfg = dict(PARAMS, batch_size=7)
    res = {}
    was = prefetch.ENABLED, prefetch.THREAD
    try:
        for key, ds, on, thread in (("inline", TensorDataset(x, y), False, False), ("thread", TensorDataset(x, y), True, True),
                                    ("nothread", TensorDataset(x, y), True, False), ("resident", TensorDataset(x.to(DEV), y.to(DEV)), True, True)):
            prefetch.ENABLED, prefetch.THREAD = on, thread
            model.lhat = None
            m, table = calibrate_model(model, ds, dict(cfg))
            full = get_loss_table(m, ds, dict(cfg))
            vl = eval_net(m, DataLoader(ds, batch_size=7, num_workers=0), DEV) if key != "resident" else None
            m.eval()
            res[key] = (float(m.lhat), table.clone(), full.clone(), vl)
    finally:
        prefetch.ENABLED, prefetch.THREAD = was
    ref = res["inline"]
    assert 0.0 < ref[0] < 6.0 and float(ref[1].sum()) > 0
    for key in ("thread", "nothread", "resident"):
        assert res[key][0] == ref[0] and torch.equal(res[key][1], ref[1]) and torch.equal(res[key][2], ref[2]), key
        if res[key][3] is not None:
            assert res[key][3] == ref[3]


# ------------------------------------------------------------------------------------------------ the reference's 8-way split
N_CAL, HW = 3474, 32


def _small_model(seed=4):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("fp32")
    torch.manual_seed(seed)
    return add_uncertainty(UNet(1, 1, depth=2, base=32), dict(PARAMS, batch_size=78)).to(DEV)


def _train_data():
    g = torch.Generator().manual_seed(31)
    return torch.randn(156, 1, HW, HW, generator=g), torch.rand(156, 1, HW, HW, generator=g)


def _calib_data(model):
    """3,474 images whose labels make the scan stop mid-grid (built from the model's own eval outputs, as bench.py does)"""
    g = torch.Generator().manual_seed(32)
    x = torch.randn(N_CAL, 1, HW, HW, generator=g)
    with torch.no_grad():
        model.eval()
        o = torch.cat([model(x[s:s + 512].to(DEV)).cpu() for s in range(0, N_CAL, 512)])
    z = torch.randn(o[:, 1].shape, generator=g)
    y = o[:, 1] + 1.5 * z * torch.where(z > 0, (o[:, 2] - o[:, 1]).clamp_min(1e-6), (o[:, 1] - o[:, 0]).clamp_min(1e-6))
    return x, y


def _worker8(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from im2im_uq_amd import nn_ops
        from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model, shard_bounds
        from im2im_uq_amd.core.scripts.eval import eval_set_metrics, get_loss_table
        from im2im_uq_amd.core.scripts.train import GlobalBatchSampler, GradSync, broadcast_module_state, train_net
        x, y = _train_data()
        # (1) one global batch of 78 through GradSync: this rank's share of the reference's batch, weighted n_r / 78, summed over ranks
        model = _small_model().train()
        broadcast_module_state(model)
        sync = GradSync(model.parameters(), bucket_bytes=64 << 10)           # several buckets even for this small net
        lo, hi = GlobalBatchSampler.share(78, rank, world)
        sync.zero_grad()
        loss = model.loss_fn(model(x[lo:hi].to(DEV)), y[lo:hi].to(DEV))
        (loss * ((hi - lo) / 78)).backward()
        sync.finish()
        torch.save({"share": (lo, hi), "loss": float(loss.detach()), "buckets": len(sync.buckets),
                    "grads": {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()} if rank in (0, 7) else None},
                   os.path.join(tmpdir, f"grad_{rank}.pt"))
        # (2) train_net: 2 steps (156 images, global batch 78, one epoch), every rank ends on the same weights
        net = _small_model()
        cfg = dict(PARAMS, batch_size=78, hip_graph=False)
        net = train_net(net, TensorDataset(x, y), TensorDataset(x[:4], y[:4]), DEV, 1, 78, 1e-3, False, None, 100, 100, cfg)
        flat = torch.cat([t.detach().flatten().float() for t in net.state_dict().values()])
        every = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(every, flat)
        same = all(torch.equal(every[0], e) for e in every)
        # (3) the 3,474-image calibration set in contiguous shards: table rows in dataset order, identical lambda-hat on every rank
        xc, yc = _calib_data(net)
        ds = TensorDataset(xc, yc)
        net, table = calibrate_model(net, ds, dict(cfg, batch_size=64))
        full = get_loss_table(net, ds, dict(cfg, batch_size=64))
        torch.manual_seed(0); np.random.seed(0)
        risk, sizes, spearman, strat, mse, spatial = eval_set_metrics(net, TensorDataset(xc[:500], yc[:500]), dict(cfg, batch_size=64))
        torch.save({"same_weights": same, "lhat": float(net.lhat), "shard": shard_bounds(N_CAL, rank, world),
                    "table": table if rank in (0, 5) else None, "table_sum": float(table.double().sum()), "full_sum": float(full.double().sum()),
                    "full": full if rank == 0 else None, "risk": float(risk), "spatial": spatial, "sizes": sizes, "mse": mse,
                    "state": {k: v.cpu() for k, v in net.state_dict().items()} if rank == 0 else None},
                   os.path.join(tmpdir, f"run_{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_run_the_references_split_of_batch_78_and_of_the_3474_image_calibration_set(tmp_path):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.scripts.eval import eval_set_metrics, get_loss_table
    from im2im_uq_amd.core.scripts.train import GlobalBatchSampler
    # Eight processes time-sharing ONE GPU (production is one process per GPU) now and then lose a rank to a queue abort of the
    # runtime -- HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in a random rank at a random point, in about one run of three on the test boxes,
    # equally with the prefetcher, the one-launch BatchNorm sums and the deferred weight-gradient reduction switched off (the
    # two-rank tests of rounds 2-5 never showed it).  A run that loses a rank that way proves nothing either way and is repeated.
    # The rate depends on the box: 0 aborts in 15 runs on one, 4 of 4 on another (profiles/r06_ab_experiments.txt section 4).  The workers
    # get two hardware queues each (GPU_MAX_HW_QUEUES: 8 x 4 streams would oversubscribe the queues the scheduler can keep mapped);
    # a box that still aborts every attempt cannot run this configuration at all, which is reported as a skip, not as a parity failure.
    was = os.environ.get("GPU_MAX_HW_QUEUES")
    os.environ["GPU_MAX_HW_QUEUES"] = "2"
    try:
        for attempt in range(4):
            try:
                mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
                break
            except mp.ProcessExitedException as e:
                if getattr(e, "signal_name", None) != "SIGABRT":
                    raise
                if attempt == 3:
                    pytest.skip("this box's runtime aborts a queue (HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION) whenever 8 processes share its GPU: 4 of 4 attempts")
                time.sleep(2.0)
    finally:
        if was is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = was
    g = [torch.load(tmp_path / f"grad_{r}.pt", weights_only=False) for r in range(8)]
    shares = [r["share"] for r in g]
    assert [hi - lo for lo, hi in shares] == [10, 10, 10, 10, 10, 10, 9, 9]                  # the reference's DataParallel scatter of 78
    assert shares[0][0] == 0 and shares[-1][1] == 78 and all(shares[i][1] == shares[i + 1][0] for i in range(7))
    assert g[0]["buckets"] > 1
    for k in g[0]["grads"]:
        assert torch.equal(g[0]["grads"][k], g[7]["grads"][k])                             # every rank holds the reduced gradient
    # the same eight replicas run one after the other by ONE process on the same weights: autograd accumulates the weighted gradients
    # in rank order; gloo's ring adds the eight addends in another order, so agreement is to fp32 summation noise, not bits
    x, y = _train_data()
    try:
        seq = _small_model().train()
        bufs = {k: v.clone() for k, v in seq.named_buffers()}
        losses = []
        for lo, hi in shares:
            l = seq.loss_fn(seq(x[lo:hi].to(DEV)), y[lo:hi].to(DEV))
            (l * ((hi - lo) / 78)).backward()
            losses.append(float(l.detach()))
            with torch.no_grad():
                for k, v in seq.named_buffers():
                    v.copy_(bufs[k])
        assert losses == [r["loss"] for r in g]
        for k, p in seq.named_parameters():
            if p.grad is None:
                assert float(g[0]["grads"][k].abs().max()) == 0.0
                continue
            a, b = g[0]["grads"][k].double(), p.grad.cpu().double()
            assert float((a - b).norm()) <= 2e-6 * float(b.norm()) + 1e-12, k
        # (2) / (3): identical weights, identical lambda-hat, rows in dataset order == ONE process on the whole set
        runs = [torch.load(tmp_path / f"run_{r}.pt", weights_only=False) for r in range(8)]
        assert all(r["same_weights"] for r in runs)
        assert [r["shard"][1] - r["shard"][0] for r in runs] == [435] * 7 + [429] and runs[-1]["shard"][1] == N_CAL
        assert len({r["lhat"] for r in runs}) == 1 and len({r["table_sum"] for r in runs}) == 1 and len({r["full_sum"] for r in runs}) == 1
        assert torch.equal(runs[0]["table"], runs[5]["table"]) and tuple(runs[0]["table"].shape) == (N_CAL, 50)
        assert len({r["risk"] for r in runs}) == 1 and all(np.array_equal(r["spatial"], runs[0]["spatial"]) for r in runs)
        single = _small_model()
        single.load_state_dict({k: v for k, v in runs[0]["state"].items() if k != "lhat"})
        xc, yc = _calib_data(single)
        cfg = dict(PARAMS, batch_size=64)
        single, table = calibrate_model(single, TensorDataset(xc, yc), dict(cfg))
        full = get_loss_table(single, TensorDataset(xc, yc), dict(cfg))
        torch.manual_seed(0); np.random.seed(0)
        risk, sizes, spearman, strat, mse, spatial = eval_set_metrics(single, TensorDataset(xc[:500], yc[:500]), dict(cfg))
        assert 0.0 < float(single.lhat) < 6.0                                              # the scan stopped inside the grid
        assert float(single.lhat) == runs[0]["lhat"] and torch.equal(table, runs[0]["table"]) and torch.equal(full, runs[0]["full"])
        assert float(risk) == runs[0]["risk"] and np.array_equal(spatial, runs[0]["spatial"]) and torch.equal(sizes, runs[0]["sizes"])
        assert mse == runs[0]["mse"]
    finally:
        nn_ops.set_compute_dtype("bf16")


def test_bench_gpus_8_over_gloo_on_one_gpu():
    """`python bench.py --gpus 8 --steps 2` exactly as the driver types it (plus a small image size): eight ranks spawned by the
    script, gloo, one shared GPU; the line carries the world, the strong-scaling companion with the reference's 10/9 split and the
    exchange record -- inside two minutes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["IM2IM_DIST_BACKEND"] = "gloo"
    env["GPU_MAX_HW_QUEUES"] = "2"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--size", "64",
           "--calib-images", "16", "--no-roofline", "--no-cpu-baseline"]
    for attempt in range(3):                                 # (a rank lost to the shared-GPU queue abort described above: run again)
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        dt = time.time() - t0
        if r.returncode == 0 or "ILLEGAL_INSTRUCTION" not in r.stderr:
            break
    if r.returncode != 0 and "ILLEGAL_INSTRUCTION" in r.stderr:
        pytest.skip("this box's runtime aborts a queue whenever 8 processes share its GPU: 3 of 3 attempts")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["distributed"]["world_size"] == 8 and d["distributed"]["backend"] == "gloo"
    assert d["distinct_devices"] == 1 and [x["rank"] for x in d["distributed"]["ranks"]] == list(range(8))
    assert d["config"]["global_batch"] == 78 * 8 and d["config"]["per_gpu_batch"] == 78 and d["scaling"] == "weak"
    assert d["strong"]["global_batch"] == 78 and d["strong"]["per_gpu_batch_rank0"] == 10 and d["strong_ms_per_step"] > 0
    assert d["value"] > 0 and d["calib"]["value"] > 0 and d["allreduce_ms"] > 0
    assert dt < 120.0, dt
