"""N > 1 logic on CPU: two gloo ranks exercise the same host-side code the RCCL path runs (contiguous calibration
shards, ragged row all-gather, identical lambda-hat on every rank, flat gradient all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from im2im_uq_amd.core.calibration.calibrate_model import gather_rows, scan_loss_table, shard_bounds
        from im2im_uq_amd.core.scripts.train import allreduce_gradients
        from oracle import calibration as oc
        cfg = dict(alpha=0.1, delta=0.1, num_lambdas=40, minimum_lambda=0.0, maximum_lambda=8.0)
        out, lab = oc.synth_outputs(n_total, 1, 12, 12, seed=5)
        lambdas = oc.lambda_grid(cfg)
        dl = lambdas[1] - lambdas[0]
        lo, hi = shard_bounds(n_total, rank, world)
        # this rank's rows of the shifted-lambda table (the HIP kernel's job on the GPU; oracle stands in on CPU)
        local = torch.stack([oc.losses_at(out[lo:hi], lab[lo:hi], lam - dl) for lam in lambdas], dim=1) if hi > lo \
            else torch.zeros((0, len(lambdas)))
        full = gather_rows(local, n_total)
        lhat, table, trace = scan_loss_table(full, lambdas, cfg["alpha"], cfg["delta"])
        ref_lhat, ref_table, _ = oc.calibrate_from_outputs(out, lab, cfg)
        assert full.shape == (n_total, len(lambdas))
        assert torch.equal(table, ref_table) and float(lhat) == float(ref_lhat)
        # one-shot gradient averaging (callers that keep their own .grad tensors)
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
        for i, p in enumerate(ps):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        allreduce_gradients(ps)
        mean_rank = sum(r + 1 for r in range(world)) / world
        for i, p in enumerate(ps):
            assert torch.allclose(p.grad, torch.full_like(p, mean_rank * (i + 1)))
        np.save(os.path.join(tmpdir, f"lhat_{rank}.npy"), np.array([float(lhat)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [16, 7, 1])   # 7 over 2 ranks: ragged shards (4 + 3); 1: rank 1's shard is empty
def test_two_rank_gloo_calibration_and_grad_allreduce(tmp_path, n_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    vals = [np.load(tmp_path / f"lhat_{r}.npy")[0] for r in range(world)]
    assert vals[0] == vals[1]


def test_shard_bounds_cover_exactly():
    from im2im_uq_amd.core.calibration.calibrate_model import shard_bounds
    for n in (0, 1, 7, 8, 3474):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _gradsync_worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from im2im_uq_amd.core.scripts.train import GlobalBatchSampler, GradSync, broadcast_module_state
        torch.manual_seed(100 + rank)                        # ranks start from DIFFERENT weights on purpose
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                  torch.nn.Linear(16, 3))
        broadcast_module_state(net)                          # ... and must end up with rank 0's
        ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                  torch.nn.Linear(16, 3))
        torch.manual_seed(100)
        ref0 = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                   torch.nn.Linear(16, 3))
        for a, b in zip(net.parameters(), ref0.parameters()):
            assert torch.equal(a, b)
        ref.load_state_dict(ref0.state_dict())
        sync = GradSync(net.parameters(), bucket_bytes=256)  # several buckets
        assert len(sync.buckets) >= 3
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        opt_ref = torch.optim.SGD(ref.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(7)
        data_x, data_y = torch.randn(23, 6, generator=g), torch.randn(23, 3, generator=g)
        sampler = GlobalBatchSampler(23, 5, rank, world, shuffle=True, seed=3)       # 5 = 3 + 2; last batch of 3 = 2 + 1
        full = GlobalBatchSampler(23, 5, 0, 1, shuffle=True, seed=3)
        for idx, idx_all in zip(sampler, full):
            n_glob = len(idx_all)
            sync.zero_grad()
            if idx:
                loss = torch.nn.functional.mse_loss(net(data_x[idx]), data_y[idx])
                (loss * (len(idx) / n_glob)).backward()
            sync.finish()
            opt.step()
            opt_ref.zero_grad()
            torch.nn.functional.mse_loss(ref(data_x[idx_all]), data_y[idx_all]).backward()   # DataParallel: loss of the gathered batch
            for a, b in zip(net.parameters(), ref.parameters()):
                assert a.grad.data_ptr() >= sync.flat.data_ptr()             # still a view of the flat buffer
                assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
            opt_ref.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        assert sync.expected is not None and all(e > 0 for e in sync.expected)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradsync_equals_global_batch_gradient(tmp_path):
    """GradSync (flat buffer, bucketed async all-reduce from backward hooks) + GlobalBatchSampler (uneven split of the
    global batch) + loss weighting == the gradient of the global-batch mean loss, i.e. what the reference's
    DataParallel computes on the gathered output (train.py:112-115,152-160); different initial weights per rank are
    overwritten by rank 0's."""
    mp.spawn(_gradsync_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def test_global_batch_sampler_partitions_every_batch():
    from im2im_uq_amd.core.scripts.train import GlobalBatchSampler
    for n, bs, world in ((78 * 3 + 5, 78, 8), (10, 4, 4), (3, 8, 8)):
        per_rank = [list(GlobalBatchSampler(n, bs, r, world, shuffle=True, seed=1)) for r in range(world)]
        whole = list(GlobalBatchSampler(n, bs, 0, 1, shuffle=True, seed=1))
        assert sorted(i for b in whole for i in b) == list(range(n))
        for k, b in enumerate(whole):
            parts = [per_rank[r][k] for r in range(world)]
            assert [i for part in parts for i in part] == b
            sizes = [len(part) for part in parts]
            assert max(sizes) - min(sizes) <= 1
        if n >= 78:
            assert [len(per_rank[r][0]) for r in range(8)] == [10, 10, 10, 10, 10, 10, 9, 9]
