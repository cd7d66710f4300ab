"""GraphedStep under data parallelism (core/scripts/train.py; the reference's loop train.py:141-165 under nn.DataParallel
:112-115) and the capturable optimizer:
  * two ranks over gloo (sharing the box's one GPU): the graphed step (forward + backward + bucket packing replayed, all-reduce
    issued after the replay, FusedAdam) ends in bit-identical parameters / buffers / losses as the eager GradSync loop;
  * ONE rank over the nccl backend -- real RCCL calls on this box: init, broadcast, all_reduce from the weight-gradient stream,
    and (IM2IM_GRAPH_COLLECTIVES=1) the all-reduces CAPTURED in the HIP graph with FusedAdam behind them;
  * the graph is not taken for a network with hooks, and a failing capture falls back to the eager loop with the
    optimizer's counters intact."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(n=9, b=6):
    g = torch.Generator().manual_seed(31)
    return [(torch.randn(b, 1, 32, 32, generator=g), torch.rand(b, 1, 32, 32, generator=g)) for _ in range(n)]


def _run(rank, world, graph, force_dist=False):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts import train as T
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(3)
    model = add_uncertainty(UNet(1, 1, depth=2), dict(PARAMS)).to(DEV).train()
    T.broadcast_module_state(model)
    opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
    sync = T.GradSync(model.parameters())
    gs = T.GraphedStep(model, opt, sync) if graph else None
    losses = []
    for x, y in _batches():
        lo, hi = T.GlobalBatchSampler.share(x.shape[0], rank, world)
        xs, ys, w = x[lo:hi].to(DEV), y[lo:hi].to(DEV), (hi - lo) / x.shape[0]
        loss = gs.step((xs,), ys, w) if gs else None
        if loss is None:
            l = model.loss_fn(model(xs), ys)
            sync.zero_grad()
            (l * w).backward()
            sync.finish()
            opt.step()
            loss = l.detach() * w
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    if graph:
        assert gs.graph is not None and gs.replays >= 5, (gs.failed, getattr(gs, "error", None))
    return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, opt


def _worker_gloo(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        le, se, _ = _run(rank, world, graph=False)
        lg, sg, _ = _run(rank, world, graph=True)
        torch.save({"le": le, "lg": lg, "se": se, "sg": sg}, os.path.join(tmpdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_graphed_step_equals_the_eager_gradsync_loop(tmp_path):
    mp.spawn(_worker_gloo, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(2)]
    for d in r:
        assert torch.equal(d["le"], d["lg"])
        for k in d["se"]:
            assert torch.equal(d["se"][k], d["sg"][k]), k
    for k in r[0]["sg"]:
        if not k.endswith(("running_mean", "running_var")):             # per-rank batch statistics differ by design (DataParallel replicas)
            assert torch.equal(r[0]["sg"][k], r[1]["sg"][k]), k


def _worker_rccl(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["IM2IM_DIST_SINGLE_RANK"] = "1"            # treat a world of ONE as distributed: every collective call site runs, on RCCL
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(DEV))
    try:
        le, se, _ = _run(rank, world, graph=False)
        os.environ["IM2IM_GRAPH_COLLECTIVES"] = "0"
        ls, ss, _ = _run(rank, world, graph=True)           # all-reduce after the replay
        os.environ["IM2IM_GRAPH_COLLECTIVES"] = "1"
        lg, sg, opt = _run(rank, world, graph=True)         # all-reduce + FusedAdam inside the graph
        t = torch.ones(1 << 20, device=DEV)
        dist.all_reduce(t)
        objs = [None]
        dist.all_gather_object(objs, {"rank": rank, "device": torch.cuda.current_device()})      # bench.py's per-rank device record
        assert objs[0]["rank"] == 0
        from im2im_uq_amd import launch                     # [r5] the start-up check of `bench.py --gpus N`, on RCCL: identities gathered,
        ids = launch.verify_world(dist, rank, world, torch.device(DEV), "nccl")      # ranks counted by an all-reduce, devices distinct
        assert len(ids) == 1 and ids[0]["local_device_index"] == 0 and launch.distinct_devices(ids) == 1
        assert launch.relax_collective_timeout()           # [r6] what init_distributed does after the world check: works on the RCCL group
        dist.all_reduce(t)                                 # ... and collectives go on working after it
        torch.cuda.synchronize()
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        torch.save({"le": le, "ls": ls, "lg": lg, "se": se, "ss": ss, "sg": sg, "rccl": ver, "sum": float(t.sum()),
                    "steps": sorted({int(st["step"]) for st in opt.state.values()})}, os.path.join(tmpdir, "rccl.pt"))
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_eager_and_captured_collectives(tmp_path):
    """the only RCCL configuration a one-GPU box can run: a world of one rank on the nccl backend.  Every call site of the
    N > 1 path executes on RCCL (broadcast of the initial state, bucketed async all-reduce from the weight-gradient stream,
    the wait) -- and with IM2IM_GRAPH_COLLECTIVES=1 inside a captured HIP graph."""
    mp.spawn(_worker_rccl, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d = torch.load(tmp_path / "rccl.pt")
    assert torch.equal(d["le"], d["ls"]) and torch.equal(d["le"], d["lg"])
    for k in d["se"]:
        assert torch.equal(d["se"][k], d["ss"][k]) and torch.equal(d["se"][k], d["sg"][k]), k
    assert d["sum"] == float(1 << 20) and d["steps"] == [9]            # (one rank: the all-reduce leaves the ones as they are)
    print("RCCL", d["rccl"])


def test_graph_not_taken_with_hooks_and_capture_failure_falls_back():
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import GraphedStep
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(3)
    model = add_uncertainty(UNet(1, 1, depth=2), dict(PARAMS)).to(DEV).train()
    assert GraphedStep.static_module(model, {})
    h = model.baseModel.inc.register_forward_hook(lambda m, i, o: None)
    assert not GraphedStep.static_module(model, {})            # a module hook would stop firing under replay
    h.remove()
    h = next(model.parameters()).register_hook(lambda g: g)
    assert not GraphedStep.static_module(model, {})            # so would a tensor hook (wandb.watch registers these)
    h.remove()
    assert GraphedStep.static_module(model, {})
    # a step that synchronises with the host cannot be captured: the graph is abandoned, counters intact, eager from then on
    opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
    gs = GraphedStep(model, opt)
    real = model.loss_fn

    def syncing_loss(pred, y):
        l = real(pred, y)
        float(l.detach().cpu())                               # host sync: illegal during capture
        return l
    model.loss_fn = syncing_loss
    x, y = _batches(1)[0]
    x, y = x.to(DEV), y.to(DEV)
    outs = [gs.step((x,), y) for _ in range(5)]
    assert all(o is not None for o in outs[:3]) and outs[3] is None and outs[4] is None and gs.failed
    assert {int(st["step"]) for st in opt.state.values()} == {3}
    model.loss_fn = real
    loss = model.loss_fn(model(x), y)                          # the eager loop goes on
    opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    assert {int(st["step"]) for st in opt.state.values()} == {4} and bool(torch.isfinite(loss))


def test_invalidate_packed_after_a_write_through_data():
    """`p.data.copy_` does not bump the version counter the packed-weight cache keys on; invalidate_packed() does."""
    from im2im_uq_amd import nn_ops
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=DEV))
    wf0, _ = nn_ops.packed_pair(w, torch.bfloat16)
    w.data.mul_(2.0)
    assert nn_ops.packed_pair(w, torch.bfloat16)[0] is wf0      # stale by construction: nothing told the cache
    nn_ops.invalidate_packed(w)
    wf1, _ = nn_ops.packed_pair(w, torch.bfloat16)
    assert torch.equal(wf1.float(), (wf0.float() * 2).to(torch.bfloat16).float())
