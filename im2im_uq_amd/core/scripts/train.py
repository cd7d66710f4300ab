"""Training driver -- drop-in for the reference's core/scripts/train.py (train_net :62-197,
run_validation :29-60).  Same signature, checkpoint naming and resume rules; what changes:

  * the forward/backward/optimizer arithmetic runs in the HIP kernels (UNet modules + FusedAdam);
  * multi-GPU is one process per GPU (torch.distributed, backend nccl == RCCL over xGMI) instead of the
    reference's single-process nn.DataParallel (:22-27,:112-115): the global batch is split across ranks,
    gradients are averaged with ONE flat all-reduce per step, BatchNorm uses per-rank batch statistics
    exactly like DataParallel replicas do;
  * the per-step `loss.item()` host sync (:155) is replaced by a device-side accumulator read once per
    epoch; the logged value `epoch_loss / num_examples` is the same quantity.
"""
import logging
import os

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .. import _pkg  # noqa: F401
from ... import nn_ops
from ...compat import load_reference_checkpoint
from ...prefetch import DevicePrefetcher, loader as prefetch_loader, to_device
from ._wandb import wandb
from .eval import eval_net, get_images


def _dist():
    """torch.distributed when this process is one of several ranks (IM2IM_DIST_SINGLE_RANK=1: also for a world of ONE rank, so
    that every collective call site can be executed on RCCL by a one-GPU box -- tests/test_graph_ddp_gpu.py)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("IM2IM_DIST_SINGLE_RANK") == "1"):
        return dist
    return None


class DataParallelPassthrough(nn.Module):
    """Kept for API compatibility (reference :22-27).  With one process per GPU there is nothing to wrap:
    this is a transparent holder whose attribute lookups fall through to `.module`."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


class GradSync:
    """Gradient exchange of data-parallel training (replaces nn.DataParallel's reduce-to-GPU0, reference :112-115).

    ONE pre-allocated flat fp32 buffer holds the reduced gradients; it is cut into buckets in reverse registration order
    (the order backward produces gradients).  A post-accumulate hook counts a bucket's parameters down and, the moment the
    last one has its gradient, packs the bucket into the flat buffer (one multi-tensor copy) and launches its all-reduce
    (RCCL, asynchronous) -- so the exchange of the decoder's gradients runs under the encoder's backward.  Packing and the
    collective are issued from the weight-gradient stream (nn_ops.side_stream): the conv weight gradients are computed
    there, so the main stream's chain (data-gradients, BatchNorm backward) never waits for them.  `finish()` launches
    whatever is left, waits, and points every `p.grad` at its slice of the flat buffer for the optimizer.

    Sum, not mean: the caller scales its loss by (local images / global images), so the summed gradients are the
    gradient of the global-batch mean loss even when the global batch does not divide evenly over the ranks."""

    def __init__(self, params, bucket_bytes=8 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.dist = _dist()
        dev = self.params[0].device
        self.cuda = dev.type == "cuda"
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        off = 0
        self.spans = []
        for p in self.params:
            self.spans.append((off, off + p.numel()))
            off += p.numel()
        self.views = [self.flat[lo:hi].view_as(p) for p, (lo, hi) in zip(self.params, self.spans)]
        self.buckets = []                       # [lo, hi) element ranges + member indices, first bucket = last parameters
        self.bucket_of = {}
        hi = off
        members = []
        for i in range(len(self.params) - 1, -1, -1):
            members.append(i)
            if (hi - self.spans[i][0]) * 4 >= bucket_bytes or i == 0:
                b = len(self.buckets)
                self.buckets.append((self.spans[i][0], hi, tuple(members)))
                for m in members:
                    self.bucket_of[m] = b
                members, hi = [], self.spans[i][0]
        self.expected = None                    # per bucket: how many parameters receive a gradient (learnt on step 1)
        self.seen = [0] * len(self.buckets)
        self.fired = set()
        self.launched = [None] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next_bucket = 0                    # buckets [0, next_bucket) are launched
        self.deferred = False                   # True: the hooks only count; nothing is launched before pack_all() / finish()
        self.hook_handles = []
        if self.dist is not None:
            for i, p in enumerate(self.params):
                self.hook_handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_p):
            b = self.bucket_of[i]
            self.fired.add(i)
            self.seen[b] += 1
            if self.expected is not None and self.seen[b] == self.expected[b] and not self.deferred:
                self.ready[b] = True
                # drain in index order; a bucket that is not complete yet (or whose parameters got no gradient on the
                # learning step, expected == -1) holds back the later ones until it completes / until finish()
                while self.next_bucket < len(self.buckets) and self.ready[self.next_bucket]:
                    self._launch(self.next_bucket)
                    self.next_bucket += 1
        return hook

    def _pack(self, b):
        lo, hi, members = self.buckets[b]
        src, dst = [], []
        for m in members:
            g = self.params[m].grad
            if g is None:
                self.views[m].zero_()
            elif g.data_ptr() != self.views[m].data_ptr():
                src.append(g)                                       # _foreach_copy_ converts a non-fp32 gradient on the way
                dst.append(self.views[m])
        if src:
            torch._foreach_copy_(dst, src)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        if self.cuda:
            main = torch.cuda.current_stream(self.flat.device)
            side = nn_ops.side_stream(self.flat.device)
            side.wait_stream(main)              # gradients produced on the main stream (BatchNorm, biases, heads)
            bn = nn_ops._bn_streams.get(self.flat.device.index)
            if bn is not None:
                side.wait_stream(bn)            # ... and dgamma / dbeta of the pipelined BatchNorm backward on its stream
            with torch.cuda.stream(side):       # ... and, in stream order, the conv weight gradients computed on this one
                nn_ops.flush_wgrad_reduce(self.flat.device.index)     # [r6] their deferred split-K reductions, in one launch
                self._pack(b)
                self.launched[b] = self.dist.all_reduce(self.flat[lo:hi], async_op=True)
        else:
            self._pack(b)
            self.launched[b] = self.dist.all_reduce(self.flat[lo:hi], async_op=True)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def own_hook_ids(self):
        """ids of the post-accumulate hooks this object registered (GraphedStep.static_module tells them from a user's)."""
        return {h.id for h in self.hook_handles}

    def _reset(self):
        if self.expected is None:                                  # learnt on the first step: parameters per bucket that get a gradient
            exp = [0] * len(self.buckets)
            for i in self.fired:
                exp[self.bucket_of[i]] += 1
            self.expected = [e if e > 0 else -1 for e in exp]
        self.seen = [0] * len(self.buckets)
        self.fired = set()
        self.launched = [None] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next_bucket = 0

    def pack_all(self):
        """deferred mode, after backward(): every bucket's gradients copied into the flat buffer on the current stream, no
        collective (GraphedStep captures exactly this; the exchange follows outside the graph with reduce_all)."""
        for b in range(len(self.buckets)):
            self._pack(b)
        self._reset()

    def reduce_all(self):
        """all-reduce the packed flat buffer bucket by bucket (index order, asynchronous, then waited for) and point every
        p.grad at its slice."""
        if self.dist is not None:
            hs = [self.dist.all_reduce(self.flat[lo:hi], async_op=True) for lo, hi, _ in self.buckets]
            for h in hs:
                h.wait()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def finish(self):
        """call after backward(): every bucket reduced, visible to the current stream, and p.grad = its flat slice."""
        if self.dist is not None:
            for b in range(self.next_bucket, len(self.buckets)):  # what the hooks did not launch (first step, unused parameters,
                self._launch(b)                                    # no local images) -- still in index order
            for h in self.launched:
                h.wait()
            self._reset()
        else:
            for b in range(len(self.buckets)):
                self._pack(b)
        for p, v in zip(self.params, self.views):
            p.grad = v


def allreduce_gradients(params):
    """one-shot form of GradSync for callers that keep torch's own .grad tensors: average the gradients over ranks
    in place, bucket by bucket (no flat copy)."""
    dist = _dist()
    if dist is None:
        return
    world = dist.get_world_size()
    handles = []
    for p in params:
        if p.grad is not None:
            handles.append(dist.all_reduce(p.grad, async_op=True))
    for h in handles:
        h.wait()
    for p in params:
        if p.grad is not None:
            p.grad /= world


def broadcast_module_state(net, src=0, buffers_only=False):
    """rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does at construction; the
    reference's DataParallel re-broadcasts them every step, :22-27): training starts from ONE model even when the
    caller did not seed the ranks identically.  buffers_only: just the BatchNorm running statistics -- every rank
    updates them from its own share of the batch during an epoch; DataParallel keeps only device 0's, so before anything
    evaluates or saves the model the ranks adopt rank 0's."""
    dist = _dist()
    if dist is None:
        return
    with torch.no_grad():
        for t in ([] if buffers_only else list(net.parameters())) + list(net.buffers()):
            if t.numel():
                dist.broadcast(t.detach(), src=src)
    # c10d collectives write in place WITHOUT bumping the version counter (no ADInplaceOrView kernel; `_version` is the same before and
    # after, through `.data` and `.detach()` alike), and the packed-weight / folded-BatchNorm caches of nn_ops key on it: forget them,
    # or a rank other than `src` that ran a forward before the broadcast would keep serving operands packed from its old weights
    nn_ops.invalidate_packed()


class GlobalBatchSampler(torch.utils.data.Sampler):
    """batch sampler of one rank: every rank walks the SAME (seeded) order of the dataset in global batches of
    `batch_size` (the reference's DataLoader batches, train.py:104) and takes its contiguous slice of each; slices
    differ by at most one image, so a global batch of 78 on 8 ranks is 10+10+10+10+10+10+9+9, never 72.
    Yields (possibly empty) index lists; `global_sizes` has the size of each global batch."""

    def __init__(self, n, batch_size, rank, world, shuffle=True, seed=0):
        self.n, self.batch_size, self.rank, self.world, self.shuffle, self.seed = n, batch_size, rank, world, shuffle, seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    @staticmethod
    def share(count, rank, world):
        base, rem = divmod(count, world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        for s in range(0, self.n, self.batch_size):
            batch = order[s:s + self.batch_size]
            lo, hi = self.share(len(batch), self.rank, self.world)
            yield batch[lo:hi]


def _ckpt_name(checkpoint_dir, epoch, config):
    return (checkpoint_dir + f'/CP_epoch{epoch}_' + config['dataset'] + "_" + config['uncertainty_type'] + "_"
            + str(config['batch_size']) + "_" + str(config['lr']) + "_" + config['input_normalization'] + "_"
            + config['output_normalization'].replace('.', '_') + '.pth')


def run_validation(net, val_loader, val_dataset, device, global_step, epoch, config):
    with torch.no_grad():
        net.eval()
        try:
            (examples_input, examples_lower_edge, examples_prediction, examples_upper_edge, examples_ground_truth,
             examples_ll, examples_ul, results_list) = get_images(net, val_dataset, device,
                                                                  list(range(config['num_validation_images'])), config)
            wandb.log({"epoch": epoch, "iter": global_step, "examples_input": examples_input})
            wandb.log({"epoch": epoch, "iter": global_step, "Lower edge": examples_lower_edge})
            wandb.log({"epoch": epoch, "iter": global_step, "Predictions": examples_prediction})
            wandb.log({"epoch": epoch, "iter": global_step, "Upper edge": examples_upper_edge})
            wandb.log({"epoch": epoch, "iter": global_step, "Ground truth": examples_ground_truth})
            wandb.log({"epoch": epoch, "iter": global_step, "Lower length": examples_ll})
            wandb.log({"epoch": epoch, "iter": global_step, "Upper length": examples_ul})
        except Exception:  # noqa: BLE001  (the reference swallows image-logging failures too, :56-57)
            print("Failed logging images.")
        val_loss = eval_net(net, val_loader, device)
        wandb.log({"epoch": epoch, "iter": global_step, "val_loss": val_loss})
        print(f"Val loss: {val_loss}")
    net.train()


_graph_streams = {}


def _graph_stream():
    """the stream every GraphedStep of this process runs and captures on.  torch hands out side streams from a pool of 32 per device,
    round-robin: a sweep that builds a GraphedStep per run would eventually get the hardware queue of the weight-gradient stream."""
    idx = torch.cuda.current_device()
    st = _graph_streams.get(idx)
    if st is None:
        st = _graph_streams[idx] = torch.cuda.Stream(device=idx)
    return st


class GraphedStep:
    """One training step of a fixed batch shape as ONE HIP graph (the launch-bound regime: a 32x32 depth-2 step is ~150 kernel
    launches of a few microseconds each, 2.9 ms of host enqueue for 1.2 ms of GPU work; replaying the captured graph costs the
    host 0.5 ms -- 11 k -> 26 k img/s, `profiles/r03_ab_experiments.txt`).  Same arithmetic, same kernels, same order: losses,
    parameters and buffers are bit-identical to the eager loop's (tests/test_round3_gpu.py, tests/test_round4_gpu.py).

    What is in the graph: forward, loss, backward (weight re-packing, BatchNorm running statistics and `num_batches_tracked`
    are kernels), and -- [r4] -- the optimizer step: FusedAdam is switched to `capturable` (step count on the device,
    im2im_adam_step_dev), so a replayed update has the right bias correction.  Another optimizer stays outside the graph.

    Data parallelism [r4] (`sync`: this rank's GradSync): the graph holds forward + backward + the packing of the gradients
    into GradSync's flat buffer; the all-reduce of the buckets is issued by the host right after the replay (any backend: gloo
    cannot be captured) and FusedAdam runs after it.  With backend nccl and IM2IM_GRAPH_COLLECTIVES=1 the RCCL all-reduces are
    captured too (they are launched from the backward hooks, so the exchange overlaps the rest of the backward inside the
    graph) and the whole step is one replay.

    The first WARM steps run eagerly on the capture stream (they are real training steps), then the step is captured once and
    replayed.  A batch of another shape (the short last one of an epoch) or another loss weight is run as a plain step on this
    object's stream ([r6] `_eager_other`: every training step -- warm-up, odd, captured -- lives on ONE stream, so autograd's
    AccumulateGrad nodes never belong to another).  A capture that fails (a module that synchronises, a hook that reads a tensor) is abandoned:
    the optimizer's counters are put back, `failed` is set and every later call returns None.  fp8 mode is not captured (it
    rotates its amax slots on the host)."""
    WARM = 3

    def __init__(self, net, optimizer, sync=None):
        self.net, self.opt, self.sync = net, optimizer, sync
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.stream = _graph_stream()                      # ONE per device and process (torch's stream pool wraps around after 32)
        self.graph = None
        self.key = None
        self.done = 0
        self.failed = False
        self.adam_in_graph = isinstance(optimizer, nn_ops.FusedAdam)
        if self.adam_in_graph:
            optimizer.capturable = True              # from its first step on: eager and replayed steps use the same kernels
        backend = None
        if sync is not None and sync.dist is not None:
            backend = sync.dist.get_backend()
        self.collectives_in_graph = (sync is not None and sync.dist is not None and backend == "nccl"
                                     and os.environ.get("IM2IM_GRAPH_COLLECTIVES", "0") == "1")
        self.replays = 0

    @staticmethod
    def wanted(config, labels_numel, world, dtype_name):
        """config key `hip_graph` (True / False / "auto", default auto; env IM2IM_HIP_GRAPH=0/1 overrides): auto = batches of at
        most IM2IM_HIP_GRAPH_MAX_PIXELS label pixels (default 2^18) in a single process -- above that the step is GPU-bound and
        a graph gains nothing (`profiles/r04_ab_experiments.txt`); with several ranks only when asked for (True / env 1)."""
        env = os.environ.get("IM2IM_HIP_GRAPH")
        if env is not None:
            want = {"0": False, "1": True}.get(env, "auto")
        else:
            try:
                want = config.get("hip_graph", "auto")
            except Exception:  # noqa: BLE001  (no config / a config object without .get)
                want = "auto"
        if dtype_name == "fp8" or want is False or str(want).lower() in ("false", "0"):
            return False
        if want is True or str(want).lower() in ("true", "1"):
            return True
        return world == 1 and labels_numel <= int(os.environ.get("IM2IM_HIP_GRAPH_MAX_PIXELS", str(1 << 18)))

    _hook_note_done = False

    @staticmethod
    def foreign_hook_owner(net, sync=None):
        """name of the first module / parameter that carries a hook this class does not account for, or None"""
        own = sync.own_hook_ids() if sync is not None else set()
        for mname, m in net.named_modules():
            for name in ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks"):
                if getattr(m, name, None):
                    return f"module '{mname or type(m).__name__}' ({name[1:]})"
        for pname, p in net.named_parameters():
            if getattr(p, "_backward_hooks", None):
                return f"parameter '{pname}' (tensor hook)"
            post = getattr(p, "_post_accumulate_grad_hooks", None)
            if post and any(k not in own for k in post):
                return f"parameter '{pname}' (post-accumulate-grad hook)"
        return None

    @staticmethod
    def has_foreign_hooks(net, sync=None):
        """a replayed graph runs no Python: forward / backward hooks on a module and tensor hooks on a parameter (wandb.watch
        registers both) would silently stop firing.  GradSync's own post-accumulate hooks are accounted for by this class."""
        own = sync.own_hook_ids() if sync is not None else set()
        for m in net.modules():
            for name in ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks"):
                if getattr(m, name, None):
                    return True
        for p in net.parameters():
            if getattr(p, "_backward_hooks", None):
                return True
            post = getattr(p, "_post_accumulate_grad_hooks", None)
            if post and any(k not in own for k in post):
                return True
        return False

    @staticmethod
    def static_module(net, config, sync=None):
        """a replayed graph repeats the kernel sequence of the captured step, so "auto" only applies to networks made of this
        package's own modules and plain torch.nn layers (a user-supplied trunk or final layer may branch in Python on its data)
        that carry no hooks; `hip_graph: true` forces the graph for those as well."""
        try:
            forced = config.get("hip_graph", "auto") is True or os.environ.get("IM2IM_HIP_GRAPH") == "1"
        except Exception:  # noqa: BLE001
            forced = os.environ.get("IM2IM_HIP_GRAPH") == "1"
        if forced:
            return True
        if not all(type(m).__module__.startswith(("im2im_uq_amd.", "torch.nn.modules.")) for m in net.modules()):
            return False
        owner = GraphedStep.foreign_hook_owner(net, sync)
        if owner is not None and not GraphedStep._hook_note_done:
            GraphedStep._hook_note_done = True
            # wandb.watch(net) (train_net calls it, reference :122) registers parameter hooks: with a live wandb run auto mode
            # would silently never graph -- say so once
            msg = (f"HIP-graph training step not used in auto mode: {owner} carries a hook that a replayed graph would not run "
                   f"(wandb.watch registers such hooks). Set `hip_graph: true` in the config (or IM2IM_HIP_GRAPH=1) to graph anyway.")
            logging.warning(msg)
            print(msg)
        return owner is None

    # ------------------------------------------------------------------ the pieces of a step
    def _fwd_bwd(self):
        pred = self.net(*self.xs)
        loss = self.net.loss_fn(pred, self.y)
        if self.sync is None:
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
        else:
            self.sync.zero_grad()
            (loss * self.weight).backward()
        return loss

    def _eager(self):
        """exactly the step train_net's plain loop runs (the WARM steps)."""
        loss = self._fwd_bwd()
        if self.sync is not None:
            self.sync.finish()
        nn_ops.join_side_streams()
        self.opt.step()
        return loss.detach() * self.weight if self.sync is not None else loss.detach()

    def _eager_other(self, x, labels, weight):
        """a batch of another shape / loss weight (the short last batch of an epoch): the plain step, on THIS object's stream and with
        its autograd graph gone when it returns.  [r6] Run by the caller on its own stream with `loss` kept alive into the next
        iteration -- what train_net did -- it left AccumulateGrad nodes bound to the caller's (default) stream; the capture that
        followed in the next epoch then accumulated gradients on the default stream and hipStreamEndCapture fell over."""
        pred = self.net(*x)
        loss = self.net.loss_fn(pred, labels)
        if self.sync is None:
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
        else:
            self.sync.zero_grad()
            (loss * weight).backward()
            self.sync.finish()
        nn_ops.join_side_streams()
        self.opt.step()
        return loss.detach() * weight if self.sync is not None else loss.detach()

    def _addresses(self):
        return (tuple(p.data_ptr() for p in self.params), tuple(b.data_ptr() for b in self.net.buffers() if b is not None),
                nn_ops._Scratch.addresses())

    def _drop_graph(self):
        if self.graph is not None:
            self.graph = None
            nn_ops._Scratch.unpin()

    def _capture(self, cur):
        nn_ops._Scratch.pinned = True
        allocs_before = self.opt._ctr_allocs if self.adam_in_graph else None
        steps_before = [(p, self.opt.state[p].get("step")) for p in self.params if self.opt.state.get(p)] if self.adam_in_graph else []
        ctrs_before = dict(self.opt._ctrs) if self.adam_in_graph else None
        graph = torch.cuda.CUDAGraph()
        try:
            if self.sync is None:
                self.opt.zero_grad(set_to_none=True)
            else:
                self.sync.zero_grad()
                self.sync.deferred = not self.collectives_in_graph
            self.stream.wait_stream(cur)
            # thread_local: only this thread's calls are held to the capture rules -- the prefetcher's producer thread (im2im_uq_amd/prefetch.py)
            # synchronises events and pins memory while this thread captures, which the default "global" mode turns into a failed capture
            with torch.cuda.graph(graph, stream=self.stream, capture_error_mode="thread_local"):
                loss = self._fwd_bwd()
                if self.sync is not None:
                    if self.collectives_in_graph:
                        self.sync.finish()
                    else:
                        self.sync.pack_all()
                nn_ops.join_side_streams()
                self.in_graph_opt = self.adam_in_graph and (self.sync is None or self.collectives_in_graph)
                if self.in_graph_opt:
                    self.opt.step()
                self.loss = loss.detach() * self.weight if self.sync is not None else loss.detach()
            if allocs_before is not None and self.opt._ctr_allocs != allocs_before:
                raise RuntimeError("FusedAdam allocated a step counter inside the capture (a replay would reset the step count)")
        except Exception as e:  # noqa: BLE001
            self.failed = True
            nn_ops._Scratch.pinned = nn_ops._Scratch.graphs > 0
            self.error = f"{type(e).__name__}: {e}"
            if self.sync is not None:
                self.sync.deferred = False
                self.sync._reset()
            for p, st in steps_before:                        # the optimizer's host bookkeeping of the step that never ran
                self.opt.state[p]["step"] = st
            if ctrs_before is not None:
                self.opt._ctrs = ctrs_before
            for p in self.params:
                p.grad = None
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            logging.warning("HIP graph capture of the training step failed (%s); continuing eagerly", self.error)
            return False
        finally:
            if self.sync is not None:
                self.sync.deferred = False
        self.graph = graph
        nn_ops._Scratch.pin()
        self.hyper = self.opt.hyper_key() if self.adam_in_graph else None     # lr / betas / eps are frozen into the graph by value
        self.grads = [p.grad for p in self.params]            # without a GradSync: the graph's own gradient tensors
        self.stepped = [p for p in self.params if p.grad is not None]    # what the captured optimizer step updates
        self.addr = self._addresses()
        self.fresh = True                                      # the capture did the host bookkeeping of its first replay
        return True

    def step(self, x, labels, weight=1.0):
        """-> the step's (weighted) loss as a 0-dim tensor, or None when the caller has to run the batch itself (after a failed capture)."""
        if self.failed:
            return None
        key = (tuple((tuple(t.shape), t.dtype) for t in x), tuple(labels.shape), labels.dtype, float(weight))
        if self.key is None:
            self.key = key
            self.weight = float(weight)
            self.xs = tuple(torch.empty_like(t) for t in x)
            self.y = torch.empty_like(labels)
        cur = torch.cuda.current_stream()
        if key != self.key:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._eager_other(x, labels, float(weight))
            cur.wait_stream(self.stream)
            return out
        for dst, src in zip(self.xs, x):
            dst.copy_(src, non_blocking=True)
        self.y.copy_(labels, non_blocking=True)
        if self.graph is not None and self._addresses() != self.addr:
            # parameters / buffers moved (train_net's checkpoint does net.cpu() ... net.to(device)) or a scratch buffer grew:
            # the captured addresses are stale
            self._drop_graph()
            self.done = 0
        if self.graph is not None and self.adam_in_graph and self.opt.hyper_key() != self.hyper:
            self._drop_graph()                                # an LR scheduler stepped / group['lr'] was edited: capture again (no warm-up needed)
        if self.graph is None and self.done < self.WARM:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._eager()
            cur.wait_stream(self.stream)
            self.done += 1
            return out
        if self.graph is None and not self._capture(cur):
            return None
        self.graph.replay()
        self.replays += 1
        if self.sync is None:
            for p, g in zip(self.params, self.grads):        # an eager step in between re-pointed p.grad
                p.grad = g
        elif self.collectives_in_graph:
            for p, v in zip(self.sync.params, self.sync.views):
                p.grad = v
        else:
            self.sync.reduce_all()                            # the exchange itself: outside the graph (any backend)
        if self.in_graph_opt:
            if not self.fresh:
                self.opt.advance(self.stepped)                # host bookkeeping of the step the graph just took
            self.fresh = False
        else:
            self.opt.step()
        return self.loss


def train_net(net, train_dataset, val_dataset, device, epochs, batch_size, lr, load_from_checkpoint, checkpoint_dir,
              checkpoint_every, validate_every, config=None):
    starting_epoch = 0
    if config is None:
        config = wandb.config
    if load_from_checkpoint:
        checkpoint_final_path = _ckpt_name(checkpoint_dir, epochs, config)
        if os.path.exists(checkpoint_final_path):
            try:
                net = load_reference_checkpoint(checkpoint_final_path)      # this package's or the reference's pickle
                net.eval()
                print(f"Model loaded from checkpoint {checkpoint_final_path}")
                return net
            except Exception:  # noqa: BLE001
                print(f"Final model cannot be loaded from checkpoint {checkpoint_final_path}. Training now, for {epochs} epochs.")
        else:
            print(f"Final model cannot be loaded from checkpoint {checkpoint_final_path}. Training now, for {epochs} epochs.")
            for e in reversed(range(epochs)):
                checkpoint_intermediate_path = _ckpt_name(checkpoint_dir, e, config)
                if os.path.exists(checkpoint_intermediate_path):
                    net = load_reference_checkpoint(checkpoint_intermediate_path)
                    starting_epoch = e
                    print(f"Starting from epoch {e}.")
                    break

    global_step = 0
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    sampler = None
    iterable = False
    try:
        if dist:
            # the global batch is split over ranks, as DataParallel's scatter does (remainder images go to the first ranks)
            sampler = GlobalBatchSampler(len(train_dataset), batch_size, rank, world, shuffle=True, seed=0)
            train_loader = None
        else:
            # (the reference's loader, :104; its batches are stacked straight into pinned staging memory -- im2im_uq_amd/prefetch.py)
            train_loader = prefetch_loader(train_dataset, device, batch_size=batch_size, shuffle=True, num_workers=0)
    except Exception:  # noqa: BLE001  (iterable datasets cannot be shuffled, reference :105-106)
        # with several ranks every rank reads the same global batches from the stream and keeps its slice of each
        iterable = True
        sampler = None
        train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=False, num_workers=0)
    val_loader = DataLoader(val_dataset, batch_size=batch_size, shuffle=False, num_workers=0)

    net = net.to(device=device)
    broadcast_module_state(net)                              # every rank starts from rank 0's weights and BatchNorm buffers
    optimizer = nn_ops.FusedAdam(net.parameters(), lr=lr)    # torch.optim.Adam defaults, reference :120
    if starting_epoch == 0:
        try:
            wandb.watch(net, log_freq=100)
        except Exception:  # noqa: BLE001
            wandb.init(config=config)
            wandb.watch(net, log_freq=100)

    sync = GradSync(net.parameters()) if dist else None
    graphed = None                                           # decided at the first batch (GraphedStep.wanted)
    print("Start Training!")
    for epoch in range(starting_epoch, epochs):
        net = net.to(device)
        net.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        print('epoch ' + str(epoch + 1) + '\n')
        epoch_loss = torch.zeros((), dtype=torch.float64, device=device)
        num_examples = 0
        if sampler is not None:
            collate = torch.utils.data.default_collate
            batches = ((collate([train_dataset[i] for i in idx]) if idx else None, min(batch_size, len(train_dataset) - k * batch_size))
                       for k, idx in enumerate(sampler))
        elif dist and iterable:
            def _sliced():
                for b in train_loader:
                    n_glob = b[-1].shape[0]
                    lo, hi = GlobalBatchSampler.share(n_glob, rank, world)
                    yield ([t[lo:hi] for t in b] if hi > lo else None), n_glob
            batches = _sliced()
        else:
            batches = ((b, None) for b in train_loader)
        # [r6] host batches reach HBM through the two-deep pinned prefetcher (im2im_uq_amd/prefetch.py): batch k+1 is fetched, staged and
        # uploaded on a copy stream while the kernels of batch k run -- same batches, same order, same values as the in-line
        # `.to(device)` of the reference (:147-149), which stalls the GPU for every collation and pageable copy
        for batch, global_n in (batches if isinstance(train_loader, DevicePrefetcher) else to_device(batches, device)):
            if batch is None:                                # this rank's share of a short last batch is empty: it still joins the exchange
                sync.zero_grad(); sync.finish(); optimizer.step()
                global_step += 1
                continue
            labels = batch[-1].to(device=device)
            x = tuple([batch[i].to(device=device, dtype=torch.float32) for i in range(len(batch) - 1)])

            weight = (labels.shape[0] / global_n if global_n else 1.0 / world) if sync is not None else 1.0
            if graphed is None:
                graphed = GraphedStep(net, optimizer, sync) if (torch.device(device).type == "cuda" and GraphedStep.wanted(
                    config, labels.numel(), world, nn_ops.compute_mode()) and GraphedStep.static_module(net, config, sync)) else False
            if graphed:
                gl = graphed.step(x, labels, weight)
                if gl is not None:
                    epoch_loss += gl
                    global_step += 1
                    num_examples += labels.shape[0]
                    continue

            labels_pred = net(*x)
            loss = net.loss_fn(labels_pred, labels)

            if sync is None:
                epoch_loss += loss.detach()
                optimizer.zero_grad()
                loss.backward()
            else:
                # summed over ranks, (n_local / n_global) * local mean == the global batch mean DataParallel's GPU-0 loss is
                epoch_loss += loss.detach() * weight
                sync.zero_grad()
                (loss * weight).backward()
                sync.finish()
            optimizer.step()

            global_step += 1
            num_examples += labels.shape[0]
            labels_pred = loss = None                        # the step's autograd graph does not outlive the step

        if dist:                                             # the logged quantity is the global one (sum of batch-mean losses / #examples)
            tot = torch.stack([epoch_loss, torch.tensor(float(num_examples), dtype=torch.float64, device=device)])
            dist.all_reduce(tot)
            epoch_loss, num_examples = tot[0], int(tot[1].item())
        wandb.log({"iter": global_step, "train_loss": epoch_loss.item() / max(num_examples, 1)})

        broadcast_module_state(net, buffers_only=True)      # BatchNorm running statistics: rank 0's, as DataParallel keeps device 0's
        with torch.no_grad():
            if (epoch) % validate_every == 0:
                run_validation(net, val_loader, val_dataset, device, global_step, epoch, config)

            if (epoch + 1) % checkpoint_every == 0 and rank == 0:
                print('saving checkpoint')
                if checkpoint_dir is not None:
                    try:
                        os.makedirs(checkpoint_dir, exist_ok=True)
                        logging.info('Created checkpoint directory')
                    except OSError:
                        pass
                    checkpoint_fname = _ckpt_name(checkpoint_dir, epoch + 1, config)
                    # whole-module pickle like the reference (:191); works on one device too (the reference's
                    # `.module` only exists on its DataParallel wrapper)
                    torch.save(getattr(net, "module", net).cpu(), checkpoint_fname)
                    net = net.to(device)
                    logging.info(f'Checkpoint {epoch + 1} saved !')
        net.eval()
    return net
