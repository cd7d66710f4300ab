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
from ._wandb import wandb
from .eval import eval_net, get_images


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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
        if self.dist is not None:
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_p):
            b = self.bucket_of[i]
            self.fired.add(i)
            self.seen[b] += 1
            if self.expected is not None and self.seen[b] == self.expected[b]:
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
                self._pack(b)
                self.launched[b] = self.dist.all_reduce(self.flat[lo:hi], async_op=True)
        else:
            self._pack(b)
            self.launched[b] = self.dist.all_reduce(self.flat[lo:hi], async_op=True)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def finish(self):
        """call after backward(): every bucket reduced, visible to the current stream, and p.grad = its flat slice."""
        if self.dist is not None:
            for b in range(self.next_bucket, len(self.buckets)):  # what the hooks did not launch (first step, unused parameters,
                self._launch(b)                                    # no local images) -- still in index order
            for h in self.launched:
                h.wait()
            if self.expected is None:
                exp = [0] * len(self.buckets)
                for i in self.fired:
                    exp[self.bucket_of[i]] += 1
                self.expected = [e if e > 0 else -1 for e in exp]
            self.seen = [0] * len(self.buckets)
            self.fired = set()
            self.launched = [None] * len(self.buckets)
            self.ready = [False] * len(self.buckets)
            self.next_bucket = 0
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
                dist.broadcast(t.data, src=src)


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


class GraphedStep:
    """Forward + loss + backward of one fixed batch shape as ONE HIP graph (the launch-bound regime: a 32x32 depth-2 step is
    ~150 kernel launches of a few microseconds each, 2.9 ms of host enqueue for 1.2 ms of GPU work; replaying the captured
    graph costs the host 0.5 ms -- 11 k -> 26 k img/s, `profiles/r03_ab_experiments.txt`).  Same arithmetic, same kernels, same
    order: the losses are bit-identical to the eager loop's (tools/graph_probe.py, tests/test_round3_gpu.py).

    The first WARM steps run eagerly on the capture stream (they are real training steps), then the step is captured once and
    replayed; Adam stays outside the graph (its bias correction takes the step count as a host scalar).  A batch of another
    shape (the short last one of an epoch) makes `step` return None and the caller runs it eagerly.  Weight re-packing,
    BatchNorm running statistics and `num_batches_tracked` are kernels and are part of the graph.  Single process, bf16 / fp32
    (the fp8 mode rotates its amax slots on the host)."""
    WARM = 3

    def __init__(self, net, optimizer):
        self.net, self.opt = net, optimizer
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.stream = torch.cuda.Stream()
        self.graph = None
        self.key = None
        self.done = 0

    @staticmethod
    def wanted(config, labels_numel, world, dtype_name):
        """config key `hip_graph` (True / False / "auto", default auto; env IM2IM_HIP_GRAPH=0/1 overrides): auto = batches of at
        most 2^18 label pixels -- above that the step is GPU-bound and a graph gains nothing (batch 10 at 320x320: 7.7 vs 7.3 ms)."""
        env = os.environ.get("IM2IM_HIP_GRAPH")
        if env is not None:
            want = {"0": False, "1": True}.get(env, "auto")
        else:
            try:
                want = config.get("hip_graph", "auto")
            except Exception:  # noqa: BLE001  (no config / a config object without .get)
                want = "auto"
        if world > 1 or dtype_name == "fp8" or want is False or str(want).lower() in ("false", "0"):
            return False
        if want is True or str(want).lower() in ("true", "1"):
            return True
        return labels_numel <= (1 << 18)

    @staticmethod
    def static_module(net, config):
        """a replayed graph repeats the kernel sequence of the captured step, so "auto" only applies to networks made of this
        package's own modules and plain torch.nn layers (a user-supplied trunk or final layer may branch in Python on its data; `hip_graph: true` forces
        the graph for those as well)."""
        try:
            forced = config.get("hip_graph", "auto") is True or os.environ.get("IM2IM_HIP_GRAPH") == "1"
        except Exception:  # noqa: BLE001
            forced = os.environ.get("IM2IM_HIP_GRAPH") == "1"
        return forced or all(type(m).__module__.startswith(("im2im_uq_amd.", "torch.nn.modules.")) for m in net.modules())

    def _eager(self):
        pred = self.net(*self.xs)
        loss = self.net.loss_fn(pred, self.y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def step(self, x, labels):
        key = (tuple((tuple(t.shape), t.dtype) for t in x), tuple(labels.shape), labels.dtype)
        if self.key is None:
            self.key = key
            self.xs = tuple(torch.empty_like(t) for t in x)
            self.y = torch.empty_like(labels)
        if key != self.key:
            return None
        cur = torch.cuda.current_stream()
        for dst, src in zip(self.xs, x):
            dst.copy_(src, non_blocking=True)
        self.y.copy_(labels, non_blocking=True)
        if self.graph is not None and any(p.data_ptr() != a for p, a in zip(self.params, self.addr)):
            # the parameters moved (train_net's checkpoint does net.cpu() ... net.to(device)): the captured addresses are stale
            self.graph, self.done = None, 0
        if self.graph is None and self.done < self.WARM:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                loss = self._eager()
                nn_ops.join_side_streams()
                self.opt.step()
                out = loss.detach()
            cur.wait_stream(self.stream)
            self.done += 1
            return out
        if self.graph is None:
            self.opt.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
            self.stream.wait_stream(cur)
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss = self._eager()
                nn_ops.join_side_streams()
            self.grads = [p.grad for p in self.params]
            self.addr = [p.data_ptr() for p in self.params]
        self.graph.replay()
        for p, g in zip(self.params, self.grads):            # an eager step in between re-pointed p.grad
            p.grad = g
        self.opt.step()
        return self.loss.detach()


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
            train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=0)
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
        for batch, global_n in batches:
            if batch is None:                                # this rank's share of a short last batch is empty: it still joins the exchange
                sync.zero_grad(); sync.finish(); optimizer.step()
                global_step += 1
                continue
            labels = batch[-1].to(device=device)
            x = tuple([batch[i].to(device=device, dtype=torch.float32) for i in range(len(batch) - 1)])

            if graphed is None:
                graphed = GraphedStep(net, optimizer) if (torch.device(device).type == "cuda" and GraphedStep.wanted(
                    config, labels.numel(), world, nn_ops.compute_mode()) and GraphedStep.static_module(net, config)) else False
            if graphed:
                gl = graphed.step(x, labels)
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
                weight = labels.shape[0] / global_n if global_n else 1.0 / world
                epoch_loss += loss.detach() * weight
                sync.zero_grad()
                (loss * weight).backward()
                sync.finish()
            optimizer.step()

            global_step += 1
            num_examples += labels.shape[0]

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
