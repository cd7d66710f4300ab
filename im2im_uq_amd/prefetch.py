"""Host dataset -> HBM without stalling the GPU: a two-deep pinned prefetcher behind the reference's own loaders.

The reference's loops fetch a batch on the host and upload it in line (core/scripts/train.py:147-149,
core/calibration/calibrate_model.py:118-123: `DataLoader(num_workers=0)`, then `.to(device)` of pageable tensors).
A pageable upload is a synchronous copy on the compute stream: the host blocks until every kernel already queued has
run, so the GPU idles while the next batch is collated (13 ms for 78 images of 320 x 320) and copied, and the host
idles while the GPU computes.  `DevicePrefetcher` keeps the order and the contents of the batches and moves that work
off the critical path:

  * a producer thread pulls batch k+1, k+2 from the wrapped iterable (the unchanged DataLoader / sampler walk: same
    order, same RNG draws, one batch at a time) while the main thread enqueues the kernels of batch k;
  * every tensor of a batch goes through a PINNED staging buffer (a ring of depth + 2 slots, allocated once) and is
    uploaded with `copy_(non_blocking=True)` on a copy stream into a ring of device buffers;
  * hand-off by events only: the consumer's stream waits for the slot's `ready` event; when the consumer asks for the
    next batch it records `released` on its stream, which the copy stream waits for before it overwrites that slot.
    No stream or device synchronisation anywhere, no `record_stream` (see nn_ops._on_side_stream for why not).

The tensors handed out are views of the device ring: valid until the NEXT batch is requested (what a `for batch in
loader:` body needs; autograd's saved inputs are consumed by `backward()` before that).  Values are bit-identical to
the in-line loop's (tests/test_round6_gpu.py).  `IM2IM_PREFETCH=0` restores the in-line uploads, `IM2IM_PREFETCH_THREAD=0`
keeps the staging + copy stream but fetches in the consumer's thread.
"""
from __future__ import annotations

import os
import queue
import threading

import torch

ENABLED = os.environ.get("IM2IM_PREFETCH", "1") != "0"
THREAD = os.environ.get("IM2IM_PREFETCH_THREAD", "1") != "0"
DEPTH = int(os.environ.get("IM2IM_PREFETCH_DEPTH", "2"))
# intra-op threads of the host-side copies (the DataLoader's collation, the staging copy).  torch's default is one thread per core, and on
# a many-core host a 32 MB `torch.stack` spread over 128 threads takes 121 ms where 8 threads take 1.7 ms (tools/host_pipe_probe.py on the
# 256-core MI355X host: the fork/join of a hundred threads on a shared machine costs more than the copy).  The producer thread sets its
# own OpenMP team size; in the consumer's thread the setting is restored after every fetch.
HOST_THREADS = int(os.environ.get("IM2IM_PREFETCH_THREADS", "8"))


class _host_threads:
    """`with _host_threads():` -- at most HOST_THREADS intra-op threads for the calling thread's torch CPU ops"""

    def __enter__(self):
        self.was = torch.get_num_threads()
        if HOST_THREADS > 0 and self.was > HOST_THREADS:
            torch.set_num_threads(HOST_THREADS)
        else:
            self.was = None

    def __exit__(self, *exc):
        if self.was is not None:
            torch.set_num_threads(self.was)


_copy_streams = {}


def _copy_stream(device):
    """ONE copy stream per device for the life of the process.  torch hands out side streams from a pool of 32 per device, round-robin:
    a stream created per loader (one per epoch, one per validation pass) would, after a few dozen epochs, be the SAME hardware queue as
    the weight-gradient stream or as GraphedStep's capture stream -- uploads would queue behind kernels, or be recorded into a capture."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    st = _copy_streams.get(idx)
    if st is None:
        st = _copy_streams[idx] = torch.cuda.Stream(device=idx)
    return st


def _pinned(n, dtype):
    return torch.empty(n, dtype=dtype).pin_memory()


class _Slot:
    """staging (pinned) and device buffers of one in-flight batch, and its two hand-off events"""
    __slots__ = ("pin", "dev", "ready", "released", "keep", "k")

    def __init__(self):
        self.pin, self.dev = [], []
        self.k = 0                 # staging buffers of this slot the collate function has filled for the batch being fetched
        self.ready = None          # recorded on the copy stream after the slot's uploads
        self.released = None       # recorded on the consumer's stream when it let go of the slot's device tensors
        self.keep = None           # already-pinned source tensors the uploads read from (no staging copy for those)

    @staticmethod
    def _fit(bufs, i, shape, dtype, make):
        n = 1
        for s in shape:
            n *= int(s)
        while len(bufs) <= i:
            bufs.append(None)
        b = bufs[i]
        if b is None or b.dtype != dtype or b.numel() < n:
            b = bufs[i] = make(max(n, 1), dtype)
        return b[:n].view(shape)


class _Stop:
    def __init__(self, error=None):
        self.error = error


def _map_tensors(item, fn):
    """rebuild `item` (tensor | list | tuple | dict | anything else) with fn applied to every tensor, traversal order fixed"""
    if isinstance(item, torch.Tensor):
        return fn(item)
    if isinstance(item, tuple) and hasattr(item, "_fields"):
        return type(item)(*(_map_tensors(v, fn) for v in item))
    if isinstance(item, (list, tuple)):
        return type(item)(_map_tensors(v, fn) for v in item)
    if isinstance(item, dict):
        return {k: _map_tensors(v, fn) for k, v in item.items()}
    return item


class DevicePrefetcher:
    """iterate `batches` with every host tensor replaced by a device tensor, `depth` batches ahead of the consumer.
    Non-tensor leaves (None for a rank's empty share, the global batch size train_net carries along) pass through."""

    def __init__(self, batches, device, depth: int | None = None, thread: bool | None = None):
        self.batches = batches
        self.device = torch.device(device)
        self.depth = max(1, DEPTH if depth is None else int(depth))
        self.use_thread = THREAD if thread is None else bool(thread)
        self.active = ENABLED and self.device.type == "cuda"
        self._collate_slot = None                          # the slot whose staging buffers `collate` may fill right now

    def __len__(self):
        return len(self.batches)

    def collate(self, samples):
        """`collate_fn` for a DataLoader this prefetcher wraps (see `loader`): torch's default collation of (tuples of) equally
        shaped tensors, stacked straight into the current slot's PINNED staging buffers -- one host copy per batch instead of
        collate-into-a-fresh-tensor (64 MB of first-touch page faults per batch of 78 images) + staging copy.  Anything else, or a
        call from outside the prefetcher's fetch, is torch.utils.data.default_collate."""
        from torch.utils.data import default_collate
        slot = self._collate_slot
        elem = samples[0] if len(samples) else None
        if (slot is None or not isinstance(elem, (tuple, list)) or hasattr(elem, "_fields")
                or not all(isinstance(t, torch.Tensor) and not t.is_cuda for t in elem)):
            return default_collate(samples)
        cols = [[smp[j] for smp in samples] for j in range(len(elem))]
        if any(len(smp) != len(elem) for smp in samples) or any(c.shape != col[0].shape or c.dtype != col[0].dtype for col in cols for c in col):
            return default_collate(samples)
        out = []
        for col in cols:
            buf = _Slot._fit(slot.pin, slot.k, (len(col),) + tuple(col[0].shape), col[0].dtype, _pinned)
            slot.k += 1
            torch.stack(col, 0, out=buf)
            out.append(buf)
        return out

    # ------------------------------------------------------------------ producer side
    def _upload(self, item, slot, copy_stream):
        """stage + enqueue the uploads of one batch; returns the batch with device tensors (views of the slot's ring buffers)"""
        dev = self.device
        k = [slot.k, 0]                                    # next free staging buffer (after the ones `collate` filled), next device buffer
        slot.keep = []
        pinned = _pinned

        def device(n, dtype):
            return torch.empty(n, dtype=dtype, device=dev)

        with torch.cuda.stream(copy_stream):
            if slot.released is not None:
                copy_stream.wait_event(slot.released)     # the consumer's kernels that read the slot's previous contents

            def move(t):
                if t.is_cuda:
                    return t
                src = t.detach()
                if src.is_pinned() and src.is_contiguous():
                    slot.keep.append(src)                  # `collate`'s views of this slot's staging buffers, or a caller's pinned tensor:
                else:                                      # referenced until the slot is taken again (after its `ready` event completed)
                    stage = _Slot._fit(slot.pin, k[0], src.shape, src.dtype, pinned)
                    k[0] += 1
                    stage.copy_(src)                       # host memcpy (the GIL is released inside)
                    src = stage
                out = _Slot._fit(slot.dev, k[1], src.shape, src.dtype, device)
                k[1] += 1
                out.copy_(src, non_blocking=True)
                return out

            out = _map_tensors(item, move)
            if slot.ready is None:
                slot.ready = torch.cuda.Event()
            slot.ready.record(copy_stream)
        return out

    def _fetch(self, it, slot):
        """next batch of the wrapped iterable; while it is fetched `collate` may fill the slot's staging buffers"""
        if slot.ready is not None:
            slot.ready.synchronize()                       # the uploads that last read this slot's staging buffers (long done)
        slot.k = 0
        self._collate_slot = slot
        try:
            return next(it)
        finally:
            self._collate_slot = None

    def _produce(self, it, free, ready, copy_stream, stop):
        try:
            torch.cuda.set_device(self.device)
            if HOST_THREADS > 0 and torch.get_num_threads() > HOST_THREADS:
                torch.set_num_threads(HOST_THREADS)        # this thread's OpenMP team (the consumer's thread keeps its own setting)
            while True:
                slot = free.get()
                if slot is None or stop.is_set():
                    return
                try:
                    item = self._fetch(it, slot)
                except StopIteration:
                    break
                ready.put((self._upload(item, slot, copy_stream), slot))
            ready.put((_Stop(), None))
        except BaseException as e:  # noqa: BLE001  -- handed to the consumer, which re-raises it
            ready.put((_Stop(e), None))

    # ------------------------------------------------------------------ consumer side
    def __iter__(self):
        if not self.active:
            yield from self.batches
            return
        dev = self.device
        copy_stream = _copy_stream(dev)
        slots = [_Slot() for _ in range(self.depth + 2)]
        held = None
        it = iter(self.batches)                            # in the consumer's thread: a DataLoader draws its base seed here
        try:
            if self.use_thread:
                free, ready, stop = queue.Queue(), queue.Queue(maxsize=self.depth), threading.Event()
                for s in slots:
                    free.put(s)
                th = threading.Thread(target=self._produce, args=(it, free, ready, copy_stream, stop), daemon=True,
                                      name="im2im-prefetch")
                th.start()
                try:
                    while True:
                        if held is not None:
                            held.released = held.released or torch.cuda.Event()
                            held.released.record(torch.cuda.current_stream(dev))
                            free.put(held)
                            held = None
                        item, slot = ready.get()
                        if isinstance(item, _Stop):
                            if item.error is not None:
                                raise item.error
                            break
                        torch.cuda.current_stream(dev).wait_event(slot.ready)
                        held = slot
                        yield item
                finally:
                    stop.set()
                    free.put(None)
                    while th.is_alive():                   # a producer blocked on a full `ready` queue
                        try:
                            ready.get(timeout=0.05)
                        except queue.Empty:
                            pass
                    th.join()
            else:
                pending = []                               # [(item on the device, slot)] in order
                free = list(slots)
                done = False
                while True:
                    if held is not None:
                        held.released = held.released or torch.cuda.Event()
                        held.released.record(torch.cuda.current_stream(dev))
                        free.append(held)
                        held = None
                    while not done and len(pending) < self.depth:
                        with _host_threads():
                            slot = free[0]
                            try:
                                nxt = self._fetch(it, slot)
                            except StopIteration:
                                done = True
                                break
                            free.pop(0)
                            pending.append((self._upload(nxt, slot, copy_stream), slot))
                    if not pending:
                        break
                    item, slot = pending.pop(0)
                    torch.cuda.current_stream(dev).wait_event(slot.ready)
                    held = slot
                    yield item
        finally:
            # the device ring belongs to the copy stream's allocator pool: whatever reuses those blocks is ordered on the copy stream,
            # which from here on runs after everything the consumer has queued so far
            copy_stream.wait_stream(torch.cuda.current_stream(dev))


def to_device(batches, device, **kw):
    """`for batch in to_device(loader, device):` -- the loader's batches with their tensors already in HBM"""
    return DevicePrefetcher(batches, device, **kw)


def sequential_slices(dataset, batch_size):
    """the batches `DataLoader(dataset, batch_size=batch_size, shuffle=False)` would yield, as views -- for a plain host TensorDataset
    (or a Subset of one over a contiguous range): batch k IS rows [k*bs, (k+1)*bs) of every tensor, so the 2 x 78 per-sample index
    operations and the stack of the default collation shrink to one slice per tensor (and one copy into the pinned ring).  None for
    anything else (a Dataset subclass may override __getitem__)."""
    from torch.utils.data import Subset, TensorDataset
    lo, hi = 0, None
    ds = dataset
    if type(ds) is Subset and isinstance(ds.indices, range) and ds.indices.step == 1:
        lo, hi, ds = ds.indices.start, ds.indices.stop, ds.dataset
    if type(ds) is not TensorDataset or not ds.tensors or any(t.is_cuda for t in ds.tensors):
        return None
    n = ds.tensors[0].shape[0]
    hi = n if hi is None else min(hi, n)
    bs = int(batch_size)
    return ([t[s:min(s + bs, hi)] for t in ds.tensors] for s in range(lo, hi, bs))


def loader(dataset, device, **dataloader_kwargs):
    """DataLoader(dataset, **dataloader_kwargs) behind a prefetcher whose `collate` stacks the samples straight into pinned staging
    memory (the loader's order, shuffling and RNG use are torch's own; only where the stacked batch is written changes)."""
    from torch.utils.data import DataLoader
    pf = DevicePrefetcher(None, device)
    if pf.active and "collate_fn" not in dataloader_kwargs:
        dataloader_kwargs = dict(dataloader_kwargs, collate_fn=pf.collate)
    pf.batches = DataLoader(dataset, **dataloader_kwargs)
    return pf
