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


class _Slot:
    """staging (pinned) and device buffers of one in-flight batch, and its two hand-off events"""
    __slots__ = ("pin", "dev", "ready", "released", "keep")

    def __init__(self):
        self.pin, self.dev = [], []
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

    def __len__(self):
        return len(self.batches)

    # ------------------------------------------------------------------ producer side
    def _upload(self, item, slot, copy_stream):
        """stage + enqueue the uploads of one batch; returns the batch with device tensors (views of the slot's ring buffers)"""
        dev = self.device
        k = [0]
        slot.keep = []
        if slot.ready is not None:
            slot.ready.synchronize()                       # the uploads that last read this slot's staging buffers (long done)

        def pinned(n, dtype):
            return torch.empty(n, dtype=dtype).pin_memory()

        def device(n, dtype):
            return torch.empty(n, dtype=dtype, device=dev)

        with torch.cuda.stream(copy_stream):
            if slot.released is not None:
                copy_stream.wait_event(slot.released)     # the consumer's kernels that read the slot's previous contents

            def move(t):
                if t.is_cuda:
                    return t
                i = k[0]
                k[0] += 1
                src = t.detach()
                if src.is_pinned() and src.is_contiguous():
                    slot.keep.append(src)                  # stays referenced until the slot's next `ready.synchronize()`
                else:
                    stage = _Slot._fit(slot.pin, i, src.shape, src.dtype, pinned)
                    stage.copy_(src)                       # host memcpy (the GIL is released inside)
                    src = stage
                out = _Slot._fit(slot.dev, i, src.shape, src.dtype, device)
                out.copy_(src, non_blocking=True)
                return out

            out = _map_tensors(item, move)
            if slot.ready is None:
                slot.ready = torch.cuda.Event()
            slot.ready.record(copy_stream)
        return out

    def _produce(self, it, free, ready, copy_stream, stop):
        try:
            torch.cuda.set_device(self.device)
            for item in it:
                slot = free.get()
                if slot is None or stop.is_set():
                    return
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
        copy_stream = torch.cuda.Stream(device=dev)
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
                        try:
                            nxt = next(it)
                        except StopIteration:
                            done = True
                            break
                        slot = free.pop(0)
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
