#!/usr/bin/env python3
"""Where does a host batch's time go?  Stage timings of the reference's data path (TensorDataset in pageable memory ->
DataLoader(num_workers=0) -> .to(device)) and of the prefetcher's, for one batch of 78 x [1,320,320] fp32 (x and y: 64 MB).

    python tools/host_pipe_probe.py
"""
import os
import sys
import time

import torch
from torch.utils.data import DataLoader, TensorDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def t(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    B, hw, nb = 78, 320, 12
    n = B * nb
    print(f"cpus {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, torch threads {torch.get_num_threads()}, interop {torch.get_num_interop_threads()}")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 1, hw, hw, generator=g)
    y = torch.rand(n, 1, hw, hw, generator=g)
    ds = TensorDataset(x, y)
    idx = list(range(B, 2 * B))
    mb = 2 * B * hw * hw * 4 / 1e6
    samples = [ds[i] for i in idx]
    print(f"one batch = {mb:.1f} MB (x and y)")
    print(f"fetch 78 samples (ds[i])                  {t(lambda: [ds[i] for i in idx]):8.2f} ms")
    print(f"default_collate (torch.stack, new tensor) {t(lambda: torch.utils.data.default_collate(samples)):8.2f} ms")
    px, py = torch.empty(B, 1, hw, hw).pin_memory(), torch.empty(B, 1, hw, hw).pin_memory()
    print(f"stack into pinned out=                    {t(lambda: (torch.stack([s[0] for s in samples], out=px), torch.stack([s[1] for s in samples], out=py))):8.2f} ms")
    cx, cy = torch.utils.data.default_collate(samples)
    print(f"pinned.copy_(pageable)                    {t(lambda: (px.copy_(cx), py.copy_(cy))):8.2f} ms")
    it = torch.tensor(idx)
    print(f"index_select into pinned out=             {t(lambda: (torch.index_select(x, 0, it, out=px), torch.index_select(y, 0, it, out=py))):8.2f} ms")
    print(f"contiguous slice copy into pinned         {t(lambda: (px.copy_(x[B:2 * B]), py.copy_(y[B:2 * B]))):8.2f} ms")
    for th in (1, 4, 8, 16):
        torch.set_num_threads(th)
        print(f"  threads={th:2d}: collate {t(lambda: torch.utils.data.default_collate(samples)):7.2f} ms, pinned.copy_ {t(lambda: (px.copy_(cx), py.copy_(cy))):7.2f} ms, "
              f"index_select->pinned {t(lambda: (torch.index_select(x, 0, it, out=px), torch.index_select(y, 0, it, out=py))):7.2f} ms")
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    dx, dy = torch.empty(B, 1, hw, hw, device=dev), torch.empty(B, 1, hw, hw, device=dev)

    def h2d():
        dx.copy_(px, non_blocking=True); dy.copy_(py, non_blocking=True); torch.cuda.synchronize()
    print(f"H2D from pinned (64 MB, sync)             {t(h2d):8.2f} ms")

    def h2d_pageable():
        cx.to(dev); cy.to(dev); torch.cuda.synchronize()
    print(f"H2D .to(device) from pageable             {t(h2d_pageable):8.2f} ms")
    # whole loaders, no model
    from im2im_uq_amd import prefetch
    for name, mk in (("DataLoader alone (host)", lambda: DataLoader(ds, batch_size=B, shuffle=True, num_workers=0)),
                     ("DataLoader + .to(device)", None),
                     ("prefetcher, thread", lambda: prefetch.DevicePrefetcher(DataLoader(ds, batch_size=B, shuffle=True, num_workers=0), dev, thread=True)),
                     ("prefetcher, no thread", lambda: prefetch.DevicePrefetcher(DataLoader(ds, batch_size=B, shuffle=True, num_workers=0), dev, thread=False))):
        for rep in range(2):
            t0 = time.perf_counter()
            if mk is None:
                for b in DataLoader(ds, batch_size=B, shuffle=True, num_workers=0):
                    b = [v.to(dev) for v in b]
            else:
                for b in mk():
                    pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name:28s} {dt / nb * 1e3:8.2f} ms per batch  ({n / dt:7.0f} img/s)")


if __name__ == "__main__":
    main()
