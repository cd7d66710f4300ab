#!/usr/bin/env python3
"""Who launches what in one training step: kernel names with counts and GPU time (torch.profiler), and for the torch-owned
launches (fill / copy / cat / elementwise) the Python frames that issued them.  Used to hunt the ~115 small launches per step
(VERDICT r5 weak #4).

    python tools/launch_census.py [--batch 78] [--size 320] [--steps 2] [--stacks]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=78)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--stacks", action="store_true")
    args = ap.parse_args()
    import bench
    from im2im_uq_amd import launch
    dist, rank, world, dev, backend = launch.init_distributed(expected_world=1)
    job = bench.Job(dist, rank, world, dev, backend)
    conf = dict(bench.CONFIGS["fastmri"], batch=args.batch, size=args.size)
    wl = bench.Workload(job, conf)
    wl.graphed = False
    wl.model.train()
    for _ in range(3):
        wl.train_step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=args.stacks) as prof:
        for _ in range(args.steps):
            wl.train_step()
        torch.cuda.synchronize()
    ev = prof.events()
    kern = collections.defaultdict(lambda: [0, 0.0])
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA:
            k = kern[e.name[:110]]
            k[0] += 1
            k[1] += e.device_time
    tot_n = sum(v[0] for v in kern.values())
    tot_t = sum(v[1] for v in kern.values())
    print(f"# batch {args.batch}, {args.size}x{args.size}: {tot_n / args.steps:.1f} launches / step, {tot_t / args.steps / 1e3:.3f} ms GPU time / step")
    print(f"{'launches/step':>14} {'us/step':>10} {'avg us':>8}  kernel")
    for name, (n, t) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
        print(f"{n / args.steps:14.1f} {t / args.steps:10.1f} {t / n:8.1f}  {name}")
    small = sum(n for n, t in kern.values() if t / n < 30.0) / args.steps
    small_t = sum(t for n, t in kern.values() if t / n < 30.0) / args.steps
    print(f"# launches averaging < 30 us: {small:.1f} / step, {small_t / 1e3:.3f} ms / step")
    # torch-owned launches: which aten op (and which Python frame) issues them
    ops = collections.defaultdict(int)
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.name in (
                "aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::mul", "aten::add_", "aten::add", "aten::ones_like",
                "aten::zeros", "aten::to", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::sum", "aten::div", "aten::mul_"):
            stack = ""
            if args.stacks and e.stack:
                fr = [s for s in e.stack if "im2im_uq_amd" in s or "bench.py" in s]
                stack = " <- " + " | ".join(f.split("/")[-1] for f in fr[:3])
            ops[e.name + stack] += 1
    print("# aten ops on the host side (per step):")
    for k, n in sorted(ops.items(), key=lambda kv: -kv[1])[:60]:
        print(f"{n / args.steps:8.1f}  {k}")


if __name__ == "__main__":
    main()
