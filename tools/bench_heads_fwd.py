#!/usr/bin/env python3
"""the quantile heads' forward (32 channels -> 3 planes, bf16) and data-gradient at the bench shape: ms and TB/s over the algorithmic bytes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from im2im_uq_amd import nn_ops

b, h, w, cs, cl = int(os.environ.get("B", "78")), 320, 320, 3, 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(b, h, w, cl, device="cuda", generator=g).to(torch.bfloat16)
wt = torch.randn(cs, cl, 3, 3, device="cuda", generator=g) * 0.1
bias = torch.randn(cs, device="cuda", generator=g)
wf, _ = nn_ops.pack_weight(wt, torch.float32, want_wd=False)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: nn_ops.smallconv_l2s(x, wf, bias, cs))
out = nn_ops.smallconv_l2s(x[:4].contiguous(), wf, bias, cs)
ref = torch.nn.functional.conv2d(x[:4].float().permute(0, 3, 1, 2), wt, bias, padding=1)
err = float((out - ref).norm() / ref.norm())
nbytes = b * h * w * (cl * 2 + cs * 4)
print(f"heads forward  batch {b}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s  rel err vs fp32 conv {err:.2e}")
