#!/usr/bin/env python3
"""Eval forward: do the two full-resolution blocks (inc: 1 -> 64 -> 64 + pool; up4: upsample + 128 -> 64 -> 64 + OutConv tail) run faster
in sub-batches whose 64-channel intermediates (13 MB per image) stay in the 256 MB Infinity Cache between producer and consumer,
while the deep levels keep the whole batch?  Times each block on the whole batch of 78 and in sub-batches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet

nn_ops.set_compute_dtype("bf16")
P = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
torch.manual_seed(0)
model = add_uncertainty(UNet(1, 1), dict(P)).cuda()
net = model.baseModel
B = 78
x = torch.randn(B, 1, 320, 320, device="cuda")
with torch.no_grad():
    model.train(); model(x[:8]); model.eval()
    # intermediates of one whole-batch forward
    skips = []
    feat, pooled = net.inc(x, lazy=True, pool=True); skips.append(feat)
    for i in range(1, 4):
        feat, pooled = getattr(net, f"down{i}")(pooled, lazy=True, pool=True, pooled=True); skips.append(feat)
    h = net.down4(pooled, lazy=True, pooled=True)
    for k in range(1, 4):
        h = getattr(net, f"up{k}")(h, skips[4 - k], lazy=True)
    x1 = skips[0]

    def timed(fn, reps=8):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for sub in (78, 39, 26, 13, 6):
        t_inc = timed(lambda: [net.inc(x[s:s + sub], lazy=True, pool=True) for s in range(0, B, sub)])
        t_up = timed(lambda: [net.up4(h[s:s + sub], x1[s:s + sub], lazy=True, tail=net.out.conv) for s in range(0, B, sub)])
        t_all = timed(lambda: model(x))
        print(f"sub-batch {sub:3d}: inc block {t_inc:6.3f} ms   up4 block (+ OutConv tail) {t_up:6.3f} ms   [whole forward at batch 78: {t_all:6.3f} ms]")
