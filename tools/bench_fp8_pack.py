"""fp8 weight operands of the UNet's 3x3 layers: per-layer pack launches vs the batched tiled pack (packed_fp8), HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.trunks.unet import UNet

dev = "cuda:0"
net = UNet(2, 1).to(dev)
ws = [p for n, p in net.named_parameters() if p.dim() == 4 and p.shape[2] == 3 and p.shape[1] >= 64]
want = [w.shape[1] % 128 == 0 for w in ws]
print(len(ws), "layers,", sum(w.numel() for w in ws) / 1e6, "M weights,", sum(want), "with a data-gradient operand")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def single():
    for w, dg in zip(ws, want):
        nn_ops.pack_weight_fp8(w)
        if dg:
            nn_ops.pack_weight_fp8_dgrad(w)


def batched():
    nn_ops.invalidate_packed()
    for w, dg in zip(ws, want):
        nn_ops.packed_fp8(w, dg)


for rep in range(2):
    print(f"per-layer {timed(single):.3f} ms   batched {timed(batched):.3f} ms")
