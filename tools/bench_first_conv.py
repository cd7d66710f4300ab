"""first conv (1 -> 64, reads NCHW fp32, writes NHWC bf16 + BatchNorm partial statistics): time per launch at batch 78, 320x320.
    python tools/bench_first_conv.py     (run from the tree whose library is to be timed)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from im2im_uq_amd import nn_ops

dev = "cuda:0"
x = torch.randn(78, 1, 320, 320, device=dev)
w = torch.randn(64, 1, 3, 3, device=dev) * 0.3
_, wd = nn_ops.pack_weight(w, torch.float32)
bias = torch.zeros(64, device=dev)
fold = torch.stack([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)]).contiguous()
for name, fn in (("train (stats)", lambda: nn_ops.smallconv_s2l(x, wd, bias, None, 64, torch.bfloat16, flip=True, want_stats=True)),
                 ("eval (affine+relu)", lambda: nn_ops.smallconv_s2l(x, wd, None, fold, 64, torch.bfloat16, relu=True, flip=True))):
    for _ in range(3):
        out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    o = out[0] if isinstance(out, tuple) else out
    print(f"{name:20s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   checksum {float(o.float().sum()):.4f}")
