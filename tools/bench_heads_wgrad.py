#!/usr/bin/env python3
"""the quantile heads' weight gradient (3 planes x 32 channels, bf16) at the bench shape: ms, TB/s over its algorithmic bytes.
IM2IM_SMALLCONV_VALU=32 selects the round-2 kernel (taps split over the waves), default the round-6 one (rows split, hi+lo in one MFMA)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from im2im_uq_amd import nn_ops

b, h, w, cs, cl = int(os.environ.get("B", "78")), 320, 320, 3, 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(b, h, w, cl, device="cuda", generator=g).to(torch.bfloat16)
go = torch.randn(b, cs, h, w, device="cuda", generator=g) * 1e-3
for _ in range(3):
    dw, db = nn_ops.smallconv_wgrad(go, x, l_major=False, want_bias=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    dw, db = nn_ops.smallconv_wgrad(go, x, l_major=False, want_bias=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = b * h * w * (cl * 2 + cs * 4)
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2)[:4].transpose(0, 1), go[:4].transpose(0, 1), padding=1).transpose(0, 1)  # [cs, cl, 3, 3] of 4 images
dw4, _ = nn_ops.smallconv_wgrad(go[:4].contiguous(), x[:4].contiguous(), l_major=False, want_bias=True)
err = float((dw4.view(cs, cl, 3, 3) - ref).norm() / ref.norm())
print(f"mode {os.environ.get('IM2IM_SMALLCONV_VALU', '0'):>2s}  batch {b}: {ms:.3f} ms (3 launches: kernel + 2-stage reduce)  {nbytes / ms / 1e9:.2f} TB/s  rel err vs fp32 conv (4 images) {err:.2e}  checksum {float(dw.double().sum()):.6e}")
