"""time the first conv's weight gradient (smallconv_wgrad, CS = 1, CL = 64) alone: python tools/bench_firstconv_wgrad.py [batch ...]
A/B through IM2IM_SMALLCONV_VALU (bit 128: bf16 kept in LDS; bit 8: the one-channel kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from im2im_uq_amd import nn_ops  # noqa: E402

for b in [int(v) for v in sys.argv[1:]] or [78, 10]:
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(b, 1, 320, 320, device="cuda")
        dz = torch.randn(b, 320, 320, 64, device="cuda").to(dt)
        for _ in range(5):
            nn_ops.smallconv_wgrad(x, dz, True, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            nn_ops.smallconv_wgrad(x, dz, True, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        gb = (dz.numel() * dz.element_size() + x.numel() * 4) / 1e9
        print(f"[fcw] IM2IM_SMALLCONV_VALU={os.environ.get('IM2IM_SMALLCONV_VALU', '0')} batch {b} {str(dt)[6:]}: {ms:.4f} ms ({gb / ms * 1e3:.0f} GB/s incl. reduce launches)", flush=True)
