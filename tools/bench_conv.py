"""micro-benchmark of single conv shapes (forward kernel / wgrad) for tuning; not part of the product path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import nn_ops
dev = "cuda:0"
shapes = [(16, 160, 160, 128, 128), (16, 320, 320, 64, 64), (16, 80, 80, 256, 256), (16, 40, 40, 512, 512), (16, 320, 320, 128, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(","))]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for (b, h, w, ci, co) in shapes:
    x = torch.randn(b, h, w, ci, device=dev).to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    wf, wd = nn_ops.pack_weight(wt, torch.bfloat16)
    dz = torch.randn(b, h, w, co, device=dev).to(torch.bfloat16)
    fl = 2.0 * b * h * w * ci * co * 9
    for name, fn in (("fwd", lambda: nn_ops.conv_fwd(x, wf, None, want_stats=True)), ("wgrad", lambda: nn_ops.conv_wgrad(x, dz, 9))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:6s} B{b} {h}x{w} {ci}->{co}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF")
