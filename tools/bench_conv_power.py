"""is the conv kernel power-limited?  same binary, same launch, random vs zero operands (zeros toggle no data lines)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import nn_ops
dev = "cuda:0"
b, h, w, ci, co = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "78,160,160,128,128").split(","))
for kind in ("random", "zeros", "random"):
    x = (torch.randn(b, h, w, ci, device=dev) if kind == "random" else torch.zeros(b, h, w, ci, device=dev)).to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05 if kind == "random" else torch.zeros(co, ci, 3, 3, device=dev)
    wf, wd = nn_ops.pack_weight(wt, torch.bfloat16)
    dz = (torch.randn(b, h, w, co, device=dev) if kind == "random" else torch.zeros(b, h, w, co, device=dev)).to(torch.bfloat16)
    fl = 2.0 * b * h * w * ci * co * 9
    for name, fn in (("fwd", lambda: nn_ops.conv_fwd(x, wf, None, want_stats=True)), ("wgrad", lambda: nn_ops.conv_wgrad(x, dz, 9))):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"{kind:7s} {name:6s} B{b} {h}x{w} {ci}->{co}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF")
