"""layer-level timing of the fp8 forward conv against the bf16 one at BASELINE layer shapes (random data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import nn_ops

dev = "cuda:0"
shapes = [(78, 320, 320, 64, 64), (78, 160, 160, 128, 128), (78, 160, 160, 64, 128), (78, 80, 80, 256, 256), (78, 40, 40, 512, 512), (78, 40, 40, 1024, 512),
          (16, 512, 512, 64, 64)]
for b, h, w, ci, co in shapes:
    x = torch.randn(b, h, w, ci, device=dev).abs().to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device=dev) * (ci * 9) ** -0.5
    bias = torch.zeros(co, device=dev)
    ss = torch.stack([torch.ones(ci), torch.zeros(ci)]).to(dev)
    wf, _ = nn_ops.pack_weight(wt, torch.bfloat16)
    wq, ws = nn_ops.pack_weight_fp8(wt)
    fl = 2.0 * b * h * w * ci * co * 9
    res = {}
    for name, fn in (("bf16", lambda: nn_ops.conv_fwd(x, wf, bias, want_stats=True, in_ss=ss)),
                     ("fp8", lambda: nn_ops.conv_fwd_fp8(x, wq, ws, bias, want_stats=True, in_ss=ss))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[name] = (ms, fl / ms / 1e9)
    print(f"B{b} {h}x{w} {ci}->{co}: bf16 {res['bf16'][0]:.3f} ms {res['bf16'][1]:.0f} TF | fp8 {res['fp8'][0]:.3f} ms {res['fp8'][1]:.0f} TF | x{res['bf16'][0] / res['fp8'][0]:.2f}")
