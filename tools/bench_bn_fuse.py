"""per-layer A/B of the BatchNorm-backward reduction: separate pass vs accumulated in the data-gradient epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import nn_ops
dev = "cuda:0"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
# (B, H, W, channels of dz, channels of dx/z, taps)
for (b, h, w, ci, co, taps) in [(78, 320, 320, 64, 64, 9), (78, 160, 160, 128, 128, 9), (78, 80, 80, 256, 256, 9), (78, 40, 40, 512, 512, 9),
                                (78, 320, 320, 32, 64, 1), (78, 160, 160, 64, 128, 9)]:
    dz = torch.randn(b, h, w, ci, device=dev).to(torch.bfloat16)
    wt = torch.randn(ci, co, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=dev) * 0.05
    _, wd = nn_ops.pack_weight(wt, torch.bfloat16)
    z = torch.randn(b, h, w, co, device=dev).to(torch.bfloat16)
    ss = torch.stack([torch.ones(co), torch.zeros(co)]).to(dev)
    mi = torch.stack([torch.zeros(co), torch.ones(co)]).to(dev)
    t_plain = timeit(lambda: nn_ops.conv_fwd(dz, wd))
    t_fused = timeit(lambda: nn_ops.conv_dgrad_bn(dz, wd, z, ss, mi))
    dx, partial = nn_ops.conv_dgrad_bn(dz, wd, z, ss, mi)
    t_full = timeit(lambda: nn_ops.bn_relu_bwd(dx, z, ss, mi))
    t_part = timeit(lambda: nn_ops.bn_relu_bwd_from_partial(dx, z, ss, mi, partial))
    print(f"B{b} {h}x{w} {ci}->{co} taps{taps}: dgrad {t_plain:.3f} fused {t_fused:.3f} (+{t_fused - t_plain:.3f}) | bn_bwd full {t_full:.3f} from-partial {t_part:.3f} (-{t_full - t_part:.3f}) | net gain {t_plain + t_full - t_fused - t_part:+.3f} ms")
