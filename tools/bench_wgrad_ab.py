"""within-process A/B of the bf16 3x3 weight-gradient kernel: 64 vs 128 output channels per workgroup (im2im_set_option
"wgrad_co128"), BASELINE layer shapes, batch 78, interleaved rounds, median.   python tools/bench_wgrad_ab.py [batch] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from im2im_uq_amd import hip_ops, nn_ops

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 78
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
MODES = tuple(int(m) for m in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 1)
KEY = sys.argv[4] if len(sys.argv) > 4 else "wgrad_co128"
LAYERS = [(320, 64, 64, False), (320, 128, 64, True), (160, 128, 64, False), (160, 64, 128, False), (160, 128, 128, False), (160, 256, 128, True), (80, 128, 256, False), (80, 256, 256, False),
          (80, 512, 256, True), (40, 256, 512, False), (40, 512, 512, False), (40, 1024, 512, True), (20, 512, 512, False)]
tot = {m: 0.0 for m in MODES}
fl_tot = 0.0
for (h, ci, co, split) in LAYERS:
    g = torch.Generator(device=dev).manual_seed(1)
    cin = ci // 2 if split else ci
    x = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16)
    xh = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16) if split else None
    dz = torch.randn(B, h, h, co, device=dev, generator=g).to(torch.bfloat16)
    ss = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)]).contiguous()
    fl = 2.0 * B * h * h * ci * co * 9
    fn = lambda: nn_ops.conv_wgrad(x, dz, 9, x_ss=None if os.environ.get('BENCH_NOLAZY') else ss, x_hi=xh)
    times = {m: [] for m in MODES}
    outs = {}
    for m in MODES:
        hip_ops.set_option(KEY, m)
        for _ in range(2):
            outs[m] = fn()
    rel = float((outs[MODES[-1]] - outs[MODES[0]]).norm() / outs[MODES[0]].norm())
    for r in range(rounds):
        for m in MODES:
            hip_ops.set_option(KEY, m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / 3)
    med = {m: statistics.median(times[m]) for m in MODES}
    for m in MODES:
        tot[m] += med[m]
    fl_tot += fl
    print(f"wgrad {h:3d}x{h:<3d} {ci:4d}->{co:<3d} " + "   ".join(f"mode{m}: {med[m]:.3f} ms {fl / med[m] / 1e9:6.0f} TF" for m in MODES)
          + f"   x{med[MODES[0]] / med[MODES[-1]]:.3f}   rel diff {rel:.2e}", flush=True)
hip_ops.set_option(KEY, MODES[-1])
print("total: " + "   ".join(f"mode{m}: {tot[m]:.2f} ms {fl_tot / tot[m] / 1e9:6.0f} TF" for m in MODES))
