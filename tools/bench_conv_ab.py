"""within-process A/B of conv kernel variants on the BASELINE layer shapes (batch 78, bf16): values of one im2im_set_option
key (default "conv_splitk": 0 = off, n = aim at n * 256 workgroups), interleaved rounds, median of the per-round times.
    python tools/bench_conv_ab.py [batch] [rounds] [modes, e.g. 0,1] [option key]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from im2im_uq_amd import hip_ops, nn_ops

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 78
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
modes = [int(m) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 3]
KEY = sys.argv[4] if len(sys.argv) > 4 else "conv_splitk"
# (h, ci, co, split_in, kind)  kind: fwd = forward + statistics + lazy input; dgrad = plain store
LAYERS = [(320, 64, 64, False, "fwd"), (320, 128, 64, True, "fwd"), (320, 64, 64, False, "dgrad"), (320, 64, 128, False, "dgrad_split"),
          (160, 64, 128, False, "fwd"), (160, 128, 128, False, "fwd"), (160, 256, 128, True, "fwd"), (160, 128, 128, False, "dgrad"),
          (80, 256, 256, False, "fwd"), (80, 512, 256, True, "fwd"), (40, 512, 512, False, "fwd"), (40, 1024, 512, True, "fwd"),
          (20, 512, 512, False, "fwd")]
tot = {m: 0.0 for m in modes}
flops_tot = 0.0
for (h, ci, co, split, kind) in LAYERS:
    g = torch.Generator(device=dev).manual_seed(1)
    cin = ci // 2 if split else ci
    x = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16)
    xh = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16) if split else None
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
    wf, _ = nn_ops.pack_weight(wt, torch.bfloat16)
    ss = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)]).contiguous()
    fl = 2.0 * B * h * h * ci * co * 9
    if kind == "fwd":
        fn = lambda: nn_ops.conv_fwd(x, wf, None, want_stats=True, in_ss=ss, x_hi=xh)
    elif kind == "dgrad_split":
        fn = lambda: nn_ops.conv_fwd(x, wf, split_out=co // 2)
    else:
        fn = lambda: nn_ops.conv_fwd(x, wf)
    times = {m: [] for m in modes}
    for m in modes:
        hip_ops.set_option(KEY, m)
        for _ in range(2):
            fn()
    for r in range(rounds):
        for m in modes:
            hip_ops.set_option(KEY, m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / 3)
    med = {m: statistics.median(times[m]) for m in modes}
    for m in modes:
        tot[m] += med[m]
    flops_tot += fl
    print(f"{kind:11s} {h:3d}x{h:<3d} {ci:4d}->{co:<3d} " + "  ".join(f"mode{m}: {med[m]:.3f} ms {fl / med[m] / 1e9:6.0f} TF" for m in modes)
          + (f"   x{med[modes[0]] / med[modes[-1]]:.3f}" if len(modes) > 1 else ""), flush=True)
    del x, xh
hip_ops.set_option(KEY, modes[0])
print("total: " + "  ".join(f"mode{m}: {tot[m]:.2f} ms {flops_tot / tot[m] / 1e9:6.0f} TF" for m in modes))
