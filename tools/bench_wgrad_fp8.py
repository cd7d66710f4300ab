"""bf16 vs fp8 weight gradient (conv_wgrad_pipe_kernel vs conv_wgrad_fp8_kernel) on the BASELINE layer shapes, interleaved rounds.
    python tools/bench_wgrad_fp8.py [batch] [size] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from im2im_uq_amd import nn_ops

dev = "cuda:0"
for kv in os.environ.get("BENCH_OPT", "").split(","):          # e.g. BENCH_OPT=wgrad_fp8_co128=0
    if "=" in kv:
        from im2im_uq_amd import hip_ops
        hip_ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 78
S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
LAYERS = [(1, 128, 64, True), (2, 128, 64, False), (2, 128, 128, False), (2, 256, 128, True), (4, 128, 256, False), (4, 256, 256, False),
          (4, 512, 256, True), (8, 256, 512, False), (8, 512, 512, False), (8, 1024, 512, True), (16, 512, 512, False)]
tot = {"bf16": 0.0, "fp8": 0.0}
fl_tot = 0.0
for (div, ci, co, split) in LAYERS:
    h = S // div
    g = torch.Generator(device=dev).manual_seed(1)
    cin = ci // 2 if split else ci
    x = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16)
    xh = torch.randn(B, h, h, cin, device=dev, generator=g).to(torch.bfloat16) if split else None
    dz = (torch.randn(B, h, h, co, device=dev, generator=g) * 1e-4).to(torch.bfloat16)
    amax = dz.float().abs().max().reshape(1).contiguous()
    ss = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)]).contiguous()
    fl = 2.0 * B * h * h * ci * co * 9
    fns = {"bf16": lambda: nn_ops.conv_wgrad(x, dz, 9, x_ss=ss, x_hi=xh),
           "fp8": lambda: nn_ops.conv_wgrad_fp8(x, dz, amax.data_ptr(), x_ss=ss, x_hi=xh)}
    outs = {k: f() for k, f in fns.items()}
    rel = float((outs["fp8"] - outs["bf16"]).norm() / outs["bf16"].norm())
    times = {k: [] for k in fns}
    for r in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 3)
    med = {k: statistics.median(v) for k, v in times.items()}
    for k in med:
        tot[k] += med[k]
    fl_tot += fl
    print(f"wgrad {h:3d}x{h:<3d} {ci:4d}->{co:<3d} bf16 {med['bf16']:.3f} ms {fl / med['bf16'] / 1e9:6.0f} TF   fp8 {med['fp8']:.3f} ms {fl / med['fp8'] / 1e9:6.0f} TF"
          f"   x{med['bf16'] / med['fp8']:.3f}   rel diff {rel:.3f}", flush=True)
print(f"total: bf16 {tot['bf16']:.2f} ms {fl_tot / tot['bf16'] / 1e9:6.0f} TF   fp8 {tot['fp8']:.2f} ms {fl_tot / tot['fp8'] / 1e9:6.0f} TF   x{tot['bf16'] / tot['fp8']:.3f}")
