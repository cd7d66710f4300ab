"""one conv layer, repeated (driver for rocprofv3 passes):  python tools/conv_one.py B,H,W,Ci,Co kind reps
kind: fwd (lazy input + statistics) | dgrad (plain) | fwd_split;  env IM2IM_CONV_ROLL=0/1 selects the kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import hip_ops, nn_ops
dev = "cuda:0"
b, h, w, ci, co = (int(v) for v in sys.argv[1].split(","))
kind = sys.argv[2] if len(sys.argv) > 2 else "dgrad"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hip_ops.set_option("conv_roll", int(os.environ.get("IM2IM_CONV_ROLL", "1")))
split = kind == "fwd_split"
cin = ci // 2 if split else ci
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(b, h, w, cin, device=dev, generator=g).to(torch.bfloat16)
xh = torch.randn(b, h, w, cin, device=dev, generator=g).to(torch.bfloat16) if split else None
wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
wf, _ = nn_ops.pack_weight(wt, torch.bfloat16)
ss = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)]).contiguous()
fn = (lambda: nn_ops.conv_fwd(x, wf)) if kind == "dgrad" else (lambda: nn_ops.conv_fwd(x, wf, None, want_stats=True, in_ss=ss, x_hi=xh))
for _ in range(2):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{kind} B{b} {h}x{w} {ci}->{co} roll={os.environ.get('IM2IM_CONV_ROLL', '1')}: {ms:.3f} ms {2.0 * b * h * w * ci * co * 9 / ms / 1e9:.0f} TF")
