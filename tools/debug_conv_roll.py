"""conv_roll64_kernel vs conv_igemm_kernel on the same inputs (option conv_roll 1 / 0): where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import hip_ops, nn_ops
dev = "cuda:0"
BF = torch.bfloat16
cases = [(1, 64, 64, 96, 64, False, False), (1, 32, 16, 64, 64, False, False), (2, 64, 32, 64, 64, False, True), (3, 96, 48, 128, 64, True, True),
         (8, 320, 320, 64, 64, False, True), (1, 64, 64, 32, 128 + 64, False, False)]
for (b, h, w, ci, co, split, lazy) in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    cin = ci // 2 if split else ci
    x = torch.randn(b, h, w, cin, device=dev, generator=g).to(BF)
    xh = torch.randn(b, h, w, cin, device=dev, generator=g).to(BF) if split else None
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) * (ci * 9) ** -0.5
    wf, _ = nn_ops.pack_weight(wt, BF)
    ss = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)]).contiguous() if lazy else None
    bias = torch.randn(co, device=dev) * 0.1
    outs = {}
    for mode in (0, 1):
        hip_ops.set_option("conv_roll", mode)
        y, st = nn_ops.conv_fwd(x, wf, bias, want_stats=True, in_ss=ss, x_hi=xh)
        torch.cuda.synchronize()
        outs[mode] = (y.float(), st.clone())
    y0, y1 = outs[0][0], outs[1][0]
    d = (y1 - y0).abs()
    rel = float((y1 - y0).norm() / y0.norm())
    print(f"case B{b} {h}x{w} {ci}->{co} split={split} lazy={lazy}: rel {rel:.4g}  max {float(d.max()):.4g}  stats rel {float((outs[1][1] - outs[0][1]).norm() / outs[0][1].norm()):.3g}")
    if rel > 1e-2:
        bad = d > 0.05
        print("   bad fraction", float(bad.float().mean()))
        print("   by image   ", [round(float(bad[i].float().mean()), 3) for i in range(b)])
        print("   by row%32  ", [round(float(bad[:, r::32].float().mean()), 2) for r in range(32)])
        print("   by col%16  ", [round(float(bad[:, :, c::16].float().mean()), 2) for c in range(16)])
        print("   by tile row", [round(float(bad[:, r * 32:(r + 1) * 32].float().mean()), 2) for r in range(h // 32)])
        print("   by tile col", [round(float(bad[:, :, c * 16:(c + 1) * 16].float().mean()), 2) for c in range(w // 16)])
        print("   by chan%64 ", [round(float(bad[..., c::64].float().mean()), 2) for c in range(0, 64, 4)])
hip_ops.set_option("conv_roll", 1)
