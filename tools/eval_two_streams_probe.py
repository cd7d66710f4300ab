"""eval-mode forward of independent batches on ONE stream vs alternating between TWO streams (batches are independent in eval mode:
does overlapping one batch's HBM-bound kernels with the other's MFMA kernels pay?).  python tools/eval_two_streams_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
dev = "cuda:0"
nn_ops.set_compute_dtype("bf16")
torch.manual_seed(0)
params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
model = add_uncertainty(UNet(1, 1), params).to(dev).eval()
N, bs = 1248, 78
x = torch.randn(N, 1, 320, 320, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i, s in enumerate(range(0, N, bs)):
            if two:
                st = streams[i & 1]
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(model(x[s:s + bs]))
            else:
                outs.append(model(x[s:s + bs]))
    torch.cuda.synchronize()
    return N / (time.perf_counter() - t0), outs
for rnd in range(3):
    a, o1 = run(False)
    b, o2 = run(True)
    same = all(torch.equal(p, q) for p, q in zip(o1, o2))
    print(f"round {rnd}: one stream {a:8.1f} img/s   two streams {b:8.1f} img/s   x{b / a:.3f}   same bits {same}", flush=True)
    del o1, o2
