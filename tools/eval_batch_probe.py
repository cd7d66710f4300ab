"""eval-mode forward throughput of the 320x320 UNet + quantile heads vs the forward batch size (calibration / validation are free
to choose it: no batch statistics in eval mode).  python tools/eval_batch_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet

dev = "cuda:0"
nn_ops.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
model = add_uncertainty(UNet(1, 1), params).to(dev).eval()
N = 624
x = torch.randn(N, 1, 320, 320, device=dev)
for rnd in range(2):
    for bs in (39, 78, 156, 312, 624):
        with torch.no_grad():
            for s in range(0, min(N, 4 * bs), bs):
                model(x[s:s + bs])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(0, N, bs):
                model(x[s:s + bs])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"round {rnd} batch {bs:4d}: {N / dt:8.1f} img/s  ({dt / (N / bs) * 1e3:.2f} ms per batch)", flush=True)
