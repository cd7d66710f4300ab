#!/usr/bin/env python3
"""one configuration of train_net with hip_graph=True per process (a crash takes the process down): which ingredient of
tests/test_round6_gpu.py::test_train_net_with_prefetcher...[bf16-True] makes hipStreamEndCapture fall over?
    python tools/debug_graph_crash.py BASE N_IMAGES EPOCHS PREFETCH"""
import faulthandler
import os
import sys

import torch
from torch.utils.data import TensorDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
base, n, epochs, pf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
from im2im_uq_amd import nn_ops, prefetch
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
from im2im_uq_amd.core.scripts import train as tr

prefetch.ENABLED = bool(pf)
var = sys.argv[5] if len(sys.argv) > 5 else ""
if var == "noval":
    tr.run_validation = lambda *a, **k: None
elif var == "noimages":
    def _raise(*a, **k):
        raise RuntimeError("skipped")
    tr.get_images = _raise
elif var == "noevalnet":
    tr.eval_net = lambda *a, **k: 0.0
elif var == "nogc":
    import gc
    gc.disable()
elif var == "threadlocal":
    import torch.cuda.graphs as G
    orig = G.graph.__init__

    def init(self, *a, **k):
        k["capture_error_mode"] = "thread_local"
        orig(self, *a, **k)
    G.graph.__init__ = init
elif var == "noside":
    nn_ops.WGRAD_SIDE_STREAM = False
elif var == "nodefer":
    nn_ops.WGRAD_DEFER_REDUCE = False
elif var == "sync":
    _orig_capture = tr.GraphedStep._capture

    def _cap(self, cur):
        nn_ops.join_side_streams()
        torch.cuda.synchronize()
        return _orig_capture(self, cur)
    tr.GraphedStep._capture = _cap
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=50, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              device="cuda:0", dataset="synthetic", batch_size=6, lr=1e-3, input_normalization="standard",
              output_normalization="min-max", num_validation_images=2, hip_graph=True)
nn_ops.set_compute_dtype("bf16")
g = torch.Generator().manual_seed(8)
x, y = torch.randn(n, 1, 32, 32, generator=g), torch.rand(n, 1, 32, 32, generator=g)
torch.manual_seed(5)
net = add_uncertainty(UNet(1, 1, depth=2, base=base), dict(PARAMS))
net = tr.train_net(net, TensorDataset(x, y), TensorDataset(x[:4], y[:4]), "cuda:0", epochs, 6, 1e-3, False, None, 100, 100, PARAMS)
torch.cuda.synchronize()
print(f"OK base={base} n={n} epochs={epochs} prefetch={pf}")
