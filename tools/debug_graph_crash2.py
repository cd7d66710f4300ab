#!/usr/bin/env python3
"""bisect of the capture crash after an eager odd-shaped step (tools/debug_graph_crash.py): GraphedStep driven directly.
    python tools/debug_graph_crash2.py VARIANT
variants: none (3 warm steps, capture)            odd (warm, eager odd step on the default stream, capture)
          odd_fwd (forward only)                  odd_nostep (forward + backward, no optimizer step)
          odd_same (eager step of the SAME shape on the default stream)
          odd_onstream (the odd step on the capture stream)   odd_sync (odd + synchronize + empty_cache before the capture)
          odd_noside (odd, weight-gradient stream off)         odd_threadlocal (capture_error_mode thread_local)"""
import faulthandler
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
v = sys.argv[1]
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
from im2im_uq_amd.core.scripts.train import GraphedStep

if v == "odd_noside":
    nn_ops.WGRAD_SIDE_STREAM = False
P = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
nn_ops.set_compute_dtype("bf16")
g = torch.Generator().manual_seed(8)
x, y = torch.randn(6, 1, 32, 32, generator=g).cuda(), torch.rand(6, 1, 32, 32, generator=g).cuda()
torch.manual_seed(5)
net = add_uncertainty(UNet(1, 1, depth=2, base=32), dict(P)).cuda().train()
opt = nn_ops.FusedAdam(net.parameters(), lr=1e-3)
gs = GraphedStep(net, opt)
for _ in range(3):
    gs.step((x,), y)


def eager(xx, yy, bwd=True, step=True):
    pred = net(xx)
    loss = net.loss_fn(pred, yy)
    if bwd:
        opt.zero_grad()
        loss.backward()
        if step:
            opt.step()


if v.startswith("odd") or v.startswith("val"):
    xx, yy = (x, y) if v == "odd_same" else (x[:2], y[:2])
    if v == "odd_onstream":
        gs.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(gs.stream):
            eager(xx, yy)
        torch.cuda.current_stream().wait_stream(gs.stream)
    else:
        eager(xx, yy, bwd=v != "odd_fwd", step=v != "odd_nostep")
    if v == "odd_sync":
        nn_ops.join_side_streams()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
if v.startswith("odd_val"):
    # what train_net does between the odd step and the next epoch: validation forwards in eval mode, mode flips
    what = v[len("odd_val"):]
    with torch.no_grad():
        net.eval()
        if what in ("", "_e2", "_e2e4"):
            net(x[:2])
        if what in ("", "_e4", "_e2e4"):
            net(x[:4])
        if what in ("", "_loss"):
            p4 = net(x[:4])
            net.loss_fn(p4, y[:4])
        if what == "":
            _ = float(net.loss_fn(net(x[:4]), y[:4]))
    net.train()
if v == "odd_threadlocal":
    import torch.cuda.graphs as G
    orig = G.graph.__init__

    def init(self, *a, **k):
        k["capture_error_mode"] = "thread_local"
        orig(self, *a, **k)
    G.graph.__init__ = init
out = gs.step((x,), y)
torch.cuda.synchronize()
print(f"OK {v}: graph={'yes' if gs.graph is not None else 'no'} failed={gs.failed} {getattr(gs, 'error', '')} loss={float(out) if out is not None else None}")
for _ in range(2):
    gs.step((x,), y)
torch.cuda.synchronize()
print("replays fine")
