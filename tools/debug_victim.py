"""multi-process determinism harness: AGGRESSOR processes run the bf16 training step in a loop (they keep the GPU busy from other
processes), a VICTIM process launches one kernel over and over on fixed inputs and counts the launches whose bits differ from the first.
    python tools/debug_victim.py aggressor SECONDS [batch]
    python tools/debug_victim.py victim REPEATS [batch]      (IM2IM_SMALLCONV_VALU selects the first-conv kernel form)
tools/debug_victim.sh runs two aggressors and a series of victims."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def aggressor(seconds, batch):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(3)
    model = add_uncertainty(UNet(1, 1), dict(PARAMS)).to(DEV).train()
    x = torch.randn(batch, 1, 320, 320, device=DEV)
    y = torch.rand(batch, 1, 320, 320, device=DEV)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for p in model.parameters():
            p.grad = None
        model.loss_fn(model(x), y).backward()
        torch.cuda.synchronize()
        n += 1
    print(f"[aggressor] {n} steps in {time.time() - t0:.0f} s", flush=True)


def victim(reps, batch):
    from im2im_uq_amd import nn_ops
    g = torch.Generator().manual_seed(5)
    hw = 320
    x1 = torch.randn(batch, 1, hw, hw, generator=g).to(DEV)
    x3 = torch.randn(batch, 3, hw, hw, generator=g).to(DEV)
    dz64 = (torch.randn(batch, hw, hw, 64, generator=g) * 1e-6).to(DEV)
    dz32 = (torch.randn(batch, hw, hw, 32, generator=g) * 1e-6).to(DEV)
    b64, b32 = dz64.bfloat16(), dz32.bfloat16()
    a64 = torch.randn(batch, hw, hw, 64, generator=g).to(DEV).bfloat16()
    w = torch.randn(64, 64, 3, 3, generator=g).to(DEV)
    cases = {
        "first-conv wgrad bf16": lambda: nn_ops.smallconv_wgrad(x1, b64, True, False)[0],
        "first-conv wgrad f32": lambda: nn_ops.smallconv_wgrad(x1, dz64, True, False)[0],
        "heads wgrad bf16": lambda: torch.cat([t.flatten() for t in nn_ops.smallconv_wgrad(x3, b32, False, True)]),
        "3x3 wgrad bf16 64->64 (MFMA)": lambda: nn_ops.conv_wgrad(a64, b64, 9),
        "colsum bf16": lambda: nn_ops.colsum(b64),
        "torch: bf16 tensor .float().sum(dim=(0,1,2))": lambda: a64.float().sum(dim=(0, 1, 2)),
        "torch: F.conv2d fp32 1->64 (MIOpen)": lambda: torch.nn.functional.conv2d(x1, w[:, :1].contiguous(), padding=1).sum(dim=(0, 2, 3)),
    }
    only = os.environ.get("VICTIM_ONLY")
    for name, fn in cases.items():
        if only and only not in name:
            continue
        first = fn().clone()
        torch.cuda.synchronize()
        bad, where = 0, ""
        for it in range(reps):
            v = fn()
            torch.cuda.synchronize()
            if not torch.equal(v, first):
                bad += 1
                if not where:
                    nz = (v != first).flatten().nonzero().flatten()
                    where = f"; first at launch {it}: {nz.numel()} of {v.numel()} elements, flat idx {nz[:10].tolist()}, max|d| {float((v - first).abs().max()):.2e} of |v|max {float(first.abs().max()):.2e}"
        print(f"[victim VALU={os.environ.get('IM2IM_SMALLCONV_VALU', '0')}] {name}: {bad} of {reps} launches differ{where}", flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    mode, n = sys.argv[1], int(sys.argv[2])
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    aggressor(n, batch) if mode == "aggressor" else victim(n, batch)
