"""bf16 (and fp8) parity on weights that mean something: the reference's DEFAULT initialisation
(nn.Conv2d kaiming-uniform / nn.BatchNorm2d defaults, restated and seeded in oracle.model.default_init_state) and the same
network after 50 fp32 Adam steps on a learnable synthetic task -- not the closed-form det_state weights of the other parity
tests, whose |channel mean| >> spread makes train-mode BatchNorm amplify bf16 storage rounding to tens of percent in ANY
implementation (tests/test_bf16_parity_gpu.py).

One train step (reference: core/scripts/train.py:141-165) at the benchmarked sizes: 320x320 (B = 4, the fastMRI config) in the
bf16 mode, 512x512 with two input channels (B = 2, the BSBCM config) in the fp8 mode.  Three evaluations of the same step:
the oracle in fp32 (the reference's arithmetic), the oracle at the storage / operand precision of the mode
(model_forward(emulate_bf16=True | "fp8")), and the HIP path.  Reported and asserted: outputs, loss and per-tensor gradients of
the HIP path vs fp32 AND vs the emulation, relative to the emulation's own distance from fp32, plus absolute ceilings
<= 1.3x the values measured on MI355X (profiles/r04_bf16_parity_default_init.txt)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _task(b, n_in, h, w, seed):
    """noise images and a target the network can learn: a smoothed, squashed function of the input (values in [0, 1])."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, n_in, h, w, generator=g)
    y = torch.sigmoid(2.0 * F.avg_pool2d(x.mean(1, keepdim=True), 5, stride=1, padding=2) * 3.0)
    return x, y


def _oracle_step(state, x, y, emulate):
    from oracle import model as om
    leaves = {k: v.clone().requires_grad_(True) for k, v in state.items() if om.is_param(k)}
    work = {k: v.clone() for k, v in state.items()}
    work.update(leaves)
    pred = om.model_forward(x, work, training=True, emulate_bf16=emulate)
    loss = om.quantile_loss(pred, y, PARAMS)
    loss.backward()
    return pred.detach(), float(loss.detach()), {k: v.grad for k, v in leaves.items()}


def _hip_step(state, x, y, n_in, mode):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype(mode)
    try:
        model = add_uncertainty(UNet(n_in, 1), dict(PARAMS))
        model.load_state_dict(state)
        model = model.to(DEV).train()
        pred = model(x.to(DEV))
        loss = model.loss_fn(pred, y.to(DEV))
        loss.backward()
        nn_ops.join_side_streams()
        torch.cuda.synchronize()
        return pred.detach().cpu(), float(loss), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    finally:
        nn_ops.set_compute_dtype("bf16")


def _trained_state(state, n_in, h, w, steps=50):
    """`steps` Adam steps (lr 1e-3) in the fp32 parity mode of the HIP path (itself pinned to the oracle: tests/test_model_gpu.py)
    on fresh batches of the task -> the state_dict of a network that has started to fit."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("fp32")
    try:
        model = add_uncertainty(UNet(n_in, 1), dict(PARAMS))
        model.load_state_dict(state)
        model = model.to(DEV).train()
        opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
        first = last = None
        for i in range(steps):
            x, y = _task(2, n_in, h, w, seed=1000 + i)
            loss = model.loss_fn(model(x.to(DEV)), y.to(DEV))
            opt.zero_grad()
            loss.backward()
            opt.step()
            first = float(loss) if first is None else first
            last = float(loss)
        torch.cuda.synchronize()
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, first, last
    finally:
        nn_ops.set_compute_dtype("bf16")


# name, mode, emulation flag, n_in, B, H, W, and per stage ("init" / "trained") the absolute ceilings
#   (outputs vs fp32, loss rel, gradient median vs fp32, gradient worst resolved tensor vs fp32): <= 1.3x what was measured on
#   MI355X for the HIP path (profiles/r04_bf16_parity_default_init.txt; the CPU emulation measured the same to two digits):
#     fastmri_320_bf16  init    outputs 0.0351  loss 3.0e-4  gradients median 0.204  worst 0.470
#                       trained outputs 0.0015  loss 3.0e-4  gradients median 0.042  worst 0.116
#     bsbcm_512x2_fp8   init    outputs 0.212   loss 1.6e-4  gradients median 0.770  worst resolved 0.814 (28 of 62 tensors unresolved)
#                       trained outputs 0.0073  loss 7.8e-3  gradients median 0.198  worst resolved 0.431 (1 unresolved)
#   i.e. at INITIALISATION (whichever: closed-form or torch's default) train-mode BatchNorm amplifies storage rounding to percents
#   of the outputs and tens of percent of the gradients in any implementation; fifty optimiser steps later the same network is
#   20x (outputs) / 5x (gradients) closer to fp32.
CASES = [
    # [r5] the "trained" ceilings are 2x the round-4 measurements, not 1.3x: the trained state comes out of 50 steps of the HIP path's
    # own fp32 mode, so ANY change of a summation order upstream (round 5: the BatchNorm-backward sums moved into the data-gradient
    # epilogue by default) lands on another, equally valid set of weights, and on those the EMULATION's own distance from fp32 moved
    # from 0.037 to 0.063 (median gradient) with the HIP path still 1.07x the emulation.  The relative assertions above are the parity
    # statement; these only catch a mode that got grossly worse.
    ("fastmri_320_bf16", "bf16", True, 1, 4, 320, 320, {"init": (0.046, 1e-3, 0.265, 0.62), "trained": (0.0031, 1e-3, 0.085, 0.40)}),
    ("bsbcm_512x2_fp8", "fp8", "fp8", 2, 2, 512, 512, {"init": (0.276, 1e-3, 1.0, 1.06), "trained": (0.015, 1.7e-2, 0.40, 0.86)}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_low_precision_step_on_default_init_and_trained_weights(case):
    from oracle import model as om
    name, mode, emulate, n_in, b, h, w, ceilings = case
    torch.set_num_threads(min(32, torch.get_num_threads()))
    x, y = _task(b, n_in, h, w, seed=77)
    init = om.default_init_state(n_in, 1, seed=5)
    trained, l_first, l_last = _trained_state(init, n_in, h, w)
    print(f"\n[{name}] 50 fp32 Adam steps on the task: loss {l_first:.4f} -> {l_last:.4f}")
    assert l_last < 0.8 * l_first                                   # the weights moved somewhere meaningful
    for stage, state in (("init", init), ("trained", trained)):
        ref_pred, ref_loss, ref_g = _oracle_step(state, x, y, False)
        emu_pred, emu_loss, emu_g = _oracle_step(state, x, y, emulate)
        pred, loss, grads = _hip_step(state, x, y, n_in, mode)
        e_hip, e_emu, e_pair = rel_l2(pred, ref_pred), rel_l2(emu_pred, ref_pred), rel_l2(pred, emu_pred)
        l_hip, l_emu = abs(loss - ref_loss) / abs(ref_loss), abs(emu_loss - ref_loss) / abs(ref_loss)
        rows = []
        for pname, gv in grads.items():
            if ".double_conv.0.bias" in pname or ".double_conv.3.bias" in pname:       # exact zeros here, rounding noise in the reference
                continue
            rows.append((pname, rel_l2(gv, ref_g[pname]), rel_l2(emu_g[pname], ref_g[pname]), rel_l2(gv, emu_g[pname])))
        hip, emu, pair = sorted(r[1] for r in rows), sorted(r[2] for r in rows), sorted(r[3] for r in rows)
        med = lambda v: v[len(v) // 2]
        # "worst tensor" is taken over the tensors this precision RESOLVES: where even the faithful CPU evaluation is >= 90 % away
        # from the fp32 gradient (a trained quantile head's bias: the mean of (q - 1[y < pred]) over the pixels, ~0 once the head
        # is calibrated) the gradient is rounding noise in any implementation and its relative error says nothing
        resolved = [r for r in rows if r[2] < 0.9] or rows
        worst = max(resolved, key=lambda r: r[1])
        hip_w, emu_w = worst[1], max(r[2] for r in resolved)
        print(f"[{name}/{stage}] outputs vs fp32: hip {e_hip:.4f} emu {e_emu:.4f} | hip vs emu {e_pair:.4f} | loss rel: hip {l_hip:.2e} emu {l_emu:.2e}")
        print(f"[{name}/{stage}] gradient rel-L2 vs fp32: median hip {med(hip):.4f} emu {med(emu):.4f} | worst resolved tensor hip {hip_w:.4f} "
              f"({worst[0]}) emu {emu_w:.4f} | {len(rows) - len(resolved)} of {len(rows)} tensors unresolved | hip vs emu median {med(pair):.4f}")
        # as close to fp32 as a faithful evaluation at this precision is
        assert e_hip <= 1.3 * e_emu + 2e-3, (stage, e_hip, e_emu)
        assert l_hip <= 1.5 * l_emu + 2e-3, (stage, l_hip, l_emu)
        assert med(hip) <= 1.3 * med(emu) + 2e-3, (stage, med(hip), med(emu))
        assert hip_w <= 1.5 * emu_w + 5e-3, (stage, worst, emu_w)
        # absolute: what the mode costs on a network someone would train
        c_out, c_loss, c_med, c_worst = ceilings[stage]
        assert e_hip < c_out and l_hip < c_loss and med(hip) < c_med and hip_w < c_worst, (stage, e_hip, l_hip, med(hip), hip_w)
