"""bf16 (the benchmarked mode) against fp32 (the parity mode, itself pinned to the reference by G4/G5/G11/G17) over a
REAL training run, end to end: same initial weights, same batches in the same order, 240 Adam steps on a 64x64
denoising task, then calibration and validation -- the sequence of core/scripts/train.py:141-165 followed by
calibrate_model.py:89-145 and eval.py:130-157.

What is asserted (numbers measured on MI355X, see the printed line in the test log; each bound has ~2x head-room over
the measured value so that a real regression -- a wrong rounding point, a lost gradient -- trips it):
  * the bf16 loss curve tracks the fp32 one: mean train loss over the last 40 steps within 4 %;
  * both trained models calibrate (alpha = delta = 0.1, 100 lambdas): lambda-hat within 3 grid steps of each other;
  * both calibrated models hold the risk on 96 held-out images: validation risk <= alpha;
  * the trained models agree as functions: prediction images within 3 % relative L2, calibrated lower / upper edges
    within 4 %.
"""
import numpy as np
import pytest
import torch
from torch.utils.data import TensorDataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=100, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              device=DEV, dataset="synthetic", batch_size=16, lr=1e-3, input_normalization="standard",
              output_normalization="min-max", num_validation_images=2)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(autouse=True)
def _restore_dtype():
    from im2im_uq_amd import nn_ops
    yield
    nn_ops.set_compute_dtype("bf16")


def _run(dt, data, steps, hw):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.eval import eval_set_metrics
    from oracle import model as om
    nn_ops.set_compute_dtype(dt)
    model = add_uncertainty(UNet(1, 1), dict(PARAMS))
    model.load_state_dict(om.det_state(1, 1))
    model = model.to(DEV).train()
    opt = nn_ops.FusedAdam(model.parameters(), lr=PARAMS["lr"])
    (xt, yt), (xc, yc), (xv, yv) = data
    nb = xt.shape[0] // 16
    losses = []
    for step in range(steps):
        s = (step % nb) * 16
        loss = model.loss_fn(model(xt[s:s + 16]), yt[s:s + 16])
        losses.append(loss)
        opt.zero_grad(); loss.backward(); opt.step()
    losses = torch.stack(losses).cpu().numpy()
    cfg = dict(PARAMS)
    model, table = calibrate_model(model, TensorDataset(xc, yc), cfg)
    torch.manual_seed(0); np.random.seed(0)
    risk = eval_set_metrics(model, TensorDataset(xv, yv), cfg)[0]
    with torch.no_grad():
        lo, mid, hi = model.nested_sets((xv,))
    return dict(losses=losses, lhat=float(model.lhat), risk=float(risk), lo=lo.float().cpu(), mid=mid.float().cpu(),
                hi=hi.float().cpu())


def test_bf16_training_tracks_fp32_training_then_calibrates_alike():
    from im2im_uq_amd.core.datasets.synthetic import SyntheticDenoiseDataset
    hw, steps = 64, 240
    ds = SyntheticDenoiseDataset(num_images=96 + 96 + 96, num_inputs=1, side=hw, noise=0.1, seed=5)
    x, y = ds.x.to(DEV), ds.y.to(DEV)
    data = ((x[:96], y[:96]), (x[96:192], y[96:192]), (x[192:], y[192:]))
    r32 = _run("fp32", data, steps, hw)
    r16 = _run("bf16", data, steps, hw)
    tail32, tail16 = r32["losses"][-40:].mean(), r16["losses"][-40:].mean()
    dl = 6.0 / 99
    d_lhat = abs(r32["lhat"] - r16["lhat"]) / dl
    e_mid, e_lo, e_hi = rel_l2(r16["mid"], r32["mid"]), rel_l2(r16["lo"], r32["lo"]), rel_l2(r16["hi"], r32["hi"])
    print(f"\n[train parity] loss0 {r32['losses'][0]:.4f}/{r16['losses'][0]:.4f}  tail40 fp32 {tail32:.5f} bf16 {tail16:.5f} "
          f"(ratio {tail16 / tail32:.4f})  lhat fp32 {r32['lhat']:.4f} bf16 {r16['lhat']:.4f} ({d_lhat:.2f} grid steps)  "
          f"val risk fp32 {r32['risk']:.4f} bf16 {r16['risk']:.4f}  rel-L2 pred {e_mid:.4f} lower {e_lo:.4f} upper {e_hi:.4f}")
    assert np.isfinite(r32["losses"]).all() and np.isfinite(r16["losses"]).all()
    assert tail32 < 0.25 * r32["losses"][0] and tail16 < 0.25 * r16["losses"][0]      # both actually trained
    assert r16["losses"][0] == pytest.approx(r32["losses"][0], rel=1e-2)              # same start
    assert abs(tail16 / tail32 - 1.0) < 0.04
    assert d_lhat <= 3.0 + 1e-6
    assert 0 < r32["lhat"] < 6 and 0 < r16["lhat"] < 6                                # stopped inside the grid
    assert r32["risk"] <= PARAMS["alpha"] and r16["risk"] <= PARAMS["alpha"]
    assert e_mid < 0.03 and e_lo < 0.04 and e_hi < 0.04
