"""bf16 (the benchmarked mode) against fp32 (the parity mode, itself pinned to the reference by G4/G5/G11/G17) over a
REAL training run, end to end: same initial weights, same batches in the same order, 600 Adam steps (lr 3e-4) on a 64x64
denoising task, then calibration and validation -- the sequence of core/scripts/train.py:141-165 followed by
calibrate_model.py:89-145 and eval.py:130-157.

Training is chaotic: two fp32 runs that differ by a 1e-4 relative perturbation of the initial weights (40x smaller than
one bf16 rounding), or just by the summation order of one kernel, end up percents apart in every metric below; at lr 1e-3
the calibrated interval size of fp32 runs alone spreads over 0.13-0.23 (tools/train_parity_probe.py), which is why this
test trains at 3e-4, where the spread is small enough to compare against.  That fp32-vs-fp32' distance is the yardstick:
the bf16 run has to land as close to the fp32 run as a second fp32 run does (<= 2x the yardstick plus a margin), and
within absolute bounds.  Measured on MI355X over 3 perturbation seeds (fp32' vs fp32 | bf16 vs fp32):
  tail-200 train loss 0.3-2.4 % | 2.0-6.3 %;  lambda-hat 1-3 | 1-4 grid steps of 100;  prediction images (rel. L2)
  6.7-6.8 % | 6.0-8.0 %;  calibrated lower edge 7.2-9.1 % | 6.2-12.5 %;  upper edge 5.5-8.2 % | 4.8-9.1 %;  mean
  calibrated interval size 0.142-0.147 | 0.138-0.162 (fp32: 0.138);  validation risk 0.050-0.052 for all (alpha = 0.1).
Bounds asserted: loss 10 %, lambda-hat 2x the fp32-vs-fp32' distance + 3 steps, prediction 12 %, lower 20 %, upper 15 %, size ratio within 2x the fp32-vs-fp32' log-ratio + 20 % [r6: was an absolute [0.75, 1.35]], risk
<= alpha -- a wrong rounding point or a lost gradient term costs tens of percent and a broken calibration moves the risk.
"""
import numpy as np
import pytest
import torch
from torch.utils.data import TensorDataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=100, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              device=DEV, dataset="synthetic", batch_size=16, lr=3e-4, input_normalization="standard",
              output_normalization="min-max", num_validation_images=2)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(autouse=True)
def _restore_dtype():
    from im2im_uq_amd import nn_ops
    yield
    nn_ops.set_compute_dtype("bf16")


def _run(dt, data, steps, hw, perturb=0.0, seed=99):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.eval import eval_set_metrics
    from oracle import model as om
    nn_ops.set_compute_dtype(dt)
    model = add_uncertainty(UNet(1, 1), dict(PARAMS))
    st = om.det_state(1, 1)
    if perturb:
        g = torch.Generator().manual_seed(seed)
        st = {k: (v * (1 + perturb * torch.randn(v.shape, generator=g)) if om.is_param(k) else v) for k, v in st.items()}
    model.load_state_dict(st)
    model = model.to(DEV).train()
    opt = nn_ops.FusedAdam(model.parameters(), lr=PARAMS["lr"])
    (xt, yt), (xc, yc), (xv, yv) = data
    nb = xt.shape[0] // 16
    losses = []
    for step in range(steps):
        s = (step % nb) * 16
        loss = model.loss_fn(model(xt[s:s + 16]), yt[s:s + 16])
        losses.append(loss.detach())
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


def _distance(a, ref):
    dl = 6.0 / 99
    return dict(loss=abs(a["losses"][-200:].mean() / ref["losses"][-200:].mean() - 1.0), lhat=abs(a["lhat"] - ref["lhat"]) / dl,
                mid=rel_l2(a["mid"], ref["mid"]), lo=rel_l2(a["lo"], ref["lo"]), hi=rel_l2(a["hi"], ref["hi"]))


def test_bf16_training_tracks_fp32_training_then_calibrates_alike():
    from im2im_uq_amd.core.datasets.synthetic import SyntheticDenoiseDataset
    hw, steps = 64, 600
    ds = SyntheticDenoiseDataset(num_images=96 + 96 + 96, num_inputs=1, side=hw, noise=0.1, seed=5)
    x, y = ds.x.to(DEV), ds.y.to(DEV)
    data = ((x[:96], y[:96]), (x[96:192], y[96:192]), (x[192:], y[192:]))
    r32 = _run("fp32", data, steps, hw)
    r32b = _run("fp32", data, steps, hw, perturb=1e-4)         # the yardstick: fp32 against (almost) itself
    r16 = _run("bf16", data, steps, hw)
    d16, dself = _distance(r16, r32), _distance(r32b, r32)
    print(f"\n[train parity] loss0 {r32['losses'][0]:.4f}/{r16['losses'][0]:.4f}  tail200 fp32 {r32['losses'][-200:].mean():.5f} "
          f"bf16 {r16['losses'][-200:].mean():.5f} fp32' {r32b['losses'][-200:].mean():.5f}  lhat {r32['lhat']:.4f}/{r16['lhat']:.4f}/"
          f"{r32b['lhat']:.4f}  val risk {r32['risk']:.4f}/{r16['risk']:.4f}/{r32b['risk']:.4f}\n"
          f"  bf16 vs fp32 : " + "  ".join(f"{k} {v:.4f}" for k, v in d16.items()) + "\n"
          f"  fp32' vs fp32: " + "  ".join(f"{k} {v:.4f}" for k, v in dself.items()))
    size = {k: float((r["hi"] - r["lo"]).mean()) for k, r in (("fp32", r32), ("bf16", r16), ("fp32'", r32b))}
    print("  mean calibrated interval size: " + "  ".join(f"{k} {v:.4f}" for k, v in size.items()))
    for r in (r32, r16, r32b):
        assert np.isfinite(r["losses"]).all()
        assert r["losses"][-200:].mean() < 0.1 * r["losses"][0]                       # actually trained
        assert 0 < r["lhat"] < 6                                                      # the scan stopped inside the grid
        assert r["risk"] <= PARAMS["alpha"]                                           # the calibrated sets hold the risk
    assert r16["losses"][0] == pytest.approx(r32["losses"][0], rel=1e-2)              # same start
    # absolute bounds (measured spreads in the module docstring)
    # (lambda-hat itself is NOT bounded absolutely: it is the ratio of two chaotic quantities -- the raw head widths of two fp32
    # runs differed by 1.5x after a kernel's summation order changed [round 3: 17 grid steps between fp32 and fp32', 11
    # between bf16 and fp32] while lambda-hat x width, the calibrated size below, agreed to 15-20 %; it is held relative to
    # the fp32-vs-fp32' distance further down)
    assert d16["loss"] < 0.10
    assert d16["mid"] < 0.12 and d16["lo"] < 0.20 and d16["hi"] < 0.15
    # [r6] the calibrated interval size is held RELATIVE to the yardstick like everything else: two fp32 runs that differ by the 1e-4
    # perturbation now sit at 0.183 vs 0.137 (ratio 0.75, the old absolute bound itself), and bf16 lands at 0.128 / 0.149 / 0.158 with
    # three bit-different but equally accurate (7e-6) heads weight-gradient kernels (profiles/r06_ab_experiments.txt section 4)
    log_ratio = abs(np.log(size["bf16"] / size["fp32"]))
    assert log_ratio <= 2 * abs(np.log(size["fp32'"] / size["fp32"])) + np.log(1.2), size
    assert 0.5 < size["bf16"] / size["fp32"] < 2.0
    # and no further from fp32 than fp32 is from itself (2x + margin)
    assert d16["loss"] <= 2 * dself["loss"] + 0.05
    assert d16["lhat"] <= 2 * dself["lhat"] + 3.0
    for k in ("mid", "lo", "hi"):
        assert d16[k] <= 2 * dself[k] + 0.04, (k, d16[k], dself[k])
