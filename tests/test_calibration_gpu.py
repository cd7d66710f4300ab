"""GPU parity of the calibration path: HIP kernels + host scan vs the CPU oracle and the golden
fixtures produced by the reference.  Integer/indicator work -> bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.utils.data import TensorDataset

from conftest import load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"
BASE = dict(uncertainty_type="quantiles", rcps_loss="fraction_missed", device=DEV, dataset="synthetic",
            q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def _identity_model():
    from im2im_uq_amd.core.models.add_uncertainty import ModelWithUncertainty
    from im2im_uq_amd.core.models.finallayers.quantile_layer import quantile_regression_nested_sets_from_output
    return ModelWithUncertainty(nn.Identity(), nn.Identity(), None, quantile_regression_nested_sets_from_output, dict(BASE))


def _cfg(v):
    return dict(BASE, alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]), minimum_lambda=float(v[3]),
                maximum_lambda=float(v[4]), batch_size=int(v[5]))


@pytest.mark.parametrize("case", ["mid", "n130", "zero_risk", "no_stop", "c2"])
def test_calibrate_model_matches_reference_golden(case):
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    g = load_golden("g7_calibrate_" + case)
    model = _identity_model()
    model, table = calibrate_model(model, TensorDataset(T(g["output"]).clone(), T(g["label"]).clone()), _cfg(g["cfg"]))
    assert np.array_equal(table.numpy(), g["table"])           # bit-exact incl. zero columns (Q2)
    assert float(model.lhat) == float(g["lhat"])


def test_loss_table_kernel_vs_oracle_random_grid():
    """full un-shifted table, odd sizes (P % 4 != 0 -> scalar path, N not multiple of anything)."""
    from im2im_uq_amd import hip_ops
    from oracle import calibration as oc
    for (n, c, h, w, L, lmax, seed) in [(5, 1, 7, 9, 17, 3.0, 0), (33, 2, 12, 20, 64, 8.0, 1), (3, 1, 64, 64, 1000, 6.0, 2),
                                         (2, 1, 33, 31, 1, 1.0, 3)]:
        out, y = oc.synth_outputs(n, c, h, w, seed=seed)
        # degenerate pixels: y == pred, tiny widths, inverted bounds
        out[0, 1, 0, 0, :3] = y[0, 0, 0, :3]
        out[0, 2, 0, 1, :3] = out[0, 1, 0, 1, :3] + 1e-7
        out[0, 0, 0, 2, :3] = out[0, 1, 0, 2, :3] + 0.3
        cfg = dict(num_lambdas=L, minimum_lambda=0.0 if L > 1 else 1.0, maximum_lambda=lmax)
        if L > 1:
            lambdas = oc.lambda_grid(cfg)
        else:
            lambdas = torch.tensor([lmax])
        table = hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), lambdas).cpu()
        ref = torch.stack([oc.fraction_missed(*oc.nested_sets(out, lam)[::2], y) for lam in lambdas], dim=1)
        assert np.array_equal(table.numpy(), ref.numpy()), (n, c, h, w, L)
        dl = lambdas[1] - lambdas[0] if L > 1 else torch.tensor(0.25)
        table = hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), lambdas - dl).cpu()     # shifted grid incl. negative lambda
        ref = torch.stack([oc.fraction_missed(*oc.nested_sets(out, lam - dl)[::2], y) for lam in lambdas], dim=1)
        assert np.array_equal(table.numpy(), ref.numpy()), (n, c, h, w, L, "shifted")


def test_loss_table_golden_g9():
    from im2im_uq_amd import hip_ops
    g = load_golden("g9_loss_table")
    lambdas = torch.linspace(float(g["cfg"][1]), float(g["cfg"][2]), int(g["cfg"][0]))
    table = hip_ops.rcps_loss_table(T(g["output"]).to(DEV), T(g["label"]).to(DEV), lambdas).cpu()
    assert np.array_equal(table.numpy(), g["table"])


def test_nested_sets_golden_g6():
    g = load_golden("g6_nested_sets")
    model = _identity_model()
    for i, lam in enumerate(T(g["lams"])):
        out = T(g["output"]).to(DEV).clone()
        lo, mid, hi = model.nested_sets_from_output(out, lam)
        assert np.array_equal(lo.cpu().numpy(), g["lower"][i])
        assert np.array_equal(hi.cpu().numpy(), g["upper"][i])
        # in-place clamp like the reference (Q5)
        ref = T(g["output"]).clone()
        assert np.array_equal(out[:, 0].cpu().numpy(), torch.minimum(ref[:, 0], ref[:, 1] - 1e-6).numpy())
        assert np.array_equal(out[:, 2].cpu().numpy(), torch.maximum(ref[:, 2], ref[:, 1] + 1e-6).numpy())


def test_metrics_golden_g10():
    from im2im_uq_amd.core.calibration.calibrate_model import get_rcps_metrics_from_outputs, fraction_missed_loss
    import random
    g = load_golden("g10_metrics")
    model = _identity_model()
    model.set_lhat(T(g["lhat"]))
    np.random.seed(0); torch.manual_seed(0); random.seed(0)     # fix_randomness(0), core/utils.py:15-19
    losses, sizes, spearman, strat, mse, spatial = get_rcps_metrics_from_outputs(
        model, TensorDataset(T(g["output"]), T(g["label"])), fraction_missed_loss, DEV)
    assert np.array_equal(losses.numpy(), g["losses"])
    assert np.array_equal(spatial, g["spatial"])
    # RNG-dependent parts: same RNG call order as the reference -> same sampled pixels
    np.testing.assert_allclose(sizes.numpy(), g["rng_sizes"], rtol=0, atol=1e-7)
    assert spearman == pytest.approx(float(g["rng_spearman"]), abs=1e-6)
    assert mse == pytest.approx(float(g["rng_mse"]), rel=1e-6)
    np.testing.assert_allclose(strat.numpy(), g["rng_strat"], rtol=1e-6)


def test_fraction_missed_and_large_image_properties():
    """size-independent properties at BASELINE size 320x320: monotone in lambda, table[:, j] equals the
    single-lambda call, counts are integers in [0, P], sum over shards == whole."""
    from im2im_uq_amd import hip_ops
    from oracle import calibration as oc
    out, y = oc.synth_outputs(6, 1, 320, 320, seed=11)
    lambdas = torch.linspace(0, 6, 1000)
    o, l = out.to(DEV), y.to(DEV)
    table, counts = hip_ops.rcps_loss_table(o, l, lambdas, want_counts=True)
    assert (table[:, 1:] <= table[:, :-1]).all()
    assert int(counts.min()) >= 0 and int(counts.max()) <= 320 * 320
    for j in (0, 1, 499, 999):
        lo, mid, hi = hip_ops.nested_sets(o.clone(), float(lambdas[j]))
        single = hip_ops.fraction_missed(lo, hi, l)
        assert torch.equal(single, table[:, j])
    # oracle on one full-size image at three lambdas
    for j in (0, 250, 999):
        ref = oc.fraction_missed(*oc.nested_sets(out[:1], lambdas[j])[::2], y[:1])
        assert torch.equal(ref, table[:1, j].cpu())
    mis = hip_ops.rcps_miscoverage(o, l, float(lambdas[300]))
    assert int(mis.sum()) == int(counts[:, 300].sum())


UTYPES = ["quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "inn"]


def _family(utype):
    import importlib
    from im2im_uq_amd.core.models.add_uncertainty import ModelWithUncertainty
    mod, prefix = {"quantiles_l1": ("quantile_l1_layer", "quantile_regression_l1"), "gaussian": ("gaussian_layer", "gaussian_regression"),
                   "residual_magnitude": ("residual_magnitude_layer", "residual_magnitude"),
                   "residual_magnitude_l1": ("residual_magnitude_l1_layer", "residual_magnitude_l1"), "inn": ("inn_layer", "inn")}[utype]
    m = importlib.import_module("im2im_uq_amd.core.models.finallayers." + mod)
    return ModelWithUncertainty(nn.Identity(), nn.Identity(), getattr(m, prefix + "_loss_fn"),
                                getattr(m, prefix + "_nested_sets_from_output"), dict(BASE, uncertainty_type=utype))


@pytest.mark.parametrize("utype", UTYPES)
def test_g12_other_final_layers_nested_sets_and_calibration_bit_exact(utype):
    """nested sets (with ModelWithUncertainty's floor and the layer's own raw edges), calibrate_model (one-pass table for
    every lambda, lhat) and the metrics at lhat for the gaussian / residual-magnitude / quantile-L1 layers: identical
    bits to the reference (fixtures from tests/golden/make_golden.py g12)."""
    import random
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model, get_rcps_metrics_from_outputs, fraction_missed_loss
    g = load_golden("g12_" + utype)
    model = _family(utype)
    for i, lam in enumerate(T(g["lams"])):
        lo, _, hi = model.nested_sets_from_output(T(g["sets_output"]).to(DEV).clone(), lam)
        assert np.array_equal(lo.cpu().numpy(), g["lower"][i]) and np.array_equal(hi.cpu().numpy(), g["upper"][i]), float(lam)
        rlo, _, rhi = model.in_nested_sets_from_output_fn(model, T(g["sets_output"]).to(DEV).clone(), lam)
        assert np.array_equal(rlo.cpu().numpy(), g["raw_lower"][i]) and np.array_equal(rhi.cpu().numpy(), g["raw_upper"][i]), float(lam)
    ds = TensorDataset(T(g["cal_output"]).clone(), T(g["cal_label"]).clone())
    model, table = calibrate_model(model, ds, _cfg(g["cfg"]))
    assert np.array_equal(table.numpy(), g["table"])
    assert float(model.lhat) == float(g["lhat"])
    np.random.seed(0); torch.manual_seed(0); random.seed(0)
    losses, sizes, spearman, strat, mse, spatial = get_rcps_metrics_from_outputs(model, ds, fraction_missed_loss, DEV)
    assert np.array_equal(losses.numpy(), g["risk"]) and np.array_equal(spatial, g["spatial"])
    np.testing.assert_allclose(sizes.numpy(), g["sizes"], rtol=0, atol=1e-7)
    assert spearman == pytest.approx(float(g["spearman"]), abs=1e-6) and mse == pytest.approx(float(g["mse"]), rel=1e-6)


@pytest.mark.parametrize("utype", ["gaussian", "residual_magnitude"])
def test_two_plane_loss_table_vs_oracle_incl_negative_lambda_and_zero_width(utype):
    from im2im_uq_amd import hip_ops
    from oracle import calibration as oc
    form = hip_ops.SETS_SQRT if utype == "gaussian" else hip_ops.SETS_SCALE
    for (n, c, h, w, L, lmax, seed) in [(5, 1, 7, 9, 17, 3.0, 0), (33, 2, 12, 20, 64, 8.0, 1), (3, 1, 64, 64, 1000, 6.0, 2)]:
        out, y = oc.synth_outputs_two_plane(n, c, h, w, seed=seed, utype=utype)
        lambdas = oc.lambda_grid(dict(num_lambdas=L, minimum_lambda=0.0, maximum_lambda=lmax))
        for grid in (lambdas, lambdas - (lambdas[1] - lambdas[0])):
            table = hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), grid, form=form).cpu()
            ref = torch.stack([oc.fraction_missed(*oc.nested_sets(out, lam, utype)[::2], y) for lam in grid], dim=1)
            assert np.array_equal(table.numpy(), ref.numpy()), (n, c, h, w, L)


def test_g13_softmax_nested_sets_and_calibration():
    """softmax nested sets (softmax -> running sum -> 5 %/95 % bins -> argmax prediction) and calibration vs the reference.
    The bin counts are threshold decisions on fp32 running sums of exponentials, which differ in the last bit between any
    two exp implementations (the reference's own CPU and GPU paths included), so a pixel may land one bin away: at most
    0.5 % of the pixels may differ, every other value is identical, the loss table moves by at most 2 pixels per image
    and lhat by at most one grid step."""
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import ModelWithUncertainty
    from im2im_uq_amd.core.models.finallayers.softmax_layer import softmax_loss_fn, softmax_nested_sets_from_output
    g = load_golden("g13_softmax")
    v = g["cfg"]
    cfg = dict(BASE, uncertainty_type="softmax", num_softmax=50, alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]),
               minimum_lambda_softmax=float(v[3]), maximum_lambda_softmax=float(v[4]), batch_size=int(v[5]))
    model = ModelWithUncertainty(nn.Identity(), nn.Identity(), softmax_loss_fn, softmax_nested_sets_from_output, cfg)
    worst = 0.0
    for i, lam in enumerate(T(g["lams"])):
        lo, mid, hi = model.nested_sets_from_output(T(g["sets_output"]).to(DEV), lam)
        rlo, _, rhi = softmax_nested_sets_from_output(model, T(g["sets_output"]).to(DEV), lam)
        for got, key in ((lo, "lower"), (hi, "upper"), (mid, "prediction"), (rlo, "raw_lower"), (rhi, "raw_upper")):
            worst = max(worst, float((got.cpu().numpy() != g[key][i]).mean()))
    assert worst <= 0.005, worst
    ds = TensorDataset(T(g["cal_output"].astype(np.float32)), T(g["cal_label"]).clone())
    model, table = calibrate_model(model, ds, cfg)
    p = g["cal_label"][0].size
    assert float(np.abs(table.numpy() - g["table"]).max()) <= 2.0 / p + 1e-7
    step = (float(v[4]) - float(v[3])) / (int(v[2]) - 1)
    assert abs(float(model.lhat) - float(g["lhat"])) <= step + 1e-6
    print("softmax: worst differing-pixel fraction", worst, "table max diff", float(np.abs(table.numpy() - g["table"]).max()),
          "lhat", float(model.lhat), float(g["lhat"]))


def test_lambda_grid_size_limits():
    """the largest lambda grid the scoring kernel takes (8192 points: histogram + grid fill the 64 KiB of LDS) is still
    bit-exact, and one more point is refused with a named error instead of a silent truncation."""
    from im2im_uq_amd import hip_ops, _lib
    from oracle import calibration as oc
    out, y = oc.synth_outputs(3, 1, 16, 16, seed=0)
    lam = torch.linspace(0, 6, 8192)
    table = hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), lam).cpu()
    cols = list(range(0, 8192, 97)) + [8191]
    ref = torch.stack([oc.fraction_missed(*oc.nested_sets(out, lam[j])[::2], y) for j in cols], dim=1)
    assert np.array_equal(table[:, cols].numpy(), ref.numpy())
    with pytest.raises(_lib.Im2ImError, match="MAX_L"):
        hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), torch.linspace(0, 6, 8193))


def test_empty_shard_is_a_no_op():
    """a rank whose calibration shard is empty (fewer images than ranks): the scoring entry points accept N = 0."""
    from im2im_uq_amd import hip_ops
    out = torch.empty((0, 3, 1, 16, 16), device=DEV)
    lab = torch.empty((0, 1, 16, 16), device=DEV)
    table = hip_ops.rcps_loss_table(out, lab, torch.linspace(0, 6, 50))
    assert tuple(table.shape) == (0, 50)
    counts = hip_ops.rcps_miscoverage(out, lab, 1.0)
    assert tuple(counts.shape) == (1, 256) and int(counts.abs().sum()) == 0


@pytest.mark.parametrize("utype,form", [("quantiles", 0), ("residual_magnitude", 1), ("gaussian", 2)])
def test_nan_and_inf_inputs_follow_torch_semantics(utype, form):
    """NaN / inf in the model output or the label: torch.maximum/minimum propagate NaN and every comparison with NaN is
    false, so such a pixel is never counted as missed; +-inf behave as ordinary floats.  Loss table, nested sets and
    miscoverage map must agree with the reference semantics bit for bit."""
    from im2im_uq_amd import hip_ops
    from oracle import calibration as oc
    if utype == "quantiles":
        out, y = oc.synth_outputs(4, 1, 16, 16, seed=11)
    else:
        out, y = oc.synth_outputs_two_plane(4, 1, 16, 16, seed=11, utype=utype)
    k = out.shape[1]
    bad = [float("nan"), float("inf"), -float("inf")]
    for i in range(60):                                        # scatter special values over every plane and the label
        v = bad[i % 3]
        if i % (k + 1) == k:
            y.view(-1)[i * 13 + 5] = v
        else:
            out[:, i % (k + 1)].reshape(-1)[i * 17 + 3] = v
    if utype == "gaussian":
        out[0, 1, 0, 0, :3] = -1.0                             # negative variance -> sqrt = NaN
    lambdas = oc.lambda_grid(dict(num_lambdas=40, minimum_lambda=0.0, maximum_lambda=6.0))
    grid = lambdas - (lambdas[1] - lambdas[0])
    table = hip_ops.rcps_loss_table(out.to(DEV), y.to(DEV), grid, form=form).cpu()
    ref = torch.stack([oc.fraction_missed(*oc.nested_sets(out, lam, utype)[::2], y) for lam in grid], dim=1)
    assert np.array_equal(table.numpy(), ref.numpy())
    lo, _, hi = hip_ops.nested_sets(out.to(DEV).clone(), 1.5, clamp_inplace=False, form=form)
    rlo, _, rhi = oc.nested_sets(out, torch.tensor(1.5), utype)
    assert np.array_equal(lo.cpu().numpy(), rlo.numpy(), equal_nan=True) and np.array_equal(hi.cpu().numpy(), rhi.numpy(), equal_nan=True)
    counts = hip_ops.rcps_miscoverage(out.to(DEV), y.to(DEV), 1.5, form=form).cpu().reshape(16, 16)
    _, spatial = oc.risk_and_miscoverage(out, y, torch.tensor(1.5), utype)
    assert np.array_equal(counts.numpy().astype(np.float32) / np.float32(4), spatial)
