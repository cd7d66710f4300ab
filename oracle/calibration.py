"""CPU restatement of the reference RCPS calibration path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Arithmetic substrate is the reference's own: PyTorch-CPU fp32 elementwise ops
(separate multiply and add, no FMA contraction) for the interval edges, and
scipy 1.15.3 float64 (``binom.cdf``, ``brentq``) for the Hoeffding-Bentkus
bound (the reference's environment.yml:14 pins scipy=1.4; the build image has
1.15.3 and the golden values in tests/golden/hb_bound.npz were produced by the
reference's bounds.py running on 1.15.3).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.optimize import brentq
from scipy.stats import binom


# ---------------------------------------------------------------- bounds.py
def kl_bernoulli(y, mu):
    """h1, core/calibration/bounds.py:6-7."""
    return y * np.log(y / mu) + (1 - y) * np.log((1 - y) / (1 - mu))


def hb_tail(mu, muhat, n, delta):
    """min(Hoeffding, Bentkus) log-tail minus log(delta), bounds.py:10-14,18-21."""
    hoeff = -n * kl_bernoulli(np.minimum(mu, muhat), mu)
    bent = np.log(max(binom.cdf(np.floor(n * muhat), n, mu), 1e-10)) + 1
    return min(hoeff, bent) - np.log(delta)


def hb_mu_plus(muhat, n, delta, maxiters=1000):
    """HB_mu_plus, bounds.py:17-29, including the exception path that returns 1.0
    (muhat == 0 makes h1 nan -> brentq raises)."""
    if hb_tail(1 - 1e-10, muhat, n, delta) > 0:
        return 1
    try:
        return brentq(lambda mu: hb_tail(mu, muhat, n, delta), muhat, 1 - 1e-10, maxiter=maxiters)
    except Exception:
        return 1.0


def evaluate_from_loss_table(loss_table, n, alpha, delta):
    """core/calibration/calibrate_model.py:62-74: shuffle the rows (torch RNG), calibrate on the first n, return the mean
    validation loss at the first lambda whose bound is <= delta (sic).  The reference hands HB_mu_plus 0-dim fp32 tensors
    and builds `torch.tensor([...])` (float32) from the results; both kept."""
    perm = torch.randperm(loss_table.shape[0])
    loss_table = loss_table[perm]
    calib_table, val_table = loss_table[:n], loss_table[n:]
    rhats = calib_table.mean(dim=0)
    rhat_plus = torch.tensor([hb_mu_plus(np.float32(r.item()), n, delta) for r in rhats])
    hits = (rhat_plus <= delta).nonzero()
    idx = hits[0] if hits.numel() else 0
    return val_table[:, idx].mean()


# ------------------------------------------------------- nested sets / loss
def lambda_grid(cfg, utype="quantiles"):
    """calibrate_model.py:97-100 / eval.py:92-95: fp32 linspace (softmax has its own lambda range keys)."""
    if utype == "softmax":
        return torch.linspace(cfg["minimum_lambda_softmax"], cfg["maximum_lambda_softmax"], cfg["num_lambdas"])
    return torch.linspace(cfg["minimum_lambda"], cfg["maximum_lambda"], cfg["num_lambdas"])


def softmax_summary(output):
    """lambda-independent part of softmax_nested_sets_from_output (softmax_layer.py:33-47): [b,K,C,H,W] class logits ->
    (lower quantile, prediction, upper quantile), each [b,C,H,W]."""
    prob = output.softmax(dim=1)
    k = prob.shape[1]
    cum = torch.cumsum(prob, dim=1)
    lq = (cum <= 0.05).float().sum(dim=1) / k
    uq = (cum <= 0.95).float().sum(dim=1) / k
    pred = torch.argmax(prob, dim=1) / k
    lq = torch.where(pred == lq, lq - 1 / k, lq)
    uq = torch.where(pred == uq, uq + 1 / k, uq)
    return lq.clamp(min=0, max=1), pred, uq.clamp(min=0, max=1)


def raw_nested_sets(output, lam, utype="quantiles"):
    """the final layer's own *_nested_sets_from_output (no floor):
    quantiles / quantiles_l1: quantile_layer.py:39-44 == quantile_l1_layer.py:39-44 (clamp, then pred -+ lam*width);
    gaussian: gaussian_layer.py:31-32 (mean -+ lam*sqrt(var));  residual_magnitude(_l1):
    residual_magnitude_layer.py:33-34 (pred -+ lam*magnitude).  ``output`` is NOT mutated here (the reference's
    quantile layers clamp in place; the clamp is idempotent)."""
    if utype == "softmax":                                             # softmax_layer.py:50-51
        lq, pred, uq = softmax_summary(output)
        return pred - (pred - lq).relu() * lam, pred, pred + (uq - pred).relu() * lam
    if utype in ("quantiles", "quantiles_l1", "inn"):                  # inn_layer.py:35-38 is the same expression
        lo, mid, hi = output[:, 0], output[:, 1], output[:, 2]
        lo = torch.minimum(lo, mid - 1e-6)
        hi = torch.maximum(hi, mid + 1e-6)
        return mid - lam * (mid - lo), mid, lam * (hi - mid) + mid
    mid = output[:, 0]
    scale = output[:, 1].sqrt() if utype == "gaussian" else output[:, 1]
    if utype not in ("gaussian", "residual_magnitude", "residual_magnitude_l1"):
        raise NotImplementedError(utype)
    return -lam * scale + mid, mid, lam * scale + mid


def nested_sets(output, lam, utype="quantiles"):
    """raw_nested_sets followed by the floor in ModelWithUncertainty.nested_sets_from_output
    (add_uncertainty.py:33-38).  ``lam`` is a 0-dim fp32 tensor or a python float, as at the reference's
    call sites."""
    lower, mid, upper = raw_nested_sets(output, lam, utype)
    upper = torch.maximum(upper, mid + 1e-6)
    lower = torch.minimum(lower, mid - 1e-6)
    return lower, mid, upper


def fraction_missed(lower, upper, label):
    """fraction_missed_loss, calibrate_model.py:76-80, per-image mean of the miss indicator.
    Written with an explicit flatten instead of the reference's squeeze so a batch of one
    image returns shape [1] (the reference returns [H] there)."""
    miss = (lower > label).float() + (upper < label).float()
    miss = miss.clamp(max=1.0)
    return miss.flatten(start_dim=1).mean(dim=1)


def losses_at(outputs, labels, lam, batch=64, utype="quantiles"):
    """get_rcps_losses_from_outputs, calibrate_model.py:21-29 (batches of 64, concatenated)."""
    parts = []
    for s in range(0, outputs.shape[0], batch):
        lo, _, hi = nested_sets(outputs[s:s + batch], lam, utype)
        parts.append(fraction_missed(lo, hi, labels[s:s + batch]))
    return torch.cat(parts, dim=0)


def calibrate_from_outputs(outputs, labels, cfg, utype="quantiles"):
    """Phase B of calibrate_model, calibrate_model.py:129-145: descending grid scan with the
    ``lam - dlambda`` shift (Q1), zero columns left of the break (Q2), stop rule
    ``Rhat >= alpha or RhatPlus > alpha`` (Q4), default lhat = last + dlambda - 1e-9.
    Returns (lhat 0-dim fp32 tensor, table [N,L] fp32, trace list of (j, Rhat, RhatPlus))."""
    alpha, delta = cfg["alpha"], cfg["delta"]
    lambdas = lambda_grid(cfg, utype)
    dlambda = lambdas[1] - lambdas[0]
    lhat = lambdas[-1] + dlambda - 1e-9
    n = outputs.shape[0]
    table = torch.zeros((n, lambdas.shape[0]))
    trace = []
    for j in range(lambdas.shape[0] - 1, -1, -1):
        lam = lambdas[j]
        losses = losses_at(outputs, labels, lam - dlambda, utype=utype)
        table[:, j] = losses
        rhat = losses.mean()
        rhat_plus = hb_mu_plus(rhat.item(), n, delta)
        trace.append((j, float(rhat.item()), float(rhat_plus)))
        if rhat >= alpha or rhat_plus > alpha:
            lhat = lam
            break
    return lhat, table, trace


def loss_table_from_outputs(outputs, labels, cfg):
    """get_loss_table's table phase, core/scripts/eval.py:116-125: un-shifted lambdas,
    batches of 4, every column filled."""
    lambdas = lambda_grid(cfg)
    n = outputs.shape[0]
    table = torch.zeros((n, cfg["num_lambdas"]))
    for s in range(0, n, 4):
        for j in range(lambdas.shape[0]):
            lo, _, hi = nested_sets(outputs[s:s + 4], lambdas[j])
            table[s:s + 4, j] = fraction_missed(lo, hi, labels[s:s + 4])
    return table


def risk_and_miscoverage(outputs, labels, lhat, utype="quantiles"):
    """RNG-free part of get_rcps_metrics_from_outputs, calibrate_model.py:31-60: per-image
    risk at lhat (:42) and the spatial miscoverage map = mean over images and channel of
    (label > upper) + (label < lower)  (:47,55)."""
    lo, _, hi = nested_sets(outputs, lhat, utype)
    losses = fraction_missed(lo, hi, labels)
    mis = (labels > hi).float() + (labels < lo).float()          # [N,C,H,W]
    spatial = mis.numpy().mean(axis=0).mean(axis=0)              # [H,W]
    return losses, spatial


def synth_logits(n, k, h, w, seed=0, sharp=0.5):
    """class logits [n,k,1,h,w] peaked around a smooth ground truth in [0,1] plus noise, and labels near that truth."""
    g = torch.Generator().manual_seed(seed)
    truth = torch.rand((n, 1, 1, h, w), generator=g)
    centres = torch.linspace(0, 1, k).view(1, k, 1, 1, 1)
    logits = -sharp * k * (centres - truth).abs() + 0.7 * torch.randn((n, k, 1, h, w), generator=g)
    y = (truth[:, 0] + 0.06 * torch.randn((n, 1, h, w), generator=g)).clamp(0, 1)
    return logits.contiguous(), y.contiguous()


def synth_outputs_two_plane(n, c, h, w, seed=0, width=0.05, utype="gaussian"):
    """calibration inputs for the two-plane layers: (pred, variance) for gaussian, (pred, magnitude) otherwise; a few
    exact zeros in the second plane (ReLU / abs outputs do hit 0) and labels equal to the prediction."""
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand((n, c, h, w), generator=g)
    mag = width * torch.rand((n, c, h, w), generator=g)
    y = pred + width * torch.randn((n, c, h, w), generator=g)
    mag.view(-1)[::97] = 0.0
    y.view(-1)[::89] = pred.view(-1)[::89]
    second = mag * mag if utype == "gaussian" else mag
    return torch.stack([pred, second], dim=1).contiguous(), y.contiguous()


def synth_outputs(n, c, h, w, seed=0, width=0.05):
    """SURVEY.md 8(d) kernel-only calibration inputs: pred~U[0,1], lower=pred-w*U,
    upper=pred+w*U, y=pred+w*N(0,1) so lhat lands mid-grid."""
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand((n, c, h, w), generator=g)
    lower = pred - width * torch.rand((n, c, h, w), generator=g)
    upper = pred + width * torch.rand((n, c, h, w), generator=g)
    y = pred + width * torch.randn((n, c, h, w), generator=g)
    return torch.stack([lower, pred, upper], dim=1).contiguous(), y.contiguous()
