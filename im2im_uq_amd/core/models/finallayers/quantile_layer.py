"""Quantile-regression final layer -- drop-in for the reference's
core/models/finallayers/quantile_layer.py (layer :8-21, loss :23-32, nested sets :34-44)."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import hip_ops


def _lam_value(model, lam):
    if lam == None:
        if model.lhat == None:
            raise Exception("You have to specify lambda unless your model is already calibrated.")
        lam = model.lhat
    return lam


def quantile_regression_nested_sets_from_output(model, output, lam=None, _floor=False):
    """lower_edge, prediction, upper_edge = f(output [b,3,C,H,W], lam).  Like the reference this clamps
    output[:,0] / output[:,2] in place (lower <= pred-1e-6, upper >= pred+1e-6).

    On the GPU one HIP kernel does clamp + lambda scaling (+ the ModelWithUncertainty floor when
    `_floor`); the floor is idempotent, so applying it here as well as in a caller gives the same values."""
    lam = _lam_value(model, lam)
    if output.is_cuda:
        lam_f = float(torch.as_tensor(lam, dtype=torch.float32))
        if not output.is_contiguous():
            raise ValueError("nested sets: model output must be contiguous")
        lower, pred, upper = hip_ops.nested_sets(output, lam_f, clamp_inplace=True)
        if _floor:
            return lower, pred, upper
        # without the floor the reference returns the raw scaled edges; recover them only where the
        # floor was active (lam*(u-p) < 1e-6), i.e. re-scale those pixels -- rare, plumbing only.
        raw_up = lam * (output[:, 2] - output[:, 1]) + output[:, 1]
        raw_lo = output[:, 1] - lam * (output[:, 1] - output[:, 0])
        return raw_lo, pred, raw_up
    output[:, 0, :, :, :] = torch.minimum(output[:, 0, :, :, :], output[:, 1, :, :, :] - 1e-6)
    output[:, 2, :, :, :] = torch.maximum(output[:, 2, :, :, :], output[:, 1, :, :, :] + 1e-6)
    upper_edge = lam * (output[:, 2, :, :, :] - output[:, 1, :, :, :]) + output[:, 1, :, :, :]
    lower_edge = output[:, 1, :, :, :] - lam * (output[:, 1, :, :, :] - output[:, 0, :, :, :])
    return lower_edge, output[:, 1, :, :, :], upper_edge


class QuantileRegressionLayer(nn.Module):
    def __init__(self, n_channels_middle, n_channels_out, params):
        super().__init__()
        raise NotImplementedError("filled in with the conv kernels")


def quantile_regression_loss_fn(pred, target, params):
    raise NotImplementedError
