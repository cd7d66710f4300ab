"""Quantile-regression final layer -- drop-in for the reference's
core/models/finallayers/quantile_layer.py (layer :8-21, loss :23-32, nested sets :34-44)."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import hip_ops, nn_ops


def _lam_value(model, lam):
    if lam == None:
        if model.lhat == None:
            raise Exception("You have to specify lambda unless your model is already calibrated.")
        lam = model.lhat
    return lam


def quantile_regression_nested_sets_from_output(model, output, lam=None, _floor=False):
    """lower_edge, prediction, upper_edge = f(output [b,3,C,H,W], lam).  Like the reference this clamps
    output[:,0] / output[:,2] in place (lower <= pred-1e-6, upper >= pred+1e-6).

    On the GPU one HIP kernel does clamp + lambda scaling (+ the ModelWithUncertainty floor when `_floor`)."""
    lam = _lam_value(model, lam)
    if output.is_cuda:
        lam_f = float(torch.as_tensor(lam, dtype=torch.float32))
        if not output.is_contiguous():
            raise ValueError("nested sets: model output must be contiguous")
        return hip_ops.nested_sets(output, lam_f, clamp_inplace=True, form=hip_ops.SETS_QUANTILE, floor=_floor)
    output[:, 0, :, :, :] = torch.minimum(output[:, 0, :, :, :], output[:, 1, :, :, :] - 1e-6)
    output[:, 2, :, :, :] = torch.maximum(output[:, 2, :, :, :], output[:, 1, :, :, :] + 1e-6)
    upper_edge = lam * (output[:, 2, :, :, :] - output[:, 1, :, :, :]) + output[:, 1, :, :, :]
    lower_edge = output[:, 1, :, :, :] - lam * (output[:, 1, :, :, :] - output[:, 0, :, :, :])
    return lower_edge, output[:, 1, :, :, :], upper_edge


quantile_regression_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_QUANTILE


class QuantileRegressionLayer(nn.Module):
    """lower / prediction / upper 3x3 heads -> [B,3,C,H,W] fp32 (reference :8-21); the three nn.Conv2d are
    parameter containers (reference names and initialisation), the forward is one fused HIP kernel."""

    def __init__(self, n_channels_middle, n_channels_out, params):
        super(QuantileRegressionLayer, self).__init__()
        self.q_lo = params["q_lo"]
        self.q_hi = params["q_hi"]
        self.params = params

        self.lower = nn.Conv2d(n_channels_middle, n_channels_out, kernel_size=3, padding=1)
        self.prediction = nn.Conv2d(n_channels_middle, n_channels_out, kernel_size=3, padding=1)
        self.upper = nn.Conv2d(n_channels_middle, n_channels_out, kernel_size=3, padding=1)
        self.compute_dtype = None

    def forward(self, x):
        cdt = getattr(self, "compute_dtype", None) or nn_ops.get_compute_dtype()
        if x.dtype in (torch.float32, torch.bfloat16) and x.permute(0, 2, 3, 1).is_contiguous():
            cdt = x.dtype                      # consume the trunk's channels-last feature map zero-copy
        return nn_ops.QuantileHeads.apply(x, self.lower.weight, self.lower.bias, self.prediction.weight,
                                          self.prediction.bias, self.upper.weight, self.upper.bias, cdt)


def quantile_regression_loss_fn(pred, target, params):
    """w_lo*pinball_{q_lo}(pred[:,0]) + w_hi*pinball_{q_hi}(pred[:,2]) + w_mse*MSE(pred[:,1])  (reference :23-32),
    one fused reduction kernel forward, one elementwise kernel backward."""
    if not pred.is_cuda:
        raise RuntimeError("quantile_regression_loss_fn: tensors must be on the GPU; the HIP path has no CPU fallback")
    if pred.dim() != 5 or pred.shape[1] != 3:
        raise ValueError(f"pred must be [B,3,C,H,W], got {tuple(pred.shape)}")
    p = pred if (pred.dtype == torch.float32 and pred.is_contiguous()) else pred.to(torch.float32).contiguous()
    t = target.detach().to(device=pred.device, dtype=torch.float32).contiguous()
    if t.numel() != p.shape[0] * p[0, 0].numel():
        raise ValueError("target shape does not match pred")
    return nn_ops.QuantileLossPacked.apply(p, t, float(params["q_lo"]), float(params["q_hi"]), float(params["q_lo_weight"]),
                                           float(params["q_hi_weight"]), float(params["mse_weight"]))
