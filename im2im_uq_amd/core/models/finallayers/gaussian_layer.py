"""Gaussian final layer -- drop-in for the reference's core/models/finallayers/gaussian_layer.py
(layer :7-17, loss :19-23, nested sets :25-34)."""

from .... import hip_ops, nn_ops
from ._common import TwoHeadLayer, fused_nested_sets, lam_value, packed_loss


class GaussianRegressionLayer(TwoHeadLayer):
    """mean and ReLU(variance) heads -> [B,2,C,H,W] (reference :12-17)."""
    _names = ("mean", "variance")
    _act = "relu"


def gaussian_regression_loss_fn(pred, target, params):
    """nn.GaussianNLLLoss()(pred[:,0], target, pred[:,1])  (reference :19-23): mean(0.5*(log v + (mean-y)^2/v)),
    v = max(var, 1e-6), one fused reduction kernel forward, one elementwise kernel backward."""
    # torch.nn.functional.gaussian_nll_loss rejects negative variances.  This package's own layer output is ReLU'd (tagged
    # by TwoHeadLayer.forward), so the check -- a host sync -- only runs for tensors that come from somewhere else.
    if pred.is_cuda and not getattr(pred, "_im2im_var_nonneg", False) and bool((pred[:, 1] < 0).any()):
        raise ValueError("var has negative entry/entries")
    return packed_loss(pred, target, 2, nn_ops.LOSS_GAUSSIAN, who="gaussian_regression_loss_fn")


def gaussian_regression_nested_sets_from_output(model, output, lam=None, _floor=False):
    """mean -+ lam*sqrt(variance)  (reference :25-34)."""
    lam = lam_value(model, lam)
    if not output.is_cuda:
        raise RuntimeError("nested sets: the output must be on the GPU; the HIP path has no CPU fallback")
    return fused_nested_sets(output, lam, hip_ops.SETS_SQRT, _floor)


gaussian_regression_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_SQRT
