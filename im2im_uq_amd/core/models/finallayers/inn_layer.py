"""Interval (INN) final layer -- drop-in for the reference's core/models/finallayers/inn_layer.py
(layer :8-21, loss :22-28, nested sets :30-40) and core/models/losses/inn.py."""
from .... import hip_ops, nn_ops
from ._common import packed_loss
from .quantile_layer import QuantileRegressionLayer, quantile_regression_nested_sets_from_output


class INNLayer(QuantileRegressionLayer):
    """lower / prediction / upper heads, same shapes and names as the quantile layer (reference :14-21)."""

    def __init__(self, n_channels_middle, n_channels_out, params):
        super().__init__(n_channels_middle, n_channels_out, dict(params, q_lo=params.get("q_lo"), q_hi=params.get("q_hi")))
        self.beta = params["beta"]
        self.params = params


def inn_loss_fn(pred, target, params):
    """MSE(pred[:,1], y) + mean(relu(y - upper)^2 + relu(lower - y)^2 + beta*|upper - lower|)  (reference :22-28 with
    losses/inn.py:12-21), one fused reduction forward, one elementwise kernel backward."""
    beta = params["beta"]
    assert 0 <= beta
    return packed_loss(pred, target, 3, nn_ops.LOSS_INN, q_lo=beta, who="inn_loss_fn")


def inn_nested_sets_from_output(model, output, lam=None, _floor=False):
    """identical to the quantile layer's nested sets (reference :30-40)."""
    return quantile_regression_nested_sets_from_output(model, output, lam, _floor)


inn_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_QUANTILE
