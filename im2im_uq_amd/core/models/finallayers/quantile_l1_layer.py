"""Quantile-regression final layer with an L1 point loss -- drop-in for the reference's
core/models/finallayers/quantile_l1_layer.py (layer :8-21, loss :23-32, nested sets :34-44)."""
from .... import hip_ops, nn_ops
from ._common import packed_loss
from .quantile_layer import QuantileRegressionLayer, quantile_regression_nested_sets_from_output


class QuantileRegressionL1Layer(QuantileRegressionLayer):
    """same three heads as QuantileRegressionLayer (reference :8-21)."""


def quantile_regression_l1_loss_fn(pred, target, params):
    """w_lo*pinball_{q_lo}(pred[:,0]) + w_hi*pinball_{q_hi}(pred[:,2]) + mse_weight*L1(pred[:,1])  (reference :23-32)."""
    return packed_loss(pred, target, 3, nn_ops.LOSS_QUANTILE_L1, params["q_lo"], params["q_hi"], params["q_lo_weight"],
                       params["q_hi_weight"], params["mse_weight"], who="quantile_regression_l1_loss_fn")


def quantile_regression_l1_nested_sets_from_output(model, output, lam=None, _floor=False):
    """identical to the quantile layer's nested sets (reference :34-44)."""
    return quantile_regression_nested_sets_from_output(model, output, lam, _floor)


quantile_regression_l1_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_QUANTILE
