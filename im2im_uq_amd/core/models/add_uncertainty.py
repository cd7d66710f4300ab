"""Plugin boundary -- drop-in for the reference's core/models/add_uncertainty.py.

`add_uncertainty(trunk, params)` wraps a trunk (needs `n_channels_middle`, `n_channels_out`) with a
final layer chosen by params["uncertainty_type"] and returns a ModelWithUncertainty exposing the same
API as the reference (:15-49): forward, loss_fn, nested_sets_from_output, nested_sets, set_lhat,
buffer `lhat`, attributes baseModel / last_layer / params.  state_dict keys are the reference's.

Scope of this build (SURVEY section 8 + 8f rank 1): "quantiles", "quantiles_l1", "gaussian", "residual_magnitude" and
"residual_magnitude_l1" run on the HIP kernels (same trunk, heads and calibration kernels; only the head count, the
fused loss and the nested-set formula differ), "softmax" too (MFMA conv to the class logits, fused cross entropy, a
softmax-quantile summary kernel feeding the same calibration kernels) and "inn" (quantile heads, interval loss): all
seven types of the reference's factory (:51-87).
"""
import torch
import torch.nn as nn

from .. import _pkg  # noqa: F401
from .finallayers.quantile_layer import (QuantileRegressionLayer, quantile_regression_loss_fn,
                                         quantile_regression_nested_sets_from_output)
from .finallayers.quantile_l1_layer import (QuantileRegressionL1Layer, quantile_regression_l1_loss_fn,
                                            quantile_regression_l1_nested_sets_from_output)
from .finallayers.gaussian_layer import (GaussianRegressionLayer, gaussian_regression_loss_fn,
                                         gaussian_regression_nested_sets_from_output)
from .finallayers.residual_magnitude_layer import (ResidualMagnitudeLayer, residual_magnitude_loss_fn,
                                                   residual_magnitude_nested_sets_from_output)
from .finallayers.residual_magnitude_l1_layer import (ResidualMagnitudeL1Layer, residual_magnitude_l1_loss_fn,
                                                      residual_magnitude_l1_nested_sets_from_output)
from .finallayers.softmax_layer import SoftmaxLayer, softmax_loss_fn, softmax_nested_sets_from_output
from .finallayers.inn_layer import INNLayer, inn_loss_fn, inn_nested_sets_from_output


def sets_form(model):
    """IM2IM_SETS_* form of the model's nested sets when they are one of this package's (then every calibration kernel
    evaluates them directly from the raw output planes), else None (plugin function: generic per-lambda path)."""
    return getattr(getattr(model, "in_nested_sets_from_output_fn", None), "im2im_sets_form", None)


def calibration_repr(model, output):
    """what calibration keeps of a model output: the output itself, or (softmax layer) its lambda-independent
    [b,3,C,H,W] quantile summary."""
    summarize = getattr(getattr(model, "in_nested_sets_from_output_fn", None), "im2im_summarize", None)
    return output if summarize is None else summarize(output)


class ModelWithUncertainty(nn.Module):
    def __init__(self, baseModel, last_layer, in_train_loss_fn, in_nested_sets_from_output_fn, params):
        super(ModelWithUncertainty, self).__init__()
        self.baseModel = baseModel
        self.last_layer = last_layer
        self.register_buffer('lhat', None)
        self.in_train_loss_fn = in_train_loss_fn
        self.in_nested_sets_from_output_fn = in_nested_sets_from_output_fn
        self.params = params

    def forward(self, x):
        x = self.baseModel(x)
        return self.last_layer(x)

    def loss_fn(self, pred, target):
        return self.in_train_loss_fn(pred, target, self.params)

    def nested_sets_from_output(self, output, lam=None):
        """(lower_edge, prediction, upper_edge) with the +-1e-6 floor (reference :33-38)."""
        if sets_form(self) is not None and output.is_cuda:
            # fused HIP path: (clamp,) scale and floor in one kernel (identical fp32 op order)
            return self.in_nested_sets_from_output_fn(self, output, lam, _floor=True)
        lower_edge, prediction, upper_edge = self.in_nested_sets_from_output_fn(self, output, lam)
        upper_edge = torch.maximum(upper_edge, prediction + 1e-6)
        lower_edge = torch.minimum(lower_edge, prediction - 1e-6)
        return lower_edge, prediction, upper_edge

    def nested_sets(self, x, lam=None):
        if lam == None:
            if self.lhat == None:
                raise Exception("You have to specify lambda unless your model is already calibrated.")
            lam = self.lhat
        output = self(*x)
        return self.nested_sets_from_output(output, lam=lam)

    def set_lhat(self, lhat):
        self.lhat = lhat


def _registry():
    """uncertainty_type -> (final layer class, train loss, nested-set function): the reference's if/elif chain (:57-85)
    as a table."""
    return {
        "quantiles": (QuantileRegressionLayer, quantile_regression_loss_fn, quantile_regression_nested_sets_from_output),
        "quantiles_l1": (QuantileRegressionL1Layer, quantile_regression_l1_loss_fn, quantile_regression_l1_nested_sets_from_output),
        "gaussian": (GaussianRegressionLayer, gaussian_regression_loss_fn, gaussian_regression_nested_sets_from_output),
        "residual_magnitude": (ResidualMagnitudeLayer, residual_magnitude_loss_fn, residual_magnitude_nested_sets_from_output),
        "residual_magnitude_l1": (ResidualMagnitudeL1Layer, residual_magnitude_l1_loss_fn,
                                  residual_magnitude_l1_nested_sets_from_output),
        "softmax": (SoftmaxLayer, softmax_loss_fn, softmax_nested_sets_from_output),
        "inn": (INNLayer, inn_loss_fn, inn_nested_sets_from_output),
    }


def add_uncertainty(model, params):
    """trunk (needs n_channels_middle / n_channels_out) + params["uncertainty_type"] -> ModelWithUncertainty."""
    try:
        layer_cls, loss_fn, sets_fn = _registry()[params["uncertainty_type"]]
    except KeyError:
        raise NotImplementedError(f"unknown uncertainty_type {params.get('uncertainty_type')!r}") from None
    head = layer_cls(model.n_channels_middle, model.n_channels_out, params)
    return ModelWithUncertainty(model, head, loss_fn, sets_fn, params)
