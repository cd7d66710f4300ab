"""Residual-magnitude final layer (L1 point loss) -- drop-in for the reference's
core/models/finallayers/residual_magnitude_l1_layer.py (layer :7-17, loss :19-25, nested sets :27-36)."""
from .... import hip_ops, nn_ops
from ._common import TwoHeadLayer, fused_nested_sets, lam_value, packed_loss


class ResidualMagnitudeL1Layer(TwoHeadLayer):
    """prediction and |residual magnitude| heads -> [B,2,C,H,W] (reference :12-17)."""
    _names = ("prediction", "residual_magnitude")
    _act = "abs"


def residual_magnitude_l1_loss_fn(pred, target, params):
    """L1(pred[:,0], y) + MSE(pred[:,1], |y - pred[:,0]|), both mean-reduced (reference :19-25); the second term
    also back-propagates into the prediction head through |y - pred[:,0]|, as autograd does for the reference."""
    return packed_loss(pred, target, 2, nn_ops.LOSS_RESIDUAL_L1, who="residual_magnitude_l1_loss_fn")


def residual_magnitude_l1_nested_sets_from_output(model, output, lam=None, _floor=False):
    """prediction -+ lam*magnitude  (reference :27-36)."""
    lam = lam_value(model, lam)
    if not output.is_cuda:
        raise RuntimeError("nested sets: the output must be on the GPU; the HIP path has no CPU fallback")
    return fused_nested_sets(output, lam, hip_ops.SETS_SCALE, _floor)


residual_magnitude_l1_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_SCALE
