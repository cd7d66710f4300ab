"""PinballLoss -- drop-in for the reference's core/models/losses/pinball.py (:4-26), computed by the fused
HIP loss kernel (a single pinball term is the quantile loss with weights (1, 0, 0))."""
import torch

from ... import _pkg  # noqa: F401
from .... import nn_ops


def _fused_pinball(output, target, q):
    """mean pinball loss of `output` against `target` at quantile q: the packed-loss kernel with only its first term on."""
    if not output.is_cuda:
        raise RuntimeError("PinballLoss: tensors must be on the GPU; the HIP path has no CPU fallback")
    pred = output.to(torch.float32).contiguous()
    frozen = pred.detach()
    return nn_ops.QuantileLoss.apply(pred, frozen, frozen, target.detach().to(torch.float32).contiguous(), 0, 1, pred.numel(),
                                     float(q), 0.5, 1.0, 0.0, 0.0)


class PinballLoss():
    """q*|e| where the prediction is below the target, (1-q)*|e| where above, 0 at ties; mean or sum over all elements."""

    def __init__(self, quantile=0.10, reduction='mean'):
        if not 0 < quantile < 1:
            raise AssertionError("quantile must lie strictly between 0 and 1")
        self.quantile, self.reduction = quantile, reduction

    def __call__(self, output, target):
        if output.shape != target.shape:
            raise AssertionError(f"shape mismatch: {tuple(output.shape)} vs {tuple(target.shape)}")
        mean = _fused_pinball(output, target, self.quantile)
        return mean * output.numel() if self.reduction == 'sum' else mean
