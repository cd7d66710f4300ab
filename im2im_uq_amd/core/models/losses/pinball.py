"""PinballLoss -- drop-in for the reference's core/models/losses/pinball.py (:4-26), computed by the fused
HIP loss kernel (a single pinball term is the quantile loss with weights (1, 0, 0))."""
import torch

from ... import _pkg  # noqa: F401
from .... import nn_ops


class PinballLoss():

    def __init__(self, quantile=0.10, reduction='mean'):
        self.quantile = quantile
        assert 0 < self.quantile
        assert self.quantile < 1
        self.reduction = reduction

    def __call__(self, output, target):
        assert output.shape == target.shape
        if not output.is_cuda:
            raise RuntimeError("PinballLoss: tensors must be on the GPU; the HIP path has no CPU fallback")
        o = output.to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        n = o.numel()
        loss = nn_ops.QuantileLoss.apply(o, o.detach(), o.detach(), t, 0, 1, n, float(self.quantile), 0.5, 1.0, 0.0, 0.0)
        if self.reduction == 'sum':
            loss = loss * n
        return loss
