"""INNLoss -- drop-in for the reference's core/models/losses/inn.py (:4-21) on the fused loss kernel."""
import torch

from ... import _pkg  # noqa: F401
from .... import nn_ops


class INNLoss():
    """mean (or sum) of relu(target - upper)^2 + relu(lower - target)^2 + beta*|upper - lower|."""

    def __init__(self, beta=0.10, reduction='mean'):
        if beta < 0:
            raise AssertionError("beta must be non-negative")
        self.beta, self.reduction = beta, reduction

    def __call__(self, lower, upper, target):
        if not (target.shape == lower.shape == upper.shape):
            raise AssertionError("lower, upper and target must have one shape")
        if not lower.is_cuda:
            raise RuntimeError("INNLoss: tensors must be on the GPU; the HIP path has no CPU fallback")
        n = lower.shape[0] if lower.dim() > 1 else 1
        # the fused kernel takes (lower, prediction, upper) planes; a prediction equal to the target switches its MSE term off
        t = target.detach().to(torch.float32).reshape(n, -1)
        pred = torch.stack([lower.to(torch.float32).reshape(n, -1), t, upper.to(torch.float32).reshape(n, -1)], dim=1).unsqueeze(2).unsqueeze(2)
        loss = nn_ops.UQLossPacked.apply(pred.contiguous(), t.contiguous(), nn_ops.LOSS_INN, float(self.beta), 0.0, 1.0, 1.0, 1.0)
        return loss * t.numel() if self.reduction == 'sum' else loss
