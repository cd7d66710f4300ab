"""Shared plumbing of the final-layer mirrors: lambda lookup, the fused nested-set call and the packed loss call."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import hip_ops, nn_ops


def lam_value(model, lam):
    if lam == None:
        if model.lhat == None:
            raise Exception("You have to specify lambda unless your model is already calibrated.")
        lam = model.lhat
    return lam


def fused_nested_sets(output, lam, form, floor, clamp_inplace=False):
    """(lower_edge, prediction, upper_edge) of a GPU output [b,K,C,H,W] in one HIP kernel (same fp32 op order as the
    reference's expression; floor=True adds ModelWithUncertainty's +-1e-6 floor)."""
    if not output.is_contiguous():
        raise ValueError("nested sets: model output must be contiguous")
    lam_f = float(torch.as_tensor(lam, dtype=torch.float32))
    return hip_ops.nested_sets(output, lam_f, clamp_inplace=clamp_inplace, form=form, floor=floor)


def packed_loss(pred, target, planes, kind, q_lo=0.0, q_hi=0.0, w0=1.0, w1=1.0, w2=1.0, who="loss_fn"):
    if not pred.is_cuda:
        raise RuntimeError(f"{who}: tensors must be on the GPU; the HIP path has no CPU fallback")
    if pred.dim() != 5 or pred.shape[1] != planes:
        raise ValueError(f"pred must be [B,{planes},C,H,W], got {tuple(pred.shape)}")
    p = pred if (pred.dtype == torch.float32 and pred.is_contiguous()) else pred.to(torch.float32).contiguous()
    t = target.detach().to(device=pred.device, dtype=torch.float32).contiguous()
    if t.numel() != p.shape[0] * p[0, 0].numel():
        raise ValueError("target shape does not match pred")
    return nn_ops.UQLossPacked.apply(p, t, kind, float(q_lo), float(q_hi), float(w0), float(w1), float(w2))


class TwoHeadLayer(nn.Module):
    """two 3x3 heads -> [B,2,C,H,W] fp32 with an activation on the second one; the nn.Conv2d are parameter containers
    (reference names and initialisation), the forward is one fused HIP kernel."""
    _names = ("first", "second")
    _act = None

    def __init__(self, n_channels_middle, n_channels_out, params):
        super().__init__()
        self.params = params
        for name in self._names:
            setattr(self, name, nn.Conv2d(n_channels_middle, n_channels_out, kernel_size=3, padding=1))
        self.compute_dtype = None

    def forward(self, x):
        cdt = getattr(self, "compute_dtype", None) or nn_ops.get_compute_dtype()
        if x.dtype in (torch.float32, torch.bfloat16) and x.permute(0, 2, 3, 1).is_contiguous():
            cdt = x.dtype                      # consume the trunk's channels-last feature map zero-copy
        a, b = (getattr(self, n) for n in self._names)
        out = nn_ops.Heads.apply(x, cdt, self._act, a.weight, a.bias, b.weight, b.bias)
        if self._act == "relu":
            out._im2im_var_nonneg = True       # lets gaussian_regression_loss_fn skip its per-step host sync
        return out
