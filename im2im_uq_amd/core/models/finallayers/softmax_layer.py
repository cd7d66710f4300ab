"""Softmax (binned classification) final layer -- drop-in for the reference's
core/models/finallayers/softmax_layer.py (layer :7-14, loss :15-25, nested sets :27-53).

Like the reference's loss (CrossEntropyLoss over dim 1 of a [B,K,1,H,W] tensor) this supports n_channels_out == 1.
The class logits live NHWC in the compute dtype; the [B,K,1,H,W] tensor handed out is a strided view of that buffer."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import hip_ops, nn_ops
from ._common import fused_nested_sets, lam_value


class SoftmaxLayer(nn.Module):
    def __init__(self, n_channels_middle, n_channels_out, params):
        super(SoftmaxLayer, self).__init__()
        self.num_softmax = params["num_softmax"]
        if n_channels_out != 1:
            raise NotImplementedError("SoftmaxLayer: n_channels_out must be 1 (the reference's softmax_loss_fn only type-checks for one output channel)")
        if not 2 <= self.num_softmax <= 64:
            raise NotImplementedError("SoftmaxLayer: num_softmax must be in [2, 64]")
        self.output_layers = nn.ModuleList([nn.Conv2d(n_channels_middle, self.num_softmax, kernel_size=3, padding=1)
                                            for i in range(n_channels_out)])
        self.compute_dtype = None

    def forward(self, x):
        cdt = getattr(self, "compute_dtype", None) or nn_ops.get_compute_dtype()
        if x.dtype in (torch.float32, torch.bfloat16) and x.permute(0, 2, 3, 1).is_contiguous():
            cdt = x.dtype
        conv = self.output_layers[0]
        logits = nn_ops.SoftmaxHead.apply(x, cdt, conv.weight, conv.bias)             # [B,H,W,S]
        out = logits.permute(0, 3, 1, 2)[:, :self.num_softmax].unsqueeze(2)            # [B,K,1,H,W] view (:14)
        out._im2im_nhwc = logits
        return out


def softmax_loss_fn(pred, target, params):
    """nn.CrossEntropyLoss()(pred, bucketize(target, linspace(0,1,K)))  (reference :15-25)."""
    if not pred.is_cuda:
        raise RuntimeError("softmax_loss_fn: tensors must be on the GPU; the HIP path has no CPU fallback")
    k = params["num_softmax"]
    logits, kk = nn_ops.logits_nhwc(pred)
    if kk != k:
        raise ValueError(f"pred has {kk} classes, params['num_softmax'] = {k}")
    classes = torch.linspace(0, 1, k, device=pred.device)
    t = target.detach().to(device=pred.device, dtype=torch.float32).contiguous()
    if t.numel() != logits.numel() // logits.shape[-1]:
        raise ValueError("target shape does not match pred")
    return nn_ops.SoftmaxCE.apply(logits, t, k, classes)


def softmax_sets_summary(output):
    """[b,K,1,H,W] class logits -> [b,3,1,H,W] (lower quantile, prediction, upper quantile); a tensor that already is
    such a summary passes through (calibration keeps only the summary of each batch in HBM: 12 B/px instead of 4K B/px)."""
    if output.dim() == 5 and output.shape[1] == 3 and getattr(output, "_im2im_nhwc", None) is None:
        return output
    return nn_ops.softmax_sets_summary(output)


def softmax_nested_sets_from_output(model, output, lam=None, _floor=False):
    """prediction -+ lam * relu(prediction - lower quantile | upper quantile - prediction) from the softmax over the
    class logits (reference :27-53)."""
    with torch.no_grad():
        lam = lam_value(model, lam)
        if not output.is_cuda:
            raise RuntimeError("nested sets: the output must be on the GPU; the HIP path has no CPU fallback")
        return fused_nested_sets(softmax_sets_summary(output), lam, hip_ops.SETS_SOFTMAX, _floor)


softmax_nested_sets_from_output.im2im_sets_form = hip_ops.SETS_SOFTMAX
softmax_nested_sets_from_output.im2im_summarize = softmax_sets_summary
