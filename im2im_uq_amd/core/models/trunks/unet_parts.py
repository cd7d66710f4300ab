"""UNet building blocks on the HIP kernels -- same class names, constructor arguments, sub-module
names and state_dict keys as the reference's core/models/trunks/unet_parts.py, so reference
checkpoints load unchanged.

The nn.Conv2d / nn.BatchNorm2d objects inside `double_conv` are parameter containers (same default
initialisation as the reference); their own forward is never called.  DoubleConv.forward drives fused
kernels instead: conv (+BatchNorm partial statistics in the epilogue) -> BN finalize -> BN+ReLU apply in
training, or a single conv with BatchNorm folded into its epilogue in eval.

Activations between blocks are logically NCHW but stored channels-last in the compute dtype
(bf16 by default; `compute_dtype` / im2im_uq_amd.set_compute_dtype('fp32') for tight parity).
Supported channel counts: conv inputs with <= 8 channels (network input) or a multiple of 32;
outputs a multiple of 32 (the reference UNet uses 64..1024).
"""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import nn_ops


def _cdt(module):
    dt = getattr(module, "compute_dtype", None)
    return dt if dt is not None else nn_ops.get_compute_dtype()


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   (reference :8-25)"""

    def __init__(self, in_channels, out_channels, mid_channels=None, norm="batch", groups=32):
        """norm="batch" is the reference's block.  norm="group" (a build-side extra named by the north star)
        puts nn.GroupNorm(min(groups, channels), channels) where the reference has BatchNorm2d."""
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        if norm not in ("batch", "group"):
            raise ValueError(f"norm must be 'batch' or 'group', got {norm!r}")
        self.norm = norm

        def make_norm(ch):
            return nn.BatchNorm2d(ch) if norm == "batch" else nn.GroupNorm(min(groups, ch), ch)
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1),
            make_norm(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1),
            make_norm(out_channels),
            nn.ReLU(inplace=True)
        )
        self.compute_dtype = None

    def forward(self, x, lazy=False, x_hi=None, pool=False, tail=None):
        """tail (eval mode only): the 1x1 nn.Conv2d that is the ONLY consumer of this block's result (UNet's OutConv); when the
        fused kernel applies, the returned tensor already is that convolution's output (tagged `_im2im_tail_done`).
        x_hi: second half of the input channels when the caller did not concatenate them (Up.forward).
        pool=True: return (result, MaxPool2d(2)(result)) -- for a skip-connection block whose output also feeds the next
        Down block; in training the pooling's backward is then folded into this block's BatchNorm backward.
        lazy=True (used between this package's own blocks): in training the result is a *lazy activation* -- it
        holds the pre-BatchNorm conv output and the consumers (next conv / max-pool / upsample-concat / 1x1 conv
        kernels) apply BatchNorm+ReLU while loading it, so the normalised tensor is never written to HBM.  The
        default returns an ordinary tensor."""
        cdt = _cdt(self)
        for ci, bi in ((0, 1), (3, 4)):
            conv, bn = self.double_conv[ci], self.double_conv[bi]
            if getattr(self, "norm", "batch") == "group":
                # GroupNorm: per-image statistics from the conv epilogue, applied lazily by the block's second conv; the
                # block's own result is materialised (the pooling / upsampling / skip consumers take plain tensors)
                if x_hi is not None:
                    x = torch.cat([nn_ops.materialize(x), nn_ops.materialize(x_hi)], dim=1)
                x = nn_ops.conv_gn_relu(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.num_groups, bn.eps, cdt,
                                        lazy_out=(ci == 0))
            elif self.training:
                momentum = bn.momentum if bn.momentum is not None else 0.1
                x = nn_ops.conv_bn_relu_train(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                              bn.running_var, momentum, bn.eps, cdt, lazy_out=(lazy or ci == 0), x_hi=x_hi,
                                              pool=(pool and ci == 3), num_batches_tracked=bn.num_batches_tracked)
            else:
                needs_graph = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or bn.weight.requires_grad)
                x = nn_ops.conv_bn_relu_eval(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                             bn.running_var, bn.eps, cdt, owner=conv, x_hi=x_hi, pool=(pool and ci == 3 and not needs_graph),
                                             tail=(tail if (ci == 3 and not needs_graph and not pool) else None))
                if needs_graph:
                    # the fused eval kernel has no backward: the forward works as in the reference, but the result is tied
                    # to its inputs by a node that raises if anyone back-propagates through it (instead of silently
                    # returning activations detached from the conv / BatchNorm weights)
                    x = nn_ops.EvalModeBarrier.apply(x, conv.weight, bn.weight)
            x_hi = None
        if pool and not isinstance(x, tuple):
            x = (x, nn_ops.MaxPool2.apply(x))
        return x


class _MaxPool2(nn.Module):
    def forward(self, x):
        return nn_ops.MaxPool2.apply(x)          # accepts lazy activations


class Down(nn.Module):
    """Downscaling with maxpool then double conv   (reference :28-40)"""

    def __init__(self, in_channels, out_channels, norm="batch"):
        super().__init__()
        self.maxpool_conv = nn.Sequential(
            _MaxPool2(),
            DoubleConv(in_channels, out_channels, norm=norm)
        )

    def forward(self, x, lazy=False, pool=False, pooled=False):
        """pooled=True: x is already MaxPool2d(2) of the previous block's output (it came from that block's pool=True)."""
        if not pooled:
            x = self.maxpool_conv[0](x)
        return self.maxpool_conv[1](x, lazy=lazy, pool=pool)


class _BilinearUp(nn.Module):
    """placeholder for nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True); the interpolation is
    fused with the pad + concat in Up.forward."""
    scale_factor = 2
    mode = 'bilinear'
    align_corners = True


class Up(nn.Module):
    """Upscaling then double conv   (reference :42-69)"""

    def __init__(self, in_channels, out_channels, bilinear=True, norm="batch"):
        super().__init__()
        if bilinear:
            self.up = _BilinearUp()
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2, norm=norm)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)   # parameter container
            self.conv = DoubleConv(in_channels, out_channels, norm=norm)
        self.compute_dtype = None

    def forward(self, x1, x2, lazy=False, tail=None):
        # x1: deep feature map, x2: skip connection.  cat([x2, pad(up(x1))]) is not materialised: the first conv reads the
        # skip half in place and the upsampled half from its own tensor (odd widths fall back to one fused
        # upsample + pad + concat kernel).
        if isinstance(self.up, nn.ConvTranspose2d):
            # learned upsampling (reference :53): 1x1 MFMA conv to 4*Co channels + depth-to-space, then zero-pad to the skip
            x1 = nn_ops.ConvTranspose2x2.apply(x1, self.up.weight, self.up.bias, _cdt(self))
            dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
            if dy or dx:
                x1 = torch.nn.functional.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
            if nn_ops.SPLIT_CONCAT and x1.shape[1] == x2.shape[1] and x2.shape[1] % 64 == 0:
                return self.conv(x2, lazy=lazy, x_hi=x1, tail=tail)
            return self.conv(torch.cat([nn_ops.materialize(x2), x1], dim=1), lazy=lazy, tail=tail)
        if nn_ops.can_split_concat(x1, x2):
            up = nn_ops.Upsample2x.apply(x1, x2.shape[2], x2.shape[3])
            return self.conv(x2, lazy=lazy, x_hi=up, tail=tail)
        x = nn_ops.UpsampleConcat.apply(x1, x2)
        return self.conv(x, lazy=lazy, tail=tail)


class OutConv(nn.Module):
    """1x1 convolution   (reference :87-94)"""

    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.compute_dtype = None

    def forward(self, x):
        return nn_ops.Conv1x1.apply(x, self.conv.weight, self.conv.bias, _cdt(self))
