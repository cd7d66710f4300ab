"""The UNet trunk -- drop-in for the reference's core/models/trunks/unet.py (:10-46): same constructor, attributes
(n_channels_in / n_channels_middle = 32 / n_channels_out / bilinear), module names and forward order.  forward takes
[B, n_in, H, W] fp32 and returns the 32-channel feature map (logically [B,32,H,W], channels-last in the compute dtype).

The reference hard-codes four levels of 64/128/256/512/(1024 // factor) channels (:20-31).  Here the same recipe is
written for any depth and base width: level i has base * 2**i channels, the deepest Down halves its width when the
upsampling is bilinear, every Up halves again except the last.  The defaults (depth=4, base=64) give the reference's
network, module for module and state_dict key for key; BASELINE configs[0] ("2-level") is depth=2 and configs[3]
("deeper UNet", 1024x1024 tiles) is depth=5."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .unet_parts import DoubleConv, Down, OutConv, Up, _cdt


def unet_plan(n_channels_in, depth=4, base=64, bilinear=True):
    """[(module name, kind, channels in, channels out)] in the reference's registration order (:20-31)."""
    if depth < 1:
        raise ValueError("UNet depth must be >= 1")
    factor = 2 if bilinear else 1
    width = [base * 2 ** i for i in range(depth + 1)]
    plan = [("inc", "inc", n_channels_in, base)]
    for i in range(1, depth + 1):
        plan.append((f"down{i}", "down", width[i - 1], width[i] // factor if i == depth else width[i]))
    for k in range(1, depth + 1):
        plan.append((f"up{k}", "up", width[depth - k + 1], width[depth - k] // factor if k < depth else base))
    return plan


class UNet(nn.Module):
    def __init__(self, n_channels_in, n_channels_out, bilinear=True, depth=4, base=64, norm="batch"):
        super(UNet, self).__init__()
        self.n_channels_in, self.n_channels_middle, self.n_channels_out = n_channels_in, 32, n_channels_out
        self.bilinear = bilinear
        self.depth, self.base, self.norm = depth, base, norm      # norm="group": GroupNorm instead of BatchNorm2d (extra)
        for name, kind, cin, cout in unet_plan(n_channels_in, depth, base, bilinear):   # registration order == state_dict order
            if kind == "inc":
                setattr(self, name, DoubleConv(cin, cout, norm=norm))
            elif kind == "down":
                setattr(self, name, Down(cin, cout, norm=norm))
            else:
                setattr(self, name, Up(cin, cout, bilinear, norm=norm))
        self.out = OutConv(base, self.n_channels_middle)

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            # inference: the last block hands OutConv's 1x1 to its own conv epilogue when it can (nn_ops.conv_bn_relu_eval tail=)
            # -- unless OutConv carries its own compute dtype or anyone listens on it (forward hooks: feature extraction)
            # [r6] ... nor on the last Up block or anything inside it (fused, its conv writes OutConv's result, not its own 64-channel
            # activation), nor through a global module forward hook
            out, conv = self.out, self.out.conv
            last = getattr(self, f"up{getattr(self, 'depth', 4)}")
            import torch.nn.modules.module as _mm
            hooked = bool(_mm._global_forward_hooks) or bool(_mm._global_forward_pre_hooks) or any(
                getattr(m, a, None) for m in (out, conv, *last.modules()) for a in ("_forward_hooks", "_forward_pre_hooks"))
            last_conv = getattr(last, "conv", None)
            fuse = not hooked and last_conv is not None and _cdt(out) == _cdt(last_conv)
            h = self.features(x, tail=conv if fuse else None)
            return h if getattr(h, "_im2im_tail_done", False) else self.out(h)
        return self.out(self.features(x))

    def features(self, x, tail=None):
        """everything but the final 1x1 OutConv: the last Up block's activation."""
        # lazy=True: between these blocks activations stay "pre-BatchNorm + (scale, shift)"; every consumer below is one
        # of this package's kernels and applies BatchNorm+ReLU on the fly (see DoubleConv.forward).
        # pool=True: a skip block also hands back MaxPool2d(2) of its output for the next Down block (pooled=True), so the
        # pooling's backward and the skip-gradient accumulation fold into that block's BatchNorm backward kernels.
        depth = getattr(self, "depth", 4)                              # reference checkpoints unpickle without the attribute
        skips = []
        feat, pooled = self.inc(x, lazy=True, pool=True)
        skips.append(feat)
        for i in range(1, depth):
            feat, pooled = getattr(self, f"down{i}")(pooled, lazy=True, pool=True, pooled=True)
            skips.append(feat)
        h = getattr(self, f"down{depth}")(pooled, lazy=True, pooled=True)
        for k in range(1, depth + 1):                                  # up1(x5, x4) ... up4(., x1)  (:40-43)
            h = getattr(self, f"up{k}")(h, skips[depth - k], lazy=True, tail=tail if k == depth else None)
        return h
