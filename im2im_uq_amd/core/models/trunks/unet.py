"""The fixed 4-level UNet trunk -- drop-in for the reference's core/models/trunks/unet.py (:10-46):
same constructor, attributes (n_channels_in / n_channels_middle = 32 / n_channels_out / bilinear), module
names and forward order.  forward takes [B, n_in, H, W] fp32 and returns the 32-channel feature map
(logically [B,32,H,W], channels-last in the compute dtype)."""
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .unet_parts import DoubleConv, Down, OutConv, Up


_ENCODER = (("down1", 64, 128), ("down2", 128, 256), ("down3", 256, 512))     # name, channels in, channels out (:21-23)


class UNet(nn.Module):
    def __init__(self, n_channels_in, n_channels_out, bilinear=True):
        super(UNet, self).__init__()
        self.n_channels_in, self.n_channels_middle, self.n_channels_out = n_channels_in, 32, n_channels_out
        self.bilinear = bilinear
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels_in, 64)                       # registration order == the reference's state_dict order
        for name, cin, cout in _ENCODER:
            setattr(self, name, Down(cin, cout))
        self.down4 = Down(512, 1024 // factor)
        for level, (cin, cout) in enumerate(((1024, 512 // factor), (512, 256 // factor), (256, 128 // factor), (128, 64)), 1):
            setattr(self, f"up{level}", Up(cin, cout, bilinear))
        self.out = OutConv(64, self.n_channels_middle)

    def forward(self, x):
        # lazy=True: between these blocks activations stay "pre-BatchNorm + (scale, shift)"; every consumer below is one
        # of this package's kernels and applies BatchNorm+ReLU on the fly (see DoubleConv.forward).
        # pool=True: a skip block also hands back MaxPool2d(2) of its output for the next Down block (pooled=True), so the
        # pooling's backward and the skip-gradient accumulation fold into that block's BatchNorm backward kernels.
        skips = []
        feat, pooled = self.inc(x, lazy=True, pool=True)
        skips.append(feat)
        for name, _, _ in _ENCODER:
            feat, pooled = getattr(self, name)(pooled, lazy=True, pool=True, pooled=True)
            skips.append(feat)
        h = self.down4(pooled, lazy=True, pooled=True)
        for level in range(1, 5):                                      # up1(x5, x4) ... up4(., x1)  (:40-43)
            h = getattr(self, f"up{level}")(h, skips[4 - level], lazy=True)
        return self.out(h)
