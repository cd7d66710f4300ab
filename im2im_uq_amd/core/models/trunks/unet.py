"""The fixed 4-level UNet trunk -- drop-in for the reference's core/models/trunks/unet.py (:10-46):
same constructor, attributes (n_channels_in / n_channels_middle = 32 / n_channels_out / bilinear), module
names and forward order.  forward takes [B, n_in, H, W] fp32 and returns the 32-channel feature map
(logically [B,32,H,W], channels-last in the compute dtype)."""
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .unet_parts import DoubleConv, Down, OutConv, Up


class UNet(nn.Module):
    def __init__(self, n_channels_in, n_channels_out, bilinear=True):
        super(UNet, self).__init__()
        self.n_channels_in = n_channels_in
        self.n_channels_middle = 32
        self.n_channels_out = n_channels_out
        self.bilinear = bilinear
        factor = 2 if bilinear else 1

        self.inc = DoubleConv(n_channels_in, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024 // factor)

        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.out = OutConv(64, self.n_channels_middle)

    def forward(self, x):
        # lazy=True: between these blocks activations stay "pre-BatchNorm + (scale, shift)"; every consumer below is one
        # of this package's kernels and applies BatchNorm+ReLU on the fly (see DoubleConv.forward)
        # pool=True: a skip block also hands back MaxPool2d(2) of its output for the next Down block (pooled=True), so the
        # pooling's backward and the skip-gradient accumulation fold into that block's BatchNorm backward kernels
        x1, p1 = self.inc(x, lazy=True, pool=True)
        x2, p2 = self.down1(p1, lazy=True, pool=True, pooled=True)
        x3, p3 = self.down2(p2, lazy=True, pool=True, pooled=True)
        x4, p4 = self.down3(p3, lazy=True, pool=True, pooled=True)
        x5 = self.down4(p4, lazy=True, pooled=True)

        x = self.up1(x5, x4, lazy=True)
        x = self.up2(x, x3, lazy=True)
        x = self.up3(x, x2, lazy=True)
        x = self.up4(x, x1, lazy=True)
        return self.out(x)
