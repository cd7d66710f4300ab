"""Two-encoder UNet ("WNet") -- drop-in for the reference's core/models/trunks/wnet.py (:9-59): two half-width encoder
paths, one per input channel, whose feature maps are concatenated level by level and decoded by the UNet's Up blocks.
Same constructor, attribute and sub-module names (p1inc .. p1down4, p2inc .. p2down4, up1 .. up4, out), so the
state_dict keys are the reference's.  Built from this package's DoubleConv / Down / Up / OutConv (HIP kernels)."""
import torch
import torch.nn as nn

from ... import _pkg  # noqa: F401
from .... import nn_ops
from .unet_parts import DoubleConv, Down, OutConv, Up

_WIDTHS = (32, 64, 128, 256)


class WNet(nn.Module):
    def __init__(self, n_channels_in, n_channels_out, bilinear=True):
        super(WNet, self).__init__()
        self.n_channels_in = n_channels_in
        self.n_channels_middle = 32
        self.n_channels_out = n_channels_out
        self.bilinear = bilinear
        factor = 2 if bilinear else 1
        for path in ("p1", "p2"):                              # reference :19-30 (registration order = state_dict order)
            setattr(self, path + "inc", DoubleConv(n_channels_in, _WIDTHS[0]))
            for level in range(1, 4):
                setattr(self, f"{path}down{level}", Down(_WIDTHS[level - 1], _WIDTHS[level]))
            setattr(self, path + "down4", Down(_WIDTHS[3], 512 // factor))
        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.out = OutConv(64, self.n_channels_middle)

    def _encode(self, path, x):
        feats = [getattr(self, path + "inc")(x, lazy=True)]
        for level in range(1, 5):
            feats.append(getattr(self, f"{path}down{level}")(feats[-1], lazy=True))
        return feats

    @staticmethod
    def _join(a, b):
        """torch.cat of the two paths' (lazy) activations on the channel axis (reference :47-53)."""
        return torch.cat((nn_ops.materialize(a), nn_ops.materialize(b)), dim=1)

    def forward(self, x):
        f1 = self._encode("p1", x[:, 0:1, :, :].contiguous())
        f2 = self._encode("p2", x[:, 1:2, :, :].contiguous())
        h = self._join(f1[4], f2[4])
        for i, up in enumerate((self.up1, self.up2, self.up3, self.up4)):
            h = up(h, self._join(f1[3 - i], f2[3 - i]), lazy=True)
        return self.out(h)
