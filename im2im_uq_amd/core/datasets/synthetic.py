"""Synthetic map-style dataset with the reference's Dataset contract (README.md:81): __getitem__ returns
(input C x H x W, target C x H x W) float tensors.  fastMRI-shaped by default (n_in = 1, 320 x 320, input
"standard"-normalised, target in [0,1]).  The real loaders (fastMRI / TEMCA / BSBCM) are out of this build's
scope (they need h5py / imageio / unreleased data); any torch Dataset honouring the
contract plugs into train_net / calibrate_model unchanged."""
import torch
from torch.utils.data import Dataset


class SyntheticDenoiseDataset(Dataset):
    def __init__(self, num_images=64, num_inputs=1, side=320, noise=0.1, seed=0):
        g = torch.Generator().manual_seed(seed)
        # smooth-ish targets in [0,1]: low-resolution noise upsampled, then min-max normalised per image
        low = torch.rand((num_images, 1, max(side // 8, 2), max(side // 8, 2)), generator=g)
        y = torch.nn.functional.interpolate(low, size=(side, side), mode="bilinear", align_corners=True)
        y = (y - y.amin(dim=(2, 3), keepdim=True)) / (y.amax(dim=(2, 3), keepdim=True) - y.amin(dim=(2, 3), keepdim=True) + 1e-8)
        x = y + noise * torch.randn((num_images, num_inputs, side, side), generator=g)
        self.x = (x - x.mean()) / x.std()
        self.y = y

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]
