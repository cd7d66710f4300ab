"""CPU restatement of the reference's fastMRI input pipeline.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What the reference does per slice (core/datasets/fastmri/, vendored from facebookresearch/fastMRI), with PyTorch-CPU /
numpy as the substrate exactly like the reference:
  * column sub-sampling mask: EquispacedMaskFunc / RandomMaskFunc, subsample.py:64-202 (numpy RandomState);
  * apply_mask, transforms.py:53-85: k-space * mask + 0.0;
  * centred orthonormal inverse FFT, fftc.py:87-110: ifftshift -> ifftn(norm="ortho") -> fftshift over the two spatial dims;
  * complex_center_crop, transforms.py:130-152, to the target's extent; magnitude, math_util.py:56-70;
  * UnetDataTransform.__call__, transforms.py:252-328 (singlecoil: no RSS); target centre crop, transforms.py:108-127;
  * the affine normalisation of FastMRIDataset.__getitem__, FastMRIDataset.py:147-160.
Pinned against the imported reference by fixture G18 (tests/golden/make_golden.py g18).
"""
from __future__ import annotations

import numpy as np
import torch


def choose_acceleration(rng, center_fractions, accelerations):
    """MaskFunc.choose_acceleration, subsample.py:64-70."""
    choice = rng.randint(0, len(accelerations))
    return center_fractions[choice], accelerations[choice]


def equispaced_mask(num_cols, center_fractions, accelerations, rng):
    """EquispacedMaskFunc.__call__ body, subsample.py:177-200 -> float32 [num_cols] (1 = kept column)."""
    center_fraction, acceleration = choose_acceleration(rng, center_fractions, accelerations)
    num_low_freqs = int(round(num_cols * center_fraction))
    mask = np.zeros(num_cols, dtype=np.float32)
    pad = (num_cols - num_low_freqs + 1) // 2
    mask[pad:pad + num_low_freqs] = True
    adjusted_accel = (acceleration * (num_low_freqs - num_cols)) / (num_low_freqs * acceleration - num_cols)
    offset = rng.randint(0, round(adjusted_accel))
    accel_samples = np.arange(offset, num_cols - 1, adjusted_accel)
    accel_samples = np.around(accel_samples).astype(np.uint)
    mask[accel_samples] = True
    return mask


def random_mask(num_cols, center_fractions, accelerations, rng):
    """RandomMaskFunc.__call__ body, subsample.py:113-131."""
    center_fraction, acceleration = choose_acceleration(rng, center_fractions, accelerations)
    num_low_freqs = int(round(num_cols * center_fraction))
    prob = (num_cols / acceleration - num_low_freqs) / (num_cols - num_low_freqs)
    mask = rng.uniform(size=num_cols) < prob
    pad = (num_cols - num_low_freqs + 1) // 2
    mask[pad:pad + num_low_freqs] = True
    return mask.astype(np.float32)


def seeded_mask(kind, num_cols, center_fractions, accelerations, seed):
    """a mask function called with `seed` (temp_seed, subsample.py:15-28): a fresh RandomState seeded with it."""
    rng = np.random.RandomState()
    rng.seed(seed)
    fn = equispaced_mask if kind == "equispaced" else random_mask
    return fn(num_cols, center_fractions, accelerations, rng)


def ifft2c(data: torch.Tensor) -> torch.Tensor:
    """fftc.py:87-110 on [..., R, C, 2] real pairs."""
    x = torch.view_as_complex(data.contiguous())
    r, c = x.shape[-2], x.shape[-1]
    x = torch.roll(x, shifts=((r + 1) // 2, (c + 1) // 2), dims=(-2, -1))          # ifftshift
    x = torch.fft.ifftn(x, dim=(-2, -1), norm="ortho")
    x = torch.roll(x, shifts=(r // 2, c // 2), dims=(-2, -1))                      # fftshift
    return torch.view_as_real(x)


def center_crop(data: torch.Tensor, shape):
    """transforms.py:108-127 (last two dims)."""
    w_from = (data.shape[-2] - shape[0]) // 2
    h_from = (data.shape[-1] - shape[1]) // 2
    return data[..., w_from:w_from + shape[0], h_from:h_from + shape[1]]


def unet_data_transform(kspace: torch.Tensor, mask: torch.Tensor, crop):
    """UnetDataTransform.__call__ for singlecoil data, transforms.py:286-312: kspace [..., R, C, 2], mask [C] -> magnitude
    image [..., crop0, crop1] of the zero-filled reconstruction."""
    masked = kspace * mask.reshape(-1, 1) + 0.0
    image = ifft2c(masked)
    crop = tuple(crop)
    if image.shape[-2] < crop[1]:                                                  # the FLAIR-203 rule, transforms.py:301-302
        crop = (image.shape[-2], image.shape[-2])
    w_from = (image.shape[-3] - crop[0]) // 2
    h_from = (image.shape[-2] - crop[1]) // 2
    image = image[..., w_from:w_from + crop[0], h_from:h_from + crop[1], :]
    return (image ** 2).sum(dim=-1).sqrt()


def normalize(v: torch.Tensor, sub: float, div: float):
    """FastMRIDataset.__getitem__, FastMRIDataset.py:147-158: (v - mean)/std or (v - min)/max."""
    return (v - sub) / div


def det_kspace(b, r, c, salt=0):
    """deterministic synthetic k-space [b, r, c, 2] with the energy profile of an MR image (strong centre, 1/f decay),
    shared by the golden generator and the GPU tests so that full-size inputs never need storing."""
    idx = torch.arange(b * r * c * 2, dtype=torch.float64).reshape(b, r, c, 2)
    noise = torch.sin(0.7391 * idx * (1 + (idx % 11)) + 0.31 * salt) + 0.5 * torch.cos(0.1273 * idx + salt)
    yy = (torch.arange(r, dtype=torch.float64) - r // 2).reshape(1, r, 1, 1)
    xx = (torch.arange(c, dtype=torch.float64) - c // 2).reshape(1, 1, c, 1)
    env = 1.0 / (1.0 + 0.02 * (yy * yy + xx * xx)) ** 0.75
    return (1e-4 * noise * (0.05 + 50.0 * env)).to(torch.float32)
