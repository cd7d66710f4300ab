"""Single-coil fastMRI sample transform on the GPU -- drop-in for the pieces of the reference's
core/datasets/fastmri/transforms.py that its FastMRIDataset uses: to_tensor (:19-34), apply_mask (:53-85), center_crop
(:108-127), complex_center_crop (:130-152) and UnetDataTransform (:217-328).  The arithmetic (mask, centred inverse DFT,
crop, magnitude) runs in HIP kernels (fftc.py); `UnetDataTransform.batch` does a whole batch of slices in one go."""
import numpy as np
import torch

from . import fftc
from .subsample import MaskFunc  # noqa: F401  (re-exported like the reference)


def to_tensor(data, device=None):
    """numpy (complex) array -> torch tensor with a trailing real/imag axis (reference :19-34), optionally on `device`."""
    if np.iscomplexobj(data):
        data = np.stack((data.real, data.imag), axis=-1)
    t = torch.from_numpy(data)
    return t.to(device) if device is not None else t


def apply_mask(data, mask_func, seed=None, padding=None):
    """(data * mask + 0.0, mask) (reference :53-85); `data` [..., R, C, 2]."""
    shape = np.array(data.shape)
    shape[:-3] = 1
    mask = mask_func(shape, seed)
    if padding is not None:
        mask[:, :, :padding[0]] = 0
        mask[:, :, padding[1]:] = 0
    return data * mask.to(data.device) + 0.0, mask


def center_crop(data, shape):
    if not (0 < shape[0] <= data.shape[-2] and 0 < shape[1] <= data.shape[-1]):
        raise ValueError("Invalid shapes.")
    y0, x0 = (data.shape[-2] - shape[0]) // 2, (data.shape[-1] - shape[1]) // 2
    return data[..., y0:y0 + shape[0], x0:x0 + shape[1]]


def complex_center_crop(data, shape):
    if not (0 < shape[0] <= data.shape[-3] and 0 < shape[1] <= data.shape[-2]):
        raise ValueError("Invalid shapes.")
    y0, x0 = (data.shape[-3] - shape[0]) // 2, (data.shape[-2] - shape[1]) // 2
    return data[..., y0:y0 + shape[0], x0:x0 + shape[1], :]


class UnetDataTransform:
    """(kspace, mask, target, attrs, fname, slice_num) -> (image, target, mean, std, fname, slice_num, max_value), as the
    reference's (:252-328) for which_challenge == "singlecoil"; tensors live on `device`."""

    def __init__(self, which_challenge, mask_func=None, use_seed=True, device="cuda"):
        if which_challenge not in ("singlecoil", "multicoil"):
            raise ValueError("Challenge should either be 'singlecoil' or 'multicoil'")
        if which_challenge == "multicoil":
            raise NotImplementedError("the HIP pipeline covers the single-coil path the reference's FastMRIDataset uses")
        self.mask_func = mask_func
        self.which_challenge = which_challenge
        self.use_seed = use_seed
        self.device = device

    def _mask_columns(self, num_cols, fname):
        seed = None if not self.use_seed else tuple(map(ord, fname))
        return self.mask_func((1, num_cols, 2), seed).reshape(-1)

    @staticmethod
    def _crop_size(rows, cols, target_shape, attrs):
        crop = (target_shape[-2], target_shape[-1]) if target_shape is not None else (attrs["recon_size"][0], attrs["recon_size"][1])
        if cols < crop[1]:                                     # "FLAIR 203": image narrower than the requested width
            crop = (cols, cols)
        return crop

    def batch(self, kspace, fnames, target_shape=None, attrs=None, masks=None, sub=0.0, div=1.0):
        """kspace [B, R, C, 2] (GPU) of equally shaped slices -> ((image - sub) / div) [B, h, w]: one mask per slice (drawn on
        the host like the reference, or given as `masks` [B, C]), then the fused GPU transform."""
        b, r, c = kspace.shape[0], kspace.shape[1], kspace.shape[2]
        if masks is None and self.mask_func is not None:
            masks = torch.stack([self._mask_columns(c, f) for f in fnames])
        crop = self._crop_size(r, c, target_shape, attrs or {})
        return fftc.masked_ifft2c_abs(kspace, masks, crop, sub, div), masks

    def __call__(self, kspace, mask, target, attrs, fname, slice_num):
        kspace = to_tensor(kspace, self.device).to(torch.float32)
        max_value = attrs["max"] if "max" in attrs.keys() else 0.0
        cols = None
        if self.mask_func and mask is None:
            cols = self._mask_columns(kspace.shape[-2], fname)
        crop = self._crop_size(kspace.shape[-3], kspace.shape[-2], target.shape if target is not None else None, attrs)
        image = fftc.masked_ifft2c_abs(kspace, cols, crop)
        if target is not None:
            target = center_crop(to_tensor(target, self.device), crop)
        else:
            target = torch.Tensor([0])
        return image, target, None, None, fname, slice_num, max_value
