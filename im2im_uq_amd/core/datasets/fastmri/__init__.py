"""GPU counterpart of the reference's core/datasets/fastmri package (the parts its FastMRIDataset uses)."""
from ... import _pkg as _unused  # noqa: F401
from .FastMRIDataset import FastMRIDataset, SyntheticKspaceDataset  # noqa: F401
from .fftc import fft2c_new as fft2c  # noqa: F401
from .fftc import ifft2c_new as ifft2c  # noqa: F401
from .fftc import center_crop_affine, masked_ifft2c_abs  # noqa: F401


def complex_abs(data):
    """sqrt(re^2 + im^2) over the trailing real/imag axis (reference math_util.py:56-70)."""
    if not data.shape[-1] == 2:
        raise ValueError("Tensor does not have separate complex dim.")
    return (data ** 2).sum(dim=-1).sqrt()
