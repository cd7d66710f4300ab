"""Centred orthonormal 2-D (inverse) DFT on the GPU -- the counterpart of the reference's core/datasets/fastmri/fftc.py
(fft2c_new :60-84, ifft2c_new :87-110) for [..., R, C, 2] real-pair tensors.

Evaluated as two dense contractions on the exact-fp32 MFMA GEMM (im2im_conv_fwd with taps = 1):  out = W_R . X . W_C^T
with  W[j, p] = exp(+-2 pi i (p + s_i)(j - s_f) / n) / sqrt(n),  s_i = (n+1)//2 (the ifftshift), s_f = n//2 (the fftshift)
-- the shifts and the normalisation live in the matrices.  `crop=(h, w)` evaluates only the centre h x w outputs
(complex_center_crop folded in), which is what makes the dense form cheap for fastMRI: 320 x 320 of 640 x 368.
Why not an FFT library: rocFFT is not part of this library's C ABI, fastMRI widths are 368 = 16*23 and 372 = 4*3*31
(Bluestein territory), and at these sizes the pruned DFT is ~1 GFLOP per slice = ~10 us on the fp32 matrix cores,
two orders of magnitude below the training step it feeds."""
import math

import numpy as np
import torch

from .... import nn_ops
from ...._lib import check, dptr, lib, stream_ptr

F32 = torch.float32
_cache = {}


def _pad(n, m):
    return (n + m - 1) // m * m


def centred_dft_matrix(n, out_from, out_count, k_pad, n_pad, inverse, device):
    """real [n_pad, k_pad] matrix M with  (M @ [re0, im0, re1, im1, ...])[2j, 2j+1] = (re, im) of output j + out_from of the
    centred orthonormal (inverse) DFT of length n; float64 twiddles with the angle reduced in integers."""
    key = (n, out_from, out_count, k_pad, n_pad, inverse, str(device))
    m = _cache.get(key)
    if m is None:
        s_i, s_f = (n + 1) // 2, n // 2
        p = np.arange(n, dtype=np.int64)[None, :]
        j = (np.arange(out_count, dtype=np.int64) + out_from)[:, None]
        k = ((p + s_i) * (j - s_f)) % n
        ang = (2.0 * math.pi / n) * k.astype(np.float64) * (1.0 if inverse else -1.0)
        wr, wi = np.cos(ang) / math.sqrt(n), np.sin(ang) / math.sqrt(n)
        full = np.zeros((n_pad, k_pad), dtype=np.float64)
        full[0:2 * out_count:2, 0:2 * n:2] = wr
        full[0:2 * out_count:2, 1:2 * n:2] = -wi
        full[1:2 * out_count:2, 0:2 * n:2] = wi
        full[1:2 * out_count:2, 1:2 * n:2] = wr
        if len(_cache) > 32:
            _cache.clear()
        m = _cache[key] = torch.from_numpy(full).to(device=device, dtype=F32).contiguous()
    return m


def gemm_rows(a, w):
    """a [M, K] @ w [N, K]^T -> [M, N], fp32, on the MFMA implicit-GEMM kernel (a 1x1 convolution over M 'pixels');
    M % 16 == 0, K % 32 == 0, N % 32 == 0."""
    m, k = a.shape
    return nn_ops.conv_fwd(a.view(1, m // 16, 16, k), w.view(w.shape[0], 1, k)).view(m, w.shape[0])


def _dft2c(data, inverse, crop=None, mask=None):
    if data.shape[-1] != 2:
        raise ValueError("Tensor does not have separate complex dim.")
    if not data.is_cuda:
        raise RuntimeError("fftc: tensors must be on the GPU; the HIP path has no CPU fallback")
    lead = data.shape[:-3]
    r, c = data.shape[-3], data.shape[-2]
    x = data.reshape(-1, r, c, 2).to(F32).contiguous()
    b = x.shape[0]
    h, w = (r, c) if crop is None else crop
    y_from, x_from = (r - h) // 2, (c - w) // 2
    dev = x.device
    k1, n1 = _pad(2 * c, 32), _pad(2 * w, 32)
    k2, n2 = _pad(2 * r, 32), _pad(2 * h, 32)
    rows1, rows2 = _pad(b * r, 16), _pad(b * w, 16)
    a1 = torch.empty((rows1, k1), dtype=F32, device=dev)
    if mask is None:
        mask_t, stride = torch.ones((c,), dtype=F32, device=dev), 0
    else:
        mask_t = mask.to(device=dev, dtype=F32).reshape(-1, c).contiguous()
        stride = c if mask_t.shape[0] > 1 else 0
        if mask_t.shape[0] not in (1, b):
            raise ValueError(f"mask must hold 1 or {b} rows of {c} columns")
    st = stream_ptr(dev)
    check(lib.im2im_fastmri_mask_pack(dptr(x), dptr(mask_t), stride, dptr(a1), b, r, c, rows1, k1, st), "im2im_fastmri_mask_pack")
    t1 = gemm_rows(a1, centred_dft_matrix(c, x_from, w, k1, n1, inverse, dev))          # [(b, r)][(x, ri)]
    a2 = torch.empty((rows2, k2), dtype=F32, device=dev)
    if rows2 > b * w:
        a2[b * w:].zero_()
    check(lib.im2im_complex_transpose(dptr(t1), dptr(a2), b, r, w, n1, k2, st), "im2im_complex_transpose")
    t2 = gemm_rows(a2, centred_dft_matrix(r, y_from, h, k2, n2, inverse, dev))          # [(b, x)][(y, ri)]
    return t2, (lead, b, h, w, n2)


def _finish_complex(t2, meta):
    lead, b, h, w, n2 = meta
    out = torch.empty((b, h, w, 2), dtype=F32, device=t2.device)
    check(lib.im2im_complex_transpose(dptr(t2), dptr(out), b, w, h, n2, 2 * w, stream_ptr(t2.device)), "im2im_complex_transpose")
    return out.reshape(*lead, h, w, 2)


def ifft2c_new(data, crop=None):
    """centred orthonormal inverse 2-D DFT of [..., R, C, 2] (reference fftc.py:87-110); crop=(h, w): only the centre."""
    return _finish_complex(*_dft2c(data, True, crop))


def fft2c_new(data, crop=None):
    """centred orthonormal forward 2-D DFT (reference fftc.py:60-84)."""
    return _finish_complex(*_dft2c(data, False, crop))


def masked_ifft2c_abs(kspace, mask, crop, sub=0.0, div=1.0):
    """the whole UnetDataTransform arithmetic for single-coil data in one go: ((|ifft2c(kspace * mask)| cropped) - sub) / div
    -> [..., h, w] (transforms.py:286-307 + FastMRIDataset.py:147-150)."""
    t2, (lead, b, h, w, n2) = _dft2c(kspace, True, crop, mask)
    out = torch.empty((b, h, w), dtype=F32, device=t2.device)
    check(lib.im2im_fastmri_abs_normalize(dptr(t2), dptr(out), b, w, h, n2, float(sub), float(div), stream_ptr(t2.device)),
          "im2im_fastmri_abs_normalize")
    return out.reshape(*lead, h, w)


def center_crop_affine(data, shape, sub=0.0, div=1.0):
    """(center_crop(data, shape) - sub) / div for real [..., Hin, Win] tensors (transforms.py:108-127 + FastMRIDataset.py:152-158)."""
    if not (0 < shape[0] <= data.shape[-2] and 0 < shape[1] <= data.shape[-1]):
        raise ValueError("Invalid shapes.")
    if not data.is_cuda:
        raise RuntimeError("center_crop_affine: tensors must be on the GPU; the HIP path has no CPU fallback")
    lead = data.shape[:-2]
    x = data.reshape(-1, data.shape[-2], data.shape[-1]).to(F32).contiguous()
    out = torch.empty((x.shape[0], shape[0], shape[1]), dtype=F32, device=x.device)
    check(lib.im2im_center_crop_affine(dptr(x), dptr(out), x.shape[0], x.shape[1], x.shape[2], shape[0], shape[1], float(sub),
                                       float(div), stream_ptr(x.device)), "im2im_center_crop_affine")
    return out.reshape(*lead, *shape)
