"""fastMRI input pipeline on the GPU -- mask x k-space -> centred inverse DFT (two exact-fp32 MFMA GEMMs)
-> crop -> magnitude -> normalise -- against the reference's UnetDataTransform (fixture G18) and, at full size and in
batches, against the CPU oracle (torch.fft, oracle/fastmri.py, itself pinned to G18).
Tolerance: 2e-5 of the image's maximum (a length-640 fp32 DFT sum vs pocketfft's fp32 FFT); masks bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def test_small_slice_matches_reference_transform():
    from im2im_uq_amd.core.datasets.fastmri import subsample, transforms
    g = load_golden("g18_fastmri_pipeline")
    tr = transforms.UnetDataTransform("singlecoil", mask_func=subsample.EquispacedMaskFunc([0.08], [4]), use_seed=True, device=DEV)
    ks = g["small_kspace"]
    kc = ks[..., 0] + 1j * ks[..., 1]
    image, target, mean, std, fname, sl, mx = tr(kc, None, g["small_target"], {"max": 1.0}, "file1000001.h5", 3)
    assert image.shape == (48, 40) and target.shape == (48, 40) and mean is None and std is None and sl == 3 and mx == 1.0
    scale = float(g["small_image"].max())
    np.testing.assert_allclose(image.cpu().numpy(), g["small_image"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(target.cpu().numpy(), g["small_target_out"], rtol=0, atol=0)
    image_f = tr(kc, None, None, {"recon_size": (64, 80, 1)}, "file1000001.h5", 3)[0]         # the FLAIR-203 crop rule
    assert image_f.shape == (72, 72)
    np.testing.assert_allclose(image_f.cpu().numpy(), g["small_image_flair"], rtol=0, atol=2e-5 * scale)


@pytest.mark.parametrize("cols", [368, 372])
def test_full_size_knee_shapes_match_reference(cols):
    """640 x 368 and 640 x 372 k-space (2C = 744 is not a multiple of 32: exercises the GEMM padding) -> 320 x 320."""
    from im2im_uq_amd.core.datasets.fastmri import subsample, transforms
    from oracle import fastmri as ofm
    g = load_golden("g18_fastmri_pipeline")
    ks = ofm.det_kspace(1, 640, cols, salt=cols)
    tr = transforms.UnetDataTransform("singlecoil", mask_func=subsample.EquispacedMaskFunc([0.08], [4]), use_seed=True, device=DEV)
    image, masks = tr.batch(ks.to(DEV), ["file1000277.h5"], target_shape=(320, 320))
    scale = float(g[f"full{cols}.max"])
    img = image[0].cpu()
    np.testing.assert_allclose(img[::4, ::4].numpy(), g[f"full{cols}.sample"], rtol=0, atol=2e-5 * scale)
    assert float(img.double().sum()) == pytest.approx(float(g[f"full{cols}.sum"]), rel=2e-5)
    assert float((img.double() ** 2).sum()) == pytest.approx(float(g[f"full{cols}.sumsq"]), rel=2e-5)
    ref = ofm.unet_data_transform(ks[0], masks[0], (320, 320))
    np.testing.assert_allclose(img.numpy(), ref.numpy(), rtol=0, atol=2e-5 * scale)


def test_batch_of_slices_with_their_own_masks_and_normalisation_vs_oracle():
    from im2im_uq_amd.core.datasets.fastmri import subsample, transforms
    from oracle import fastmri as ofm
    b, r, c = 5, 640, 368
    ks = ofm.det_kspace(b, r, c, salt=3)
    fnames = [f"file10{i:05d}.h5" for i in range(b)]
    tr = transforms.UnetDataTransform("singlecoil", mask_func=subsample.RandomMaskFunc([0.08], [4]), use_seed=True, device=DEV)
    sub, div = 1.3e-5, 7.7e-5
    image, masks = tr.batch(ks.to(DEV), fnames, target_shape=(320, 320), sub=sub, div=div)
    assert image.shape == (b, 320, 320) and masks.shape == (b, c)
    assert len({tuple(m.tolist()) for m in masks}) > 1                      # different volumes, different masks
    for i in range(b):
        want = ofm.normalize(ofm.unet_data_transform(ks[i], masks[i], (320, 320)), sub, div)
        scale = float(want.abs().max())
        np.testing.assert_allclose(image[i].cpu().numpy(), want.numpy(), rtol=0, atol=3e-5 * scale)


@pytest.mark.parametrize("shape", [(2, 96, 72), (1, 50, 34), (3, 64, 64)])
def test_ifft2c_and_fft2c_general_sizes_vs_torch_fft(shape):
    """the complex-output entry points (fftc.ifft2c / fft2c, reference fftc.py:60-110), odd-ish sizes, round trip."""
    from im2im_uq_amd.core.datasets import fastmri as hip_fm
    from oracle import fastmri as ofm
    b, r, c = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, r, c, 2, generator=g)
    got = hip_fm.ifft2c(x.to(DEV)).cpu()
    ref = ofm.ifft2c(x)
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    back = hip_fm.fft2c(got.to(DEV)).cpu()
    assert float((back - x).abs().max()) < 4e-5 * float(x.abs().max())
    crop = (r // 2, c // 2)
    got_c = hip_fm.fftc.ifft2c_new(x.to(DEV), crop=crop).cpu()
    y0, x0 = (r - crop[0]) // 2, (c - crop[1]) // 2
    assert float((got_c - ref[:, y0:y0 + crop[0], x0:x0 + crop[1]]).abs().max()) < 2e-5 * float(ref.abs().max())


def test_center_crop_affine_and_synthetic_dataset_contract():
    from im2im_uq_amd.core.datasets.fastmri import SyntheticKspaceDataset, center_crop_affine
    t = torch.arange(2 * 9 * 11, dtype=torch.float32).reshape(2, 9, 11)
    got = center_crop_affine(t.to(DEV), (4, 5), 2.0, 3.0).cpu()
    want = (t[:, 2:6, 3:8] - 2.0) / 3.0
    assert torch.equal(got, want)
    ds = SyntheticKspaceDataset(num_slices=6, rows=128, cols=80, crop=(64, 64), device=DEV)
    ds.norm_params = {"input_mean": 1e-5, "input_std": 2e-5, "output_min": 0.0, "output_max": 1e-4}
    x, y = ds[2]
    assert x.shape == (1, 64, 64) and y.shape == (1, 64, 64) and x.is_cuda and x.dtype == torch.float32
    xb, yb = ds.batch([0, 2, 5])
    assert torch.equal(xb[1], x) and torch.equal(yb[1], y)                  # eval of a slice does not depend on its batch
