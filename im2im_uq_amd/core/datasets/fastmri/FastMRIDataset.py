"""fastMRI knee single-coil dataset with the GPU transform -- the counterpart of the reference's
core/datasets/fastmri/FastMRIDataset.py (:50-163): same constructor arguments, the same (input 1xHxW, target 1xHxW)
samples, the same `norm_params` protocol with core/datasets/utils.normalize_dataset.  Reading the HDF5 volumes needs h5py
(absent from the build image: construction then raises ImportError); the transform itself is exercised on synthetic
k-space (SyntheticKspaceDataset below, tests/test_fastmri_gpu.py)."""
import os
import random
import xml.etree.ElementTree as etree
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from . import subsample, transforms


def et_query(root, qlist, namespace="http://www.ismrm.org/ISMRMRD"):
    path = "." + "".join(f"//ismrmrd_namespace:{el}" for el in qlist)
    value = root.find(path, {"ismrmrd_namespace": namespace})
    if value is None:
        raise RuntimeError("Element not found")
    return str(value.text)


class _NormalisedSlices(Dataset):
    """shared tail of __getitem__ (reference :147-163): affine normalisation by `norm_params`, channel axis."""
    norm_params = None
    normalize_input = normalize_output = None

    def _affine(self, which, kind):
        p = self.norm_params
        if p is None:
            return 0.0, 1.0
        if kind == "standard":
            return float(p[which + "_mean"]), float(p[which + "_std"])
        if kind == "min-max":
            return float(p[which + "_min"]), float(p[which + "_max"])
        return 0.0, 1.0


class FastMRIDataset(_NormalisedSlices):
    def __init__(self, path, normalize_input, normalize_output, mask_info, num_volumes=None, slice_sample_period=1, device="cuda"):
        try:
            import h5py  # noqa: F401
        except ImportError as e:                                # pragma: no cover - the build image has no h5py
            raise ImportError("FastMRIDataset reads the fastMRI HDF5 volumes and needs h5py") from e
        self.h5py = h5py
        self.norm_params = None
        self.challenge = 'singlecoil'
        self.recons_key = "reconstruction_esc"
        self.cache_path = os.path.join(path, '.cache/')
        os.makedirs(self.cache_path, exist_ok=True)
        files = list(Path(path).iterdir())
        random.shuffle(files)
        files = files[0:num_volumes] if (num_volumes and num_volumes < len(files)) else files
        self.examples = []
        for fname in files:
            if 'cache' in str(fname):
                continue
            metadata, num_slices = self._retrieve_metadata(fname)
            assert num_slices > slice_sample_period
            self.examples += [(fname, s, metadata) for s in range(0, num_slices, slice_sample_period)]
        random.shuffle(self.examples)
        mask_func = subsample.create_mask_for_mask_type(mask_info['type'], mask_info['center_fraction'], mask_info['acceleration'])
        self.transform = transforms.UnetDataTransform(self.challenge, mask_func=mask_func, use_seed=False, device=device)
        self.normalize_input = normalize_input
        self.normalize_output = normalize_output
        self.device = device

    def _retrieve_metadata(self, fname):
        with self.h5py.File(fname, "r") as hf:
            root = etree.fromstring(hf["ismrmrd_header"][()])
            enc = ["encoding", "encodedSpace", "matrixSize"]
            enc_size = tuple(int(et_query(root, enc + [a])) for a in "xyz")
            rec = ["encoding", "reconSpace", "matrixSize"]
            recon_size = tuple(int(et_query(root, rec + [a])) for a in "xyz")
            lims = ["encoding", "encodingLimits", "kspace_encoding_step_1"]
            center = int(et_query(root, lims + ["center"]))
            maximum = int(et_query(root, lims + ["maximum"])) + 1
            padding_left = enc_size[1] // 2 - center
            num_slices = hf["kspace"].shape[0]
        return {"padding_left": padding_left, "padding_right": padding_left + maximum, "encoding_size": enc_size,
                "recon_size": recon_size}, num_slices

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, idx):
        fname, dataslice, metadata = self.examples[idx]
        with self.h5py.File(fname, "r") as hf:
            kspace = hf["kspace"][dataslice]
            # test / challenge volumes are already sub-sampled and carry their mask: the transform then applies none (reference :136)
            mask = np.asarray(hf["mask"]) if "mask" in hf else None
            target = hf[self.recons_key][dataslice] if self.recons_key in hf else None
            attrs = dict(hf.attrs)
            attrs.update(metadata)
        image, target = self.transform(kspace, mask, target, attrs, fname.name, dataslice)[:2]
        sub, div = self._affine("input", self.normalize_input)
        tsub, tdiv = self._affine("output", self.normalize_output)
        return ((image - sub) / div).unsqueeze(0), ((target - tsub) / tdiv).unsqueeze(0)


class SyntheticKspaceDataset(_NormalisedSlices):
    """fastMRI-shaped samples without the files: random k-space volumes [n, R, C, 2] held on the GPU, targets = the centre
    crop of the fully sampled reconstruction; goes through exactly the transform FastMRIDataset uses.  `batch(indices)`
    transforms many slices at once (what a GPU-side collate would call)."""

    def __init__(self, num_slices=16, rows=640, cols=368, crop=(320, 320), mask_info=None, normalize_input='standard',
                 normalize_output='min-max', device="cuda", seed=0):
        from . import fftc
        mask_info = mask_info or {'type': 'equispaced', 'center_fraction': [0.08], 'acceleration': [4]}
        g = torch.Generator(device=device).manual_seed(seed)
        yy = (torch.arange(rows, device=device, dtype=torch.float32) - rows // 2).reshape(rows, 1)
        xx = (torch.arange(cols, device=device, dtype=torch.float32) - cols // 2).reshape(1, cols)
        env = (1.0 / (1.0 + 0.02 * (yy * yy + xx * xx)) ** 0.75).reshape(1, rows, cols, 1)
        self.kspace = 1e-4 * torch.randn((num_slices, rows, cols, 2), device=device, generator=g) * (0.05 + 50.0 * env)
        self.crop = crop
        self.fnames = [f"synthetic{i // 8:07d}.h5" for i in range(num_slices)]
        self.targets = fftc.masked_ifft2c_abs(self.kspace, None, crop)
        mask_func = subsample.create_mask_for_mask_type(mask_info['type'], mask_info['center_fraction'], mask_info['acceleration'])
        self.transform = transforms.UnetDataTransform('singlecoil', mask_func=mask_func, use_seed=True, device=device)
        self.normalize_input, self.normalize_output = normalize_input, normalize_output
        self.norm_params = None

    def __len__(self):
        return self.kspace.shape[0]

    def batch(self, indices):
        idx = torch.as_tensor(indices, device=self.kspace.device)
        sub, div = self._affine("input", self.normalize_input)
        tsub, tdiv = self._affine("output", self.normalize_output)
        image, _ = self.transform.batch(self.kspace[idx], [self.fnames[i] for i in indices], target_shape=self.crop, sub=sub, div=div)
        return image.unsqueeze(1), ((self.targets[idx] - tsub) / tdiv).unsqueeze(1)

    def __getitem__(self, i):
        x, y = self.batch([i])
        return x[0], y[0]
