"""k-space column sub-sampling masks -- drop-in for the reference's core/datasets/fastmri/subsample.py (MaskFunc :31-70,
RandomMaskFunc :73-133, EquispacedMaskFunc :136-202, create_mask_for_mask_type :205-222).  Host code like the reference's
(a few hundred numbers per call, numpy RandomState so seeded masks are the reference's bit for bit); the mask is applied
on the GPU by the transform."""
import contextlib

import numpy as np
import torch


@contextlib.contextmanager
def temp_seed(rng, seed):
    if seed is None:
        yield
        return
    state = rng.get_state()
    rng.seed(seed)
    try:
        yield
    finally:
        rng.set_state(state)


class MaskFunc:
    def __init__(self, center_fractions, accelerations):
        if len(center_fractions) != len(accelerations):
            raise ValueError("Number of center fractions should match number of accelerations")
        self.center_fractions = center_fractions
        self.accelerations = accelerations
        self.rng = np.random.RandomState()

    def choose_acceleration(self):
        choice = self.rng.randint(0, len(self.accelerations))
        return self.center_fractions[choice], self.accelerations[choice]

    def columns(self, num_cols):
        """float32 [num_cols] with 1 on the kept columns (consumes self.rng)."""
        raise NotImplementedError

    def __call__(self, shape, seed=None):
        if len(shape) < 3:
            raise ValueError("Shape should have 3 or more dimensions")
        with temp_seed(self.rng, seed):
            cols = self.columns(shape[-2])
        view = [1] * len(shape)
        view[-2] = shape[-2]
        return torch.from_numpy(cols.reshape(*view).astype(np.float32))


class RandomMaskFunc(MaskFunc):
    def columns(self, num_cols):
        center_fraction, acceleration = self.choose_acceleration()
        num_low = int(round(num_cols * center_fraction))
        prob = (num_cols / acceleration - num_low) / (num_cols - num_low)
        keep = self.rng.uniform(size=num_cols) < prob
        pad = (num_cols - num_low + 1) // 2
        keep[pad:pad + num_low] = True
        return keep.astype(np.float32)


class EquispacedMaskFunc(MaskFunc):
    def columns(self, num_cols):
        center_fraction, acceleration = self.choose_acceleration()
        num_low = int(round(num_cols * center_fraction))
        keep = np.zeros(num_cols, dtype=np.float32)
        pad = (num_cols - num_low + 1) // 2
        keep[pad:pad + num_low] = 1.0
        # spacing adjusted for the fully sampled centre so that num_cols / acceleration columns survive on average
        step = (acceleration * (num_low - num_cols)) / (num_low * acceleration - num_cols)
        offset = self.rng.randint(0, round(step))
        keep[np.around(np.arange(offset, num_cols - 1, step)).astype(np.uint)] = 1.0
        return keep


def create_mask_for_mask_type(mask_type_str, center_fractions, accelerations):
    if mask_type_str == "random":
        return RandomMaskFunc(center_fractions, accelerations)
    elif mask_type_str == "equispaced":
        return EquispacedMaskFunc(center_fractions, accelerations)
    else:
        raise Exception(f"{mask_type_str} not supported")
