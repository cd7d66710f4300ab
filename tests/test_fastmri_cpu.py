"""Host half of the fastMRI input pipeline (no GPU): the mask functions against the reference's (fixture G18)."""
import numpy as np

from conftest import load_golden


def test_mask_functions_match_reference_bit_for_bit():
    from im2im_uq_amd.core.datasets.fastmri import subsample
    g = load_golden("g18_fastmri_pipeline")
    n = 0
    for k in g:
        if k.startswith("mask.") and k != "mask.two_rates":
            _, kind, cols, fname = k.split(".", 3)
            fn = subsample.create_mask_for_mask_type(kind, [0.08], [4])
            m = fn((1, int(cols), 2), tuple(map(ord, fname)))
            assert tuple(m.shape) == (1, int(cols), 1) and m.dtype.is_floating_point
            assert np.array_equal(m.reshape(-1).numpy().astype(np.uint8), g[k]), k
            n += 1
    assert n == 24
    two = subsample.EquispacedMaskFunc([0.08, 0.04], [4, 8])
    got = np.stack([two((1, 368, 2), (s,)).reshape(-1).numpy() for s in range(6)]).astype(np.uint8)
    assert np.array_equal(got, g["mask.two_rates"])
    # seeded calls leave the generator's own stream untouched (temp_seed restores the state)
    fn = subsample.EquispacedMaskFunc([0.08], [4])
    fn.rng.seed(5)
    a = fn((1, 368, 2)).numpy()
    fn.rng.seed(5)
    fn((1, 368, 2), seed=(1, 2, 3))
    b = fn((1, 368, 2)).numpy()
    assert np.array_equal(a, b)
