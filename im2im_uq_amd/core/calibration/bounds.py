"""Drop-in for the reference's core/calibration/bounds.py (HB_mu_plus, :17-29).

The float64 Hoeffding-Bentkus solve runs in C++ behind the C ABI (im2im_hb_mu_plus,
csrc/hb_bound.cpp): regularised-incomplete-beta binomial CDF + Brent's method, no scipy.
"""
from .. import _pkg  # noqa: F401  (fails loudly if the HIP library is missing)
from ... import hip_ops


def HB_mu_plus(muhat, n, delta, maxiters=1000):
    """Same signature and return convention as the reference (1 / root / 1.0 on solver failure)."""
    return hip_ops.hb_mu_plus(float(muhat), int(n), float(delta), int(maxiters))
