"""CPU oracle for the im2im-uq hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU fp32 ops + numpy/scipy float64)
of the reference algorithm for the one hot path this repo rebuilds:
quantile-regression UNet training + RCPS calibration.  Every function cites
the reference file:line it follows (paths relative to the reference root).

Rules (enforced by tests/test_layout.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here -- and only as the *checker* / timed baseline;
  * nothing under im2im_uq_amd/ imports it; the product path fails loudly when
    the HIP library is missing instead of falling back to this code.

Pinning: the reference holds no golden vectors of its own for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, generated in the build container by
tests/golden/make_golden.py (which imports /root/reference) and committed as
tests/golden/*.npz.  tests/test_oracle_golden.py checks every function here
against those fixtures.
"""
