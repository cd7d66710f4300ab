"""Host float64 Hoeffding-Bentkus bound (C++ behind the C ABI) vs the reference's values (G8) and the
scipy-based oracle on a dense sweep.  CPU only (pure host code)."""
import numpy as np
import pytest

from conftest import load_golden


def test_hb_golden_g8():
    from im2im_uq_amd.core.calibration.bounds import HB_mu_plus
    for muhat, n, delta, expect in load_golden("g8_hb_bound")["rows"]:
        assert HB_mu_plus(muhat, int(n), delta) == pytest.approx(expect, abs=1e-9), (muhat, n, delta)
    assert HB_mu_plus(0.1, 10000, 0.1, 1000) == pytest.approx(0.10551758004098838, abs=1e-9)   # bounds.py:46
    assert HB_mu_plus(0.0, 50, 0.1) == 1.0                                                      # Q3 exception path


def test_hb_dense_sweep_vs_oracle():
    import warnings
    from oracle import calibration as oc
    from im2im_uq_amd.core.calibration.bounds import HB_mu_plus
    rng = np.random.RandomState(1)
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n in (2, 7, 64, 1000, 3474, 34742):
            for delta in (0.1, 0.05, 0.9):
                for muhat in list(rng.rand(12)) + [k / n for k in (0, 1, n // 2, n - 1, n)]:
                    a, b = HB_mu_plus(muhat, n, delta), oc.hb_mu_plus(muhat, n, delta)
                    worst = max(worst, abs(a - b))
                    assert a == pytest.approx(b, abs=1e-9), (muhat, n, delta)
    assert worst < 1e-9


def test_scan_quirks_on_host_table():
    """Q1-Q4 of the descending scan, driven from a hand-made table (no GPU needed)."""
    import torch
    from im2im_uq_amd.core.calibration.calibrate_model import scan_loss_table
    lambdas = torch.linspace(0, 6, 13)
    n = 200
    risk = torch.linspace(0.5, 0.0, 13)                   # column j: loss at lambdas[j]-dlambda
    table = risk.repeat(n, 1)
    lhat, out, trace = scan_loss_table(table, lambdas, alpha=0.1, delta=0.1)
    j_stop = trace[-1][0]
    assert float(lhat) == float(lambdas[j_stop])
    assert (out[:, :j_stop] == 0).all() and (out[:, j_stop:] == table[:, j_stop:]).all()       # Q2
    assert trace[0][0] == 12 and trace[0][2] == 1.0      # Rhat == 0 at the largest lambda -> RhatPlus = 1.0 (Q3)
    assert j_stop == 12                                   # ... so the scan stops immediately (Q3)
    table2 = torch.full((n, 13), 0.01)
    lhat2, out2, trace2 = scan_loss_table(table2, lambdas, alpha=0.5, delta=0.5)
    assert len(trace2) == 13 and float(lhat2) == float(lambdas[-1] + (lambdas[1] - lambdas[0]) - 1e-9)  # Q4


@pytest.mark.parametrize("case", ["mid", "n130", "zero_risk", "no_stop", "c2"])
def test_rcps_scan_c_abi_on_reference_tables_g7(case):
    """im2im_rcps_scan through ctypes on the reference's own loss tables (fixtures G7, calibrate_model.py:130-144 run by the
    imported reference): same stop column, same lambda-hat bit for bit, same number of visited lambdas; Rhat within one fp32
    ulp of torch's mean (the C scan rounds the exact column mean once), RhatPlus accordingly."""
    import ctypes
    import torch
    from im2im_uq_amd._lib import lib
    g = load_golden("g7_calibrate_" + case)
    alpha, delta, L, lo, hi = float(g["cfg"][0]), float(g["cfg"][1]), int(g["cfg"][2]), float(g["cfg"][3]), float(g["cfg"][4])
    table = np.ascontiguousarray(g["table"], dtype=np.float32)            # [N][L] row-major, zero left of the stop (never visited)
    n = table.shape[0]
    lambdas = torch.linspace(lo, hi, L).numpy()
    stop, stopped, visited = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
    lhat = ctypes.c_float(0.0)
    rhat = np.zeros(L, np.float32)
    rplus = np.zeros(L, np.float64)
    rc = lib.im2im_rcps_scan(table.ctypes.data, n, L, L, 1, lambdas.ctypes.data, alpha, delta, 1000, ctypes.byref(stop),
                             ctypes.byref(stopped), ctypes.byref(lhat), ctypes.byref(visited), rhat.ctypes.data, rplus.ctypes.data, None)
    assert rc == 0
    ref = g["trace"]
    assert visited.value == len(ref) and stop.value == int(ref[-1][0])
    assert np.float32(lhat.value) == np.float32(g["lhat"])
    assert bool(stopped.value) == (case != "no_stop")
    for j, r, rp in ref:
        j = int(j)
        assert abs(float(rhat[j]) - r) <= 1.2e-7 * max(abs(r), 1e-30), (j, rhat[j], r)
        assert rplus[j] == pytest.approx(rp, abs=2e-6)
    # strided form: the transposed copy calibrate_model hands over
    cols = np.ascontiguousarray(table.T)
    stop2, lhat2 = ctypes.c_int32(-1), ctypes.c_float(0.0)
    assert lib.im2im_rcps_scan(cols.ctypes.data, n, L, 1, n, lambdas.ctypes.data, alpha, delta, 1000, ctypes.byref(stop2), None,
                               ctypes.byref(lhat2), None, None, None, None) == 0
    assert stop2.value == stop.value and lhat2.value == lhat.value


def test_rcps_scan_takes_torch_means_where_a_last_bit_decides():
    """the reference decides `Rhat >= alpha` on torch's fp32 `losses.mean()`; the C scan's correctly rounded mean may
    differ from it in the last bit.  hip_ops.rcps_scan re-decides every column whose outcome a few ulp could flip on torch's own
    mean of that row (im2im_rcps_scan's rhat_in), so the stop column is the reference loop's.  Column 2's mean is pushed across
    alpha by one ulp through rhat_in directly; the Python wrapper must agree with a plain torch re-statement of the loop."""
    import ctypes
    import torch
    from im2im_uq_amd import hip_ops
    from im2im_uq_amd._lib import lib
    torch.manual_seed(3)
    n, L = 997, 6
    lambdas = torch.linspace(0, 5, L)
    alpha, delta = 0.25, 0.9999                            # delta ~ 1: the HB bound hugs Rhat, so `Rhat >= alpha` is what decides
    cols = torch.rand(L, n) * 0.2                          # means ~0.1
    cols[2] = 0.25                                         # mean == alpha to the bit in exact arithmetic
    cols[2, ::2] += 3e-8; cols[2, 1::2] -= 3e-8            # ... and noise that only a summation order can see
    # the reference loop, restated with torch's means
    want = None
    for j in range(L - 1, -1, -1):
        r = cols[j].mean()
        rp = hip_ops.hb_mu_plus(r.item(), n, delta)
        if r >= alpha or rp > alpha:
            want = j
            break
    stop, stopped, lhat, trace = hip_ops.rcps_scan(cols, lambdas, alpha, delta)
    assert stopped and stop == want
    assert trace[-1][1] == float(cols[want].mean())        # the deciding Rhat is torch's, bit for bit
    # rhat_in is honoured by the C entry itself: a mean above alpha handed in for column 4 (visited before `want`) stops the scan
    # there and is reported back as that column's Rhat; NaN entries mean "compute it"
    assert want < 4
    given = np.full(L, np.nan, np.float32)
    given[4] = 0.9
    s_, st, lh = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_float(0)
    rh = np.zeros(L, np.float32)
    c = cols.numpy()
    assert lib.im2im_rcps_scan(c.ctypes.data, n, L, 1, n, lambdas.numpy().ctypes.data, alpha, delta, 1000, ctypes.byref(s_),
                               ctypes.byref(st), ctypes.byref(lh), None, rh.ctypes.data, None, given.ctypes.data) == 0
    assert s_.value == 4 and st.value == 1 and rh[4] == np.float32(0.9) and lh.value == float(lambdas[4])
    assert abs(float(rh[5]) - float(cols[5].mean())) <= 1.2e-7 * float(cols[5].mean())
