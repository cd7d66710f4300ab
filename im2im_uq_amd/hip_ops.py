"""Thin tensor-level wrappers over the C ABI (include/im2im_uq.h).

Each wrapper validates device/dtype/contiguity, allocates outputs with torch (plumbing), and
launches the HIP kernel on torch's current stream.  No wrapper has a non-HIP fallback.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, dptr, lib, require_gpu, stream_ptr

F32 = torch.float32

# how a final layer's output planes become the nested set (include/im2im_uq.h IM2IM_SETS_*)
SETS_QUANTILE, SETS_SCALE, SETS_SQRT, SETS_SOFTMAX = 0, 1, 2, 3
_FORM_PLANES = {SETS_QUANTILE: 3, SETS_SCALE: 2, SETS_SQRT: 2, SETS_SOFTMAX: 3}


def _check_form(outputs, form):
    if form not in _FORM_PLANES:
        raise _lib.Im2ImError(f"unknown nested-set form {form!r}")
    if outputs.dim() < 3 or outputs.shape[1] != _FORM_PLANES[form]:
        raise _lib.Im2ImError(f"outputs must be [N,{_FORM_PLANES[form]},...] for form {form}, got {tuple(outputs.shape)}")


# ------------------------------------------------------------------ calibration (K11-K14)
def rcps_loss_table(outputs: torch.Tensor, labels: torch.Tensor, lam: torch.Tensor, want_counts: bool = False,
                    form: int = SETS_QUANTILE):
    """outputs [N,K,C,H,W] fp32 (K = 3 quantile planes, or 2 for the scale / sqrt forms), labels [N,C,H,W] fp32 (GPU);
    lam [L] fp32 ascending (any device).
    Returns table [N,L] fp32 on the GPU (and int32 counts if asked): one pass over HBM for all lambdas."""
    outputs = require_gpu(outputs, F32, "outputs")
    labels = require_gpu(labels, F32, "labels")
    _check_form(outputs, form)
    n = outputs.shape[0]
    p = outputs[0, 0].numel() if n else int(torch.tensor(outputs.shape[2:]).prod())
    if labels.numel() != n * p:
        raise _lib.Im2ImError(f"labels {tuple(labels.shape)} do not match outputs {tuple(outputs.shape)}")
    lam_cpu = lam.detach().to("cpu", F32).contiguous()
    if lam_cpu.dim() != 1 or lam_cpu.numel() < 1:
        raise _lib.Im2ImError("lam must be a non-empty 1-D grid")
    if lam_cpu.numel() > 1 and bool((lam_cpu[1:] < lam_cpu[:-1]).any()):
        raise _lib.Im2ImError("lam grid must be ascending (the one-pass scoring relies on monotone edges)")
    L = lam_cpu.numel()
    dev = outputs.device
    lam_dev = _grid_on_device(lam_cpu, dev)
    hist = torch.empty((max(1, lib.im2im_rcps_workspace_bytes(n, p, L) // 4),), dtype=torch.int32, device=dev)
    table = torch.empty((n, L), dtype=F32, device=dev)
    counts = torch.empty((n, L), dtype=torch.int32, device=dev) if want_counts else None
    rcps_loss_table_raw(outputs, labels, n, p, lam_dev, hist, table, counts, form)
    return (table, counts) if want_counts else table


_grid_cache = {}


def _grid_on_device(lam_cpu: torch.Tensor, dev) -> torch.Tensor:
    """the lambda grid is tiny and reused call after call: keep the device copy (keyed by its bytes)."""
    key = (lam_cpu.numpy().tobytes(), str(dev))
    t = _grid_cache.get(key)
    if t is None:
        if len(_grid_cache) > 64:
            _grid_cache.clear()
        t = _grid_cache[key] = lam_cpu.to(dev)
    return t


def rcps_loss_table_raw(outputs, labels, n, p, lam_dev, hist, table, counts=None, form=SETS_QUANTILE):
    """no allocation, no host traffic: histogram kernel + suffix kernel on the current stream."""
    dev = outputs.device
    with torch.cuda.device(dev):
        check(lib.im2im_rcps_loss_table(dptr(outputs), dptr(labels), n, p, dptr(lam_dev), lam_dev.numel(), int(form), dptr(hist),
                                        dptr(table), dptr(counts), stream_ptr(dev)), "im2im_rcps_loss_table")


def rcps_miscoverage(outputs: torch.Tensor, labels: torch.Tensor, lam: float, form: int = SETS_QUANTILE) -> torch.Tensor:
    """int32 [C, H*W] counts of (label > upper) + (label < lower) over images at one lambda."""
    outputs = require_gpu(outputs, F32, "outputs")
    labels = require_gpu(labels, F32, "labels")
    _check_form(outputs, form)
    n, three, c = outputs.shape[0], outputs.shape[1], outputs.shape[2]
    hw = 1
    for d in outputs.shape[3:]:
        hw *= int(d)
    out = torch.empty((c, hw), dtype=torch.int32, device=outputs.device)
    with torch.cuda.device(outputs.device):
        check(lib.im2im_rcps_miscoverage(dptr(outputs), dptr(labels), n, c, hw, float(lam), int(form), dptr(out),
                                         stream_ptr(outputs.device)), "im2im_rcps_miscoverage")
    return out


def nested_sets(output: torch.Tensor, lam: float, clamp_inplace: bool = True, form: int = SETS_QUANTILE, floor: bool = True):
    """output [N,K,...] fp32 GPU -> (lower_edge, prediction view, upper_edge), each [N,...].  floor: with the +-1e-6 floor
    of ModelWithUncertainty.nested_sets_from_output; without it, the final layer's own raw edges."""
    if not output.is_contiguous():
        raise _lib.Im2ImError("nested_sets: output must be contiguous (it is clamped in place like the reference)")
    require_gpu(output, F32, "output")
    _check_form(output, form)
    n = output.shape[0]
    p = output[0, 0].numel()
    lower = torch.empty(output.shape[:1] + output.shape[2:], dtype=F32, device=output.device)
    upper = torch.empty_like(lower)
    with torch.cuda.device(output.device):
        check(lib.im2im_nested_sets(dptr(output), n, p, float(lam), int(form), dptr(lower), dptr(upper), int(clamp_inplace),
                                    int(floor), stream_ptr(output.device)), "im2im_nested_sets")
    return lower, output[:, 1 if form in (SETS_QUANTILE, SETS_SOFTMAX) else 0], upper


def fraction_missed(lower: torch.Tensor, upper: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    lower = require_gpu(lower, F32, "lower")
    upper = require_gpu(upper, F32, "upper")
    label = require_gpu(label, F32, "label")
    n = lower.shape[0]
    p = lower[0].numel()
    if upper.numel() != n * p or label.numel() != n * p:
        raise _lib.Im2ImError("fraction_missed: shape mismatch")
    loss = torch.empty((n,), dtype=F32, device=lower.device)
    with torch.cuda.device(lower.device):
        check(lib.im2im_fraction_missed(dptr(lower), dptr(upper), dptr(label), n, p, dptr(loss), stream_ptr(lower.device)),
              "im2im_fraction_missed")
    return loss


def hb_mu_plus(muhat: float, n: int, delta: float, maxiters: int = 1000) -> float:
    return float(lib.im2im_hb_mu_plus(float(muhat), int(n), float(delta), int(maxiters)))


def set_option(key: str, value: int) -> None:
    """run-time A/B switch of a kernel variant (include/im2im_uq.h im2im_set_option)."""
    check(lib.im2im_set_option(key.encode(), int(value)), "im2im_set_option")


def rcps_scan(cols: torch.Tensor, lambdas: torch.Tensor, alpha: float, delta: float, maxiters: int = 1000, torch_means: bool = True):
    """the reference's descending lambda scan (calibrate_model.py:130-144) in the C library: `cols` [L, N] fp32 host tensor
    whose row j holds the N losses at lambdas[j] - dlambda.  Returns (stop_index, stopped, lhat, trace) with trace =
    [(j, Rhat, RhatPlus)] in visiting order.

    torch_means: the C scan's Rhat is the correctly rounded mean; the reference's is torch's fp32 `losses.mean()`, whose last
    bits depend on the host's vector width.  Every visited column whose decision a change of a few ulp could flip (Rhat next
    to alpha, n * Rhat next to an integer -- the floor inside HB_mu_plus --, RhatPlus next to alpha) gets torch's own mean of
    that row and the scan is repeated with it, until no such column is left: lambda-hat is then what the reference's loop
    yields on this host, at the cost of a handful of row means instead of one per visited lambda."""
    import ctypes
    import math
    cols = cols.to(torch.float32).contiguous().cpu()
    lam = lambdas.to(torch.float32).contiguous().cpu()
    L, n = cols.shape
    given = torch.full((L,), float("nan"), dtype=torch.float32)
    a32 = float(torch.tensor(alpha, dtype=torch.float32))
    while True:
        stop, stopped, visited = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        lhat = ctypes.c_float(0.0)
        rhat = torch.zeros((L,), dtype=torch.float32)
        rplus = torch.zeros((L,), dtype=torch.float64)
        check(lib.im2im_rcps_scan(cols.data_ptr(), int(n), int(L), 1, int(n), lam.data_ptr(), float(alpha), float(delta), int(maxiters),
                                  ctypes.byref(stop), ctypes.byref(stopped), ctypes.byref(lhat), ctypes.byref(visited), rhat.data_ptr(),
                                  rplus.data_ptr(), given.data_ptr() if torch_means else None), "im2im_rcps_scan")
        if not torch_means:
            break
        r = rhat[stop.value:].double()
        slack = 32.0 * 2.0 ** -24 * r.clamp_min(2.0 ** -100)                  # far more than any fp32 summation order moves a mean
        nr = r * n
        close = ((r - a32).abs() <= slack) | ((rplus[stop.value:] - alpha).abs() <= 1e-4 * max(alpha, 1e-12)) \
            | (torch.floor(nr - slack * n) != torch.floor(nr + slack * n))
        todo = [stop.value + int(i) for i in torch.nonzero(close).flatten() if math.isnan(float(given[stop.value + int(i)]))]
        if not todo:
            break
        for j in todo:
            given[j] = cols[j].mean()                                         # the reference's `losses.mean()` on this host
    trace = [(j, float(rhat[j]), float(rplus[j])) for j in range(L - 1, stop.value - 1, -1)]
    return stop.value, bool(stopped.value), lhat.value, trace


def hb_mu_plus_batch(muhat: torch.Tensor, n: int, delta: float, maxiters: int = 1000) -> torch.Tensor:
    """float64 [count] Hoeffding-Bentkus bounds of a vector of float32 empirical risks (host, multi-threaded)."""
    m = muhat.detach().to("cpu", F32).contiguous().reshape(-1)
    out = torch.empty((m.numel(),), dtype=torch.float64)
    check(lib.im2im_hb_mu_plus_batch(m.data_ptr(), m.numel(), int(n), float(delta), int(maxiters), out.data_ptr()),
          "im2im_hb_mu_plus_batch")
    return out
