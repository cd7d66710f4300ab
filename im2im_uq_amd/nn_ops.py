"""autograd.Function layer over the HIP training kernels (C ABI in include/im2im_uq.h).

Internal activation convention: tensors are *logically* NCHW ([B,C,H,W], so user code and
state_dicts see the reference's shapes) but live in channels-last memory in the compute dtype
(bf16 by default, fp32 for tight parity); `nhwc()` / `nchw()` are zero-copy views between the two.
All arithmetic on activations happens in the HIP kernels; torch is used for allocation and autograd
bookkeeping only.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch
from torch.utils.weak import WeakTensorKeyDictionary

from . import _lib
from ._lib import check, dptr, lib, stream_ptr

F32 = torch.float32
BF16 = torch.bfloat16
_DT = {F32: 0, BF16: 1}

_mode = os.environ.get("IM2IM_COMPUTE_DTYPE", "bf16").lower()
_default_dtype = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "float32": F32, "fp8": BF16}[_mode]
_fp8_forward = _mode == "fp8"


def set_compute_dtype(dt) -> None:
    """'bf16' (throughput: bf16 storage + bf16 MFMA, fp32 accumulate), 'fp32' (exact-fp32 MFMA, parity) or 'fp8' (bf16
    storage; the FORWARD 3x3 convolutions with >= 64 channels take e4m3 operands on the block-scaled MFMA at ~2x the bf16
    rate, everything else -- first conv, 1x1 / head convs, the whole backward pass -- as in 'bf16')."""
    global _default_dtype, _fp8_forward
    _default_dtype = {"bf16": BF16, "fp32": F32, "fp8": BF16, BF16: BF16, F32: F32}[dt]
    _fp8_forward = dt == "fp8"


def get_compute_dtype():
    return _default_dtype


def fp8_forward() -> bool:
    return _fp8_forward


def compute_mode() -> str:
    return "fp8" if _fp8_forward else ("bf16" if _default_dtype == BF16 else "fp32")


def nhwc(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """[B,C,H,W] logical -> contiguous [B,H,W,C] (view when x is channels-last in `dtype`)."""
    y = x.permute(0, 2, 3, 1)
    if dtype is not None and y.dtype != dtype:
        y = y.to(dtype)
    return y if y.is_contiguous() else y.contiguous()


def nchw(y: torch.Tensor) -> torch.Tensor:
    return y.permute(0, 3, 1, 2)


class _Scratch:
    """one growing byte buffer per device; every kernel runs on the same stream, so consecutive users
    of the scratch are ordered.

    A captured HIP graph holds the ADDRESSES of the buffers its kernels used.  Once a graph exists (`pinned`), a buffer that
    has to grow is retired, not freed: the graph may go on writing to it, so its memory must not be handed to anyone else."""
    bufs = {}
    pinned = False
    retired = []
    graphs = 0                 # live captured graphs (GraphedStep): `pinned` while > 0

    @classmethod
    def pin(cls):
        cls.graphs += 1
        cls.pinned = True

    @classmethod
    def unpin(cls):
        """a captured graph was dropped: when it was the last one, nothing can write to the retired buffers any more"""
        cls.graphs = max(cls.graphs - 1, 0)
        if cls.graphs == 0:
            cls.pinned = False
            if cls.retired:
                torch.cuda.synchronize()       # replays still in flight
                cls.retired.clear()

    @classmethod
    def get(cls, nbytes: int, device, key: str = "a") -> torch.Tensor:
        key = (torch.device(device).index, key)
        buf = cls.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None and cls.pinned:
                cls.retired.append(buf)
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            cls.bufs[key] = buf
        return buf

    @classmethod
    def addresses(cls):
        return tuple(sorted((k, b.data_ptr()) for k, b in cls.bufs.items()))

    ctrs = {}

    @classmethod
    def counters(cls, device, key: str = "a") -> torch.Tensor:
        """the zero-initialised ticket counters of the one-launch BatchNorm reductions (include/im2im_uq.h IM2IM_BN_COUNTERS; the
        kernels leave them zero): one array per (device, scratch key) = per stream that uses that scratch, allocated once and never
        moved (a captured HIP graph holds its address)."""
        k = (torch.device(device).index, key)
        c = cls.ctrs.get(k)
        if c is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.Im2ImError("BatchNorm ticket counters would be created inside a graph capture; run one eager step first")
            c = cls.ctrs[k] = torch.zeros(64, dtype=torch.int32, device=device)
        return c


class KernelTimer:
    """optional per-launch timing of the MFMA conv kernels with HIP events recorded on the launch stream
    (bench.py's roofline leg).  Usage: nn_ops.TIMER = KernelTimer(); ...; rows = nn_ops.TIMER.collect()."""

    def __init__(self):
        self.records = []

    def wrap(self, name, flops, device):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(device))
        self.records.append((name, flops, e0, e1))
        return e1

    def collect(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            n, f, t = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, f + flops, t + ms)
        return out


TIMER = None


def _tile_name(kind, h, w_, co, taps, dtype):
    small = h < 64 or w_ < 64
    bn = 128 if co % 128 == 0 else 64 if co % 64 == 0 else 32
    if kind == "wgrad":
        return f"conv_wgrad_kernel<{'bf16' if dtype == BF16 else 'f32'},8,16,{taps}>"
    tb, th, tw = (1, 16, 16) if not small else (4, 8, 8)
    return f"conv_igemm_kernel<{'bf16' if dtype == BF16 else 'f32'},{tb}x{th}x{tw},{bn},taps={taps}>"


def _gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.Im2ImError(f"{name} must live on the GPU; the HIP path has no CPU fallback (got {t.device})")


# ----------------------------------------------------------------------------------------- raw ops
def pack_weight(w: torch.Tensor, dtype, want_wd=True):
    """w [Co,Ci,kh,kw] fp32 -> (wf [Co,taps,Ci], wd [Ci,taps(reversed),Co]) in `dtype`."""
    co, ci = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3]
    w = w.detach()
    if w.dtype != F32 or not w.is_contiguous():
        w = w.to(F32).contiguous()
    wf = torch.empty((co, taps, ci), dtype=dtype, device=w.device)
    wd = torch.empty((ci, taps, co), dtype=dtype, device=w.device) if want_wd else None
    check(lib.im2im_pack_conv_weight(dptr(w), co, ci, taps, _DT[dtype], dptr(wf), dptr(wd), stream_ptr(w.device)),
          "im2im_pack_conv_weight")
    return wf, wd


# ---- all conv weights of a model packed by ONE launch per training step -----------------------------------------------
# A conv's forward asks for its packed operands (wf, wd).  The first request after the weights changed (the optimizer bumps
# their version counters) packs EVERY registered weight that is stale in one multi-tensor launch; the other layers of the
# step then find theirs ready.  Weights register themselves at their first use.
# Staleness is detected by (storage address, version counter).  In-place writes through `p.data` (legacy optimizers written as
# `p.data.add_`, EMA swaps as `p.data.copy_`, `dist.broadcast(p.data)`) do NOT bump the counter: call invalidate_packed()
# (or touched(p)) after such a write, write through `p.detach()` / under torch.no_grad() instead, or set IM2IM_BATCH_PACK=0 to
# re-pack on every forward.  The package's own writers: FusedAdam and load_state_dict bump it; broadcast_module_state cannot (c10d
# collectives do not touch the counter whichever alias they are given) and calls invalidate_packed() after its broadcasts.
BATCH_WEIGHT_PACKING = os.environ.get("IM2IM_BATCH_PACK", "1") != "0"
_pack_registry = {}        # id(weight) -> weakref(weight)
_pack_cache = {}           # (id(weight), dtype) -> (data_ptr, version, wf, wd)


def invalidate_packed(*weights) -> None:
    """forget the packed conv operands (training cache and the eval-mode per-module caches) of `weights` -- of every weight when
    called without arguments.  For code that changed parameters behind autograd's back (`p.data.copy_`, a raw-pointer write)."""
    if not weights:
        _pack_cache.clear()
        _EVAL_CACHE.clear()
        _HEADS_CACHE.clear()
        _fp8_pack_cache.clear()
        return
    ids = {id(w) for w in weights}
    for k in [k for k in _pack_cache if k[0] in ids]:
        _pack_cache.pop(k, None)
    for i in ids:
        _fp8_pack_cache.pop(i, None)
    for w in weights:
        touched(w)                                           # the eval caches key on the version counter


def packed_pair(weight: torch.Tensor, dtype):
    """(wf, wd) of a 4-d conv weight Parameter in `dtype` through the per-step batched packing (falls back to pack_weight for
    tensors that are not long-lived fp32 leaves)."""
    if (not BATCH_WEIGHT_PACKING or not isinstance(weight, torch.nn.Parameter) or weight.dtype != F32 or not weight.is_contiguous()
            or not weight.is_cuda):
        return pack_weight(weight, dtype)
    wid = id(weight)
    hit = _pack_cache.get((wid, dtype))
    if hit is not None and hit[0] == weight.data_ptr() and hit[1] == weight._version:
        return hit[2], hit[3]
    if wid not in _pack_registry:
        _pack_registry[wid] = weakref.ref(weight, lambda _r, wid=wid: (_pack_registry.pop(wid, None),
                                                                        [_pack_cache.pop(k, None) for k in list(_pack_cache) if k[0] == wid]))
    stale = []
    for oid, ref in list(_pack_registry.items()):
        w = ref()
        if w is None or w.device != weight.device or w.dtype != F32 or not w.is_contiguous():
            continue
        h = _pack_cache.get((oid, dtype))
        if h is None and oid != wid:
            continue                                         # never asked for in this dtype: not part of this model's step
        if h is None or h[0] != w.data_ptr() or h[1] != w._version:
            stale.append((oid, w))
    n = len(stale)
    outs = []
    for _, w in stale:
        co, ci, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        outs.append((torch.empty((co, taps, ci), dtype=dtype, device=w.device), torch.empty((ci, taps, co), dtype=dtype, device=w.device)))
    arr, i32 = ctypes.c_void_p * n, ctypes.c_int32 * n
    check(lib.im2im_pack_conv_weights_multi(n, arr(*[w.data_ptr() for _, w in stale]), i32(*[w.shape[0] for _, w in stale]),
                                            i32(*[w.shape[1] for _, w in stale]), i32(*[w.shape[2] * w.shape[3] for _, w in stale]),
                                            _DT[dtype], arr(*[o[0].data_ptr() for o in outs]), arr(*[o[1].data_ptr() for o in outs]),
                                            stream_ptr(weight.device)), "im2im_pack_conv_weights_multi")
    for (oid, w), (wf, wd) in zip(stale, outs):
        _pack_cache[(oid, dtype)] = (w.data_ptr(), w._version, wf, wd)
    hit = _pack_cache[(wid, dtype)]
    return hit[2], hit[3]


def pack_weight_fp8(w: torch.Tensor):
    """w [Co,Ci,3,3] fp32 -> (wq uint8 [Co,9,Ci] e4m3 bytes -- an opaque operand, logical order via fp8_pack_logical --,
    wscale [Co] fp32 power-of-two scales)."""
    co, ci = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3]
    w = w.detach()
    if w.dtype != F32 or not w.is_contiguous():
        w = w.to(F32).contiguous()
    wq = torch.empty((co, taps, ci), dtype=torch.uint8, device=w.device)
    wscale = torch.empty((co,), dtype=F32, device=w.device)
    check(lib.im2im_pack_conv_weight_fp8(dptr(w), co, ci, taps, dptr(wq), dptr(wscale), stream_ptr(w.device)),
          "im2im_pack_conv_weight_fp8")
    return wq, wscale


def fp8_pack_logical(wq: torch.Tensor) -> torch.Tensor:
    """the packed fp8 weight bytes in logical order [N, 9, K].  The packed buffer is an opaque operand: for N % 32 == 0 and
    K % 64 == 0 its storage order is fragment-major (csrc/conv_fp8.hip wfrag8_index: [row block of 32][tap][64-channel chunk]
    [16-byte part][lane = (k / 32 % 2) * 32 + row % 32][16]); this undoes it (tests, tools)."""
    n, taps, k = wq.shape
    if taps != 9 or n % 32 or k % 64:
        return wq
    f = wq.reshape(n // 32, 9, k // 64, 2, 2, 32, 16)          # [row block][tap][chunk][part][k half][row in block][byte]
    return f.permute(0, 5, 1, 2, 4, 3, 6).reshape(n, 9, k)     # -> [row block, row][tap][chunk, half, part, byte]


def pack_weight_fp8_dgrad(w: torch.Tensor):
    """w [Co,Ci,3,3] fp32 -> (wq_d uint8 [Ci,9 reversed,Co] e4m3 bytes, wscale_d [Ci] fp32): operand of the fp8 data-gradient."""
    co, ci = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3]
    w = w.detach()
    if w.dtype != F32 or not w.is_contiguous():
        w = w.to(F32).contiguous()
    wq = torch.empty((ci, taps, co), dtype=torch.uint8, device=w.device)
    wscale = torch.empty((ci,), dtype=F32, device=w.device)
    check(lib.im2im_pack_conv_weight_fp8_dgrad(dptr(w), co, ci, taps, dptr(wq), dptr(wscale), stream_ptr(w.device)),
          "im2im_pack_conv_weight_fp8_dgrad")
    return wq, wscale


# ---- fp8 mode: every 3x3 weight's fp8 operands packed by one launch per kind and training step [r4] ---------------------------------
FP8_BATCH_PACK = os.environ.get("IM2IM_FP8_BATCH_PACK", "1") != "0"
_fp8_pack_registry = {}    # id(weight) -> [weakref(weight), wants the data-gradient operand]
_fp8_pack_cache = {}       # id(weight) -> (data_ptr, version, (wq, wscale), (wq_d, wscale_d) | None)


def packed_fp8(weight: torch.Tensor, want_dgrad: bool):
    """((wq, wscale), (wq_d, wscale_d) | None): the e4m3 forward operand of a [Co,Ci,3,3] fp32 weight and, if asked for, its
    data-gradient operand -- through a per-step batch like packed_pair: the first request after the optimizer changed the weights
    packs EVERY registered fp8 weight in one launch per kind (31 per-layer launches were 0.46 ms of a 20 ms fp8-mode step)."""
    if (not (BATCH_WEIGHT_PACKING and FP8_BATCH_PACK) or not isinstance(weight, torch.nn.Parameter) or weight.dtype != F32
            or not weight.is_contiguous() or not weight.is_cuda or weight.shape[2] * weight.shape[3] != 9):
        return pack_weight_fp8(weight), (pack_weight_fp8_dgrad(weight) if want_dgrad else None)
    wid = id(weight)
    ent = _fp8_pack_registry.get(wid)
    if ent is None:
        ent = _fp8_pack_registry[wid] = [weakref.ref(weight, lambda _r, wid=wid: (_fp8_pack_registry.pop(wid, None), _fp8_pack_cache.pop(wid, None))),
                                         False]
    newly_dgrad = want_dgrad and not ent[1]
    ent[1] = ent[1] or want_dgrad
    hit = _fp8_pack_cache.get(wid)
    if hit is not None and hit[0] == weight.data_ptr() and hit[1] == weight._version and not newly_dgrad:
        return hit[2], hit[3]
    stale = []
    for oid, (ref, dg) in list(_fp8_pack_registry.items()):
        w = ref()
        if w is None or w.device != weight.device or not w.is_contiguous():
            continue
        h = _fp8_pack_cache.get(oid)
        if h is None or h[0] != w.data_ptr() or h[1] != w._version or (dg and h[3] is None):
            stale.append((oid, w, dg))
    dev = weight.device
    fwd = [(torch.empty((w.shape[0], 9, w.shape[1]), dtype=torch.uint8, device=dev), torch.empty((w.shape[0],), dtype=F32, device=dev)) for _, w, _ in stale]
    bwd = [(torch.empty((w.shape[1], 9, w.shape[0]), dtype=torch.uint8, device=dev), torch.empty((w.shape[1],), dtype=F32, device=dev)) if dg else None
           for _, w, dg in stale]
    for kind, outs in ((0, fwd), (1, bwd)):
        idx = [i for i, o in enumerate(outs) if o is not None]
        if not idx:
            continue
        n = len(idx)
        arr, i32 = ctypes.c_void_p * n, ctypes.c_int32 * n
        check(lib.im2im_pack_conv_weights_fp8_multi(n, arr(*[stale[i][1].data_ptr() for i in idx]), i32(*[stale[i][1].shape[0] for i in idx]),
                                                    i32(*[stale[i][1].shape[1] for i in idx]), arr(*[outs[i][0].data_ptr() for i in idx]),
                                                    arr(*[outs[i][1].data_ptr() for i in idx]), kind, stream_ptr(dev)),
              "im2im_pack_conv_weights_fp8_multi")
    for (oid, w, _), f, b_ in zip(stale, fwd, bwd):
        _fp8_pack_cache[oid] = (w.data_ptr(), w._version, f, b_)
    hit = _fp8_pack_cache[wid]
    return hit[2], hit[3]


FP8_DGRAD = os.environ.get("IM2IM_FP8_DGRAD", "1") != "0"     # fp8 mode: data-gradients on the fp8 kernel too (e5m2 operand)
# [r4] ... and the weight gradients of those layers (e5m2 dz x e4m3 input, conv_wgrad_fp8_roll_kernel): "auto" = where the fp8 kernel is
# the faster one, i.e. the layers with 64 output channels (x1.12-1.17 over the bf16 roll kernel; on the 128-channel-wide layers the two
# are level -- its 96 accumulators + staged tile + conversions do not fit 168 registers -- and at 40x40 the bf16 kernel leads by 5 %:
# profiles/r04_ab_experiments.txt section 20); "1" = every eligible layer, "0" = none
FP8_WGRAD = os.environ.get("IM2IM_FP8_WGRAD", "auto")
FP8_WGRAD = {"0": False, "1": True}.get(FP8_WGRAD, "auto")


_fp8_grad_scales = WeakTensorKeyDictionary()        # conv weight -> Fp8GradScale; NOT an attribute of the Parameter: Parameter.__reduce_ex__
                                                    # pickles extra attributes, and the whole-module checkpoint (train.py, reference :191)
                                                    # must not embed a CUDA tensor or depend on this class to load


class Fp8GradScale:
    """delayed-scaling state of ONE gradient tensor (the dz a conv's data-gradient consumes): three device floats rotated
    step by step -- max |dz| of the previous step (sets this step's scale), this step's accumulator, the next one's (zeroed
    by this step's kernel).  Seeded from the tensor itself the first time."""
    __slots__ = ("amax", "step", "__weakref__")

    def __init__(self):
        self.amax, self.step = None, 0

    def slots(self, dz):
        if self.amax is None:
            self.amax = torch.zeros(3, dtype=F32, device=dz.device)
            self.amax[2] = dz.detach().abs().max().to(F32)           # plays "previous step" for step 0
        s = self.step
        self.step += 1
        base, prev, now, nxt = self.amax.data_ptr(), (s + 2) % 3, s % 3, (s + 1) % 3
        return base + 4 * prev, base + 4 * now, base + 4 * nxt


def conv_wgrad_fp8(x, dz, amax_prev, x_ss=None, x_hi=None, x_ss_hi=None, scratch_key="a", out=None, defer=False):
    """dw [Co,Ci,9] fp32 of a 3x3 conv with e5m2 dz (scaled from the device scalar at `amax_prev`, an address) and e4m3 input
    (csrc/conv_wgrad.hip conv_wgrad_fp8_kernel); arguments as conv_wgrad."""
    b, h, w_, ci = x.shape
    ci_lo = ci
    if x_hi is not None:
        ci = 2 * ci
    co = dz.shape[3]
    nbytes = lib.im2im_conv_wgrad_workspace_bytes(b, h, w_, ci, co, 9)
    ws = _Scratch.get(nbytes, x.device, scratch_key)
    dw = torch.empty((co, ci, 9), dtype=F32, device=x.device) if out is None else out
    width = _wgrad_width(x.device)
    ev = TIMER.wrap(f"conv_wgrad_fp8_kernel<{128 if co % 128 == 0 else 64}>", 2.0 * b * h * w_ * co * ci * 9, x.device) if TIMER else None
    nsplit = ctypes.c_int32(0)
    check(lib.im2im_conv_wgrad_fp8(dptr(x), dptr(x_ss), dptr(x_hi), dptr(x_ss_hi), ci_lo, dptr(dz), amax_prev, None if defer else dptr(dw),
                                   dptr(ws), ws.numel(), b, h, w_, ci, co, width, ctypes.addressof(nsplit), stream_ptr(x.device)),
          "im2im_conv_wgrad_fp8")
    if defer:
        _pending_reduce.setdefault(x.device.index, []).append((ws, nsplit.value, co, ci, 9, dw))
    if ev is not None:
        ev.record(torch.cuda.current_stream(x.device))
    return dw


def conv_dgrad_fp8(dz, wq_d, wscale_d, state: Fp8GradScale, split_out=0, slots=None):
    """dx (bf16) = data-gradient of a 3x3 pad-1 conv with e5m2 dz / e4m3 weights (csrc/conv_fp8.hip, GRAD form).
    slots: this step's (previous, now, next) amax addresses when the caller already took them (the fp8 weight gradient of the same
    layer reads `previous` too)."""
    b, h, w_, cz = dz.shape
    cx = wq_d.shape[0]
    if split_out:
        dx = torch.empty((b, h, w_, split_out), dtype=BF16, device=dz.device)
        dx_hi = torch.empty((b, h, w_, cx - split_out), dtype=BF16, device=dz.device)
    else:
        dx, dx_hi = torch.empty((b, h, w_, cx), dtype=BF16, device=dz.device), None
    prev, now, nxt = slots if slots is not None else state.slots(dz)
    small = h < 64 or w_ < 64
    name = f"conv_fp8_kernel<{'2x8x8' if small else '1x16x16'},{128 if cx % 128 == 0 else 64},dgrad>"
    ev = TIMER.wrap(name, 2.0 * b * h * w_ * cz * cx * 9, dz.device) if TIMER else None
    check(lib.im2im_conv_dgrad_fp8(dptr(dz), dptr(wq_d), dptr(wscale_d), dptr(dx), dptr(dx_hi), int(split_out), prev, now, nxt,
                                   b, h, w_, cz, cx, stream_ptr(dz.device)), "im2im_conv_dgrad_fp8")
    if ev is not None:
        ev.record(torch.cuda.current_stream(dz.device))
    return (dx, dx_hi) if split_out else dx


def fp8_eligible(ci, co, x_dtype, ci_lo=None) -> bool:
    return _fp8_forward and x_dtype == BF16 and ci % 64 == 0 and co % 64 == 0 and (ci_lo is None or ci_lo % 64 == 0)


def conv_fwd_fp8(x, wq, wscale, bias=None, scale_shift=None, relu=False, want_stats=False, in_ss=None, x_hi=None, in_ss_hi=None):
    """3x3 pad-1 conv with e4m3 operands (see csrc/conv_fp8.hip): x [B,H,W,Ci] bf16 -> y [B,H,W,Co] bf16 (+ stats)."""
    b, h, w_, ci = x.shape
    ci_lo = ci
    if x_hi is not None:
        ci = 2 * ci
    co = wq.shape[0]
    y = torch.empty((b, h, w_, co), dtype=BF16, device=x.device)
    stats = torch.empty((lib.im2im_conv_fp8_stats_rows(b, h, w_), 3, co), dtype=F32, device=x.device) if want_stats else None
    sc = sh = None
    if scale_shift is not None:
        sc, sh = scale_shift[0], scale_shift[1]
    small = h < 64 or w_ < 64
    name = f"conv_fp8_kernel<{'2x8x8' if small else '1x16x16'},{128 if co % 128 == 0 else 64}>"
    ev = TIMER.wrap(name, 2.0 * b * h * w_ * co * ci * 9, x.device) if TIMER else None
    check(lib.im2im_conv_fwd_fp8(dptr(x), dptr(in_ss), dptr(x_hi), dptr(in_ss_hi), ci_lo, dptr(wq), dptr(wscale), dptr(bias), dptr(sc),
                                 dptr(sh), dptr(y), dptr(stats), b, h, w_, ci, co, int(relu), stream_ptr(x.device)), "im2im_conv_fwd_fp8")
    if ev is not None:
        ev.record(torch.cuda.current_stream(x.device))
    return (y, stats) if want_stats else y


if os.environ.get("IM2IM_BN_FUSED_SMALL") is not None:   # A/B: one-launch BatchNorm sums for <= 256 partial rows (default on)
    check(lib.im2im_set_option(b"bn_fused_small", int(os.environ["IM2IM_BN_FUSED_SMALL"])), "im2im_set_option")
if os.environ.get("IM2IM_BN_ONELAUNCH") is not None:     # A/B: 1 = BatchNorm statistics / backward sums of many partial rows in ONE launch (default 0: two, measured faster)
    check(lib.im2im_set_option(b"bn_onelaunch", int(os.environ["IM2IM_BN_ONELAUNCH"])), "im2im_set_option")
if os.environ.get("IM2IM_BN_APPLY_KEEP_MB") is not None:   # A/B: dz tensors up to n MB written with cacheable stores (default 0: all streamed)
    check(lib.im2im_set_option(b"bn_apply_keep_mb", int(os.environ["IM2IM_BN_APPLY_KEEP_MB"])), "im2im_set_option")
if os.environ.get("IM2IM_POOL_BWD_BLOCKS") is not None:   # A/B: workgroups of bn_relu_pool_bwd (default 6144; 2048 until round 5)
    check(lib.im2im_set_option(b"pool_bwd_blocks", int(os.environ["IM2IM_POOL_BWD_BLOCKS"])), "im2im_set_option")
if os.environ.get("IM2IM_POOL_BWD_FULL") is not None:     # A/B: 0 = the branching form of bn_relu_pool_bwd also for even extents
    check(lib.im2im_set_option(b"pool_bwd_full", int(os.environ["IM2IM_POOL_BWD_FULL"])), "im2im_set_option")
if os.environ.get("IM2IM_CONV_ROLL") is not None:         # A/B: 0 = conv_igemm_kernel also for the 64-output-channel full-resolution layers
    check(lib.im2im_set_option(b"conv_roll", int(os.environ["IM2IM_CONV_ROLL"])), "im2im_set_option")
if os.environ.get("IM2IM_CONV_SPLITK") is not None:       # A/B of the split-K target (see im2im_set_option): 0 = off
    check(lib.im2im_set_option(b"conv_splitk", int(os.environ["IM2IM_CONV_SPLITK"])), "im2im_set_option")

BF16_CENTERING = False    # opt-in: bf16 train mode stores z - running_mean (im2im_conv_fwd `center`); measured gain on the
                          # train-forward rounding error is modest (5.8 % -> 4.8 %), so the default keeps the plain storage
                          # whose rounding points the bf16-emulating oracle reproduces


def conv_fwd(x, wf, bias=None, scale_shift=None, relu=False, want_stats=False, in_ss=None, center=None, x_hi=None,
             in_ss_hi=None, split_out=0, out=None, out_hi=None):
    """x [B,H,W,Ci], wf [Co,taps,Ci] -> y [B,H,W,Co] (+ stats [R,2,Co]).  in_ss [2,Ci]: x is a producer's pre-BN z and
    the kernel applies max(z*scale+shift, 0) while staging it (lazy BatchNorm+ReLU).
    x_hi: second half of the input channels (the concatenation [x, x_hi] is never materialised); split_out = Co_lo > 0:
    the output is returned as two tensors (channels [0,Co_lo) and [Co_lo,Co))."""
    b, h, w_, ci = x.shape
    ci_lo = ci
    if x_hi is not None:
        if x_hi.shape != x.shape or x_hi.dtype != x.dtype:
            raise _lib.Im2ImError(f"conv_fwd: split input halves must match, got {tuple(x.shape)} and {tuple(x_hi.shape)}")
        ci = 2 * ci
    co, taps = wf.shape[0], wf.shape[1]
    y_hi = None
    if out is not None:                               # preallocated result(s): a batch slice of a larger tensor
        y, y_hi = out, out_hi
    elif split_out:
        y = torch.empty((b, h, w_, split_out), dtype=x.dtype, device=x.device)
        y_hi = torch.empty((b, h, w_, co - split_out), dtype=x.dtype, device=x.device)
    else:
        y = torch.empty((b, h, w_, co), dtype=x.dtype, device=x.device)
    stats = None
    if want_stats:
        rows = lib.im2im_conv_stats_rows(b, h, w_, co)
        stats = torch.empty((rows, 3, co), dtype=F32, device=x.device)     # per-tile (mean, M2, count)
    sc = sh = None
    if scale_shift is not None:
        sc, sh = scale_shift[0], scale_shift[1]
    ev = TIMER.wrap(_tile_name("igemm", h, w_, co, taps, x.dtype), 2.0 * b * h * w_ * co * ci * taps, x.device) if TIMER else None
    # under-filled launches (a strong-scaled job's ~10 images per GPU at the 40x40 / 20x20 levels) split their reduction
    # through an fp32 workspace; 0 bytes = this shape is never split
    ws, ws_bytes = None, 0
    if sc is None:
        ws_bytes = lib.im2im_conv_splitk_workspace_bytes(b, h, w_, ci, co, taps)
        if ws_bytes > 0:
            ws = _Scratch.get(ws_bytes, x.device, "splitk")
    check(lib.im2im_conv_fwd_split_ws(dptr(x), dptr(in_ss), dptr(x_hi), dptr(in_ss_hi), ci_lo, dptr(wf), dptr(bias), dptr(center),
                                      dptr(sc), dptr(sh), dptr(y), dptr(y_hi), int(split_out), dptr(stats), b, h, w_, ci, co,
                                      taps, int(relu), _DT[x.dtype], dptr(ws), ws.numel() if ws is not None else 0,
                                      stream_ptr(x.device)), "im2im_conv_fwd_split_ws")
    if ev is not None:
        ev.record(torch.cuda.current_stream(x.device))
    if split_out:
        return y, y_hi
    return (y, stats) if want_stats else y


def conv_dgrad_bn(dz, wd, bn_z, bn_ss, bn_mi):
    """data-gradient dx = da of a lazy BatchNorm+ReLU activation, plus the per-tile partial sums of the BatchNorm backward
    (sum g, sum g*xhat) accumulated by the same epilogue -> (dx [B,H,W,Co], partial [R,2,Co])."""
    b, h, w_, ci = dz.shape
    co, taps = wd.shape[0], wd.shape[1]
    dx = torch.empty((b, h, w_, co), dtype=dz.dtype, device=dz.device)
    partial = torch.empty((lib.im2im_conv_stats_rows(b, h, w_, co), 2, co), dtype=F32, device=dz.device)
    ev = TIMER.wrap(_tile_name("igemm", h, w_, co, taps, dz.dtype), 2.0 * b * h * w_ * co * ci * taps, dz.device) if TIMER else None
    check(lib.im2im_conv_dgrad_bn(dptr(dz), dptr(wd), dptr(dx), dptr(bn_z), dptr(bn_ss), dptr(bn_mi), dptr(partial), b, h, w_, ci, co,
                                  taps, _DT[dz.dtype], stream_ptr(dz.device)), "im2im_conv_dgrad_bn")
    if ev is not None:
        ev.record(torch.cuda.current_stream(dz.device))
    return dx, partial


def bn_relu_bwd_from_partial(da, z, scale_shift, mean_invstd, partial):
    c = z.shape[-1]
    m = z.numel() // c
    dev = z.device
    dz = torch.empty_like(z)
    dgamma = torch.empty((c,), dtype=F32, device=dev)
    dbeta = torch.empty((c,), dtype=F32, device=dev)
    ws = _Scratch.get(lib.im2im_reduce_workspace_bytes(2 * c) + 2 * c * 4, dev)
    check(lib.im2im_bn_relu_bwd_from_partial(dptr(da), dptr(z), dptr(scale_shift), dptr(mean_invstd), dptr(partial), partial.shape[0],
                                             dptr(dz), dptr(dgamma), dptr(dbeta), m, c, _DT[z.dtype], dptr(ws), ws.numel(),
                                             dptr(_Scratch.counters(dev)), stream_ptr(dev)), "im2im_bn_relu_bwd_from_partial")
    return dz, dgamma, dbeta


# ---- second stream for the weight gradients --------------------------------------------------------------------------
# In the backward pass the weight gradient of a conv (MFMA-bound, needs only dz and the saved input) is independent of the
# chain  data-gradient -> BatchNorm backward of the producer -> next layer ...  whose BatchNorm / pooling / upsampling
# kernels are HBM-bound.  Launched on a second HIP stream the two kinds of kernels share the chip: the side passes run in
# the memory system while the matrix cores work on dW.  Measured on MI355X (bench.py, batch 78): 46.0 -> 44.2 ms per step
# (+4 %); launching dW only after the data-gradient gave nothing (46.7 ms) -- profiles/README.md.  The streams meet again
# when the backward pass ends (an autograd engine callback) and, defensively, before the optimizer reads the gradients.
WGRAD_SIDE_STREAM = os.environ.get("IM2IM_WGRAD_STREAM", "1") != "0"
_side_streams = {}
_side_busy = set()
_side_keep = {}            # device index -> tensors the side stream still reads or writes; dropped once the main stream joined it
_callback_queued = set()


def side_stream(device):
    idx = torch.device(device).index
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=device)
    return st


def join_side_streams():
    """make the current stream of every device wait for the weight-gradient and BatchNorm-backward streams' work."""
    for idx in list(_side_busy):
        main = torch.cuda.current_stream(idx)
        if _pending_reduce.get(idx) and idx in _side_streams:
            with torch.cuda.stream(_side_streams[idx]):  # [r6] every deferred split-K reduction of this backward pass in one launch
                flush_wgrad_reduce(idx)
        for pool in (_side_streams, _bn_streams):
            if idx in pool:
                main.wait_stream(pool[idx])
        _side_keep.pop(idx, None)                        # from here on the main stream is ordered after every side-stream use
    _side_busy.clear()
    _callback_queued.clear()
    _halves.clear()
    _wgrad_calls[0] = 0


def _queue_join(idx):
    _side_busy.add(idx)
    if idx not in _callback_queued:                      # join when this backward pass ends
        _callback_queued.add(idx)
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)


# ---- backward pass pipelined over two halves of the batch ------------------------------------------------------------
# The backward chain  data-gradient (MFMA-bound) -> BatchNorm backward (two HBM-bound passes) -> data-gradient of the
# layer below ...  is strictly sequential per tensor, so the matrix cores idle during the BatchNorm passes and HBM idles
# during the data-gradients.  Cut along the batch it pipelines: the data-gradient writes images [0,b/2) and then
# [b/2,b); BatchNorm's reduction starts on the first half while the second is still being computed, and its second pass
# hands the first half of dz to the next data-gradient while finishing the second.  The BatchNorm kernels run on their
# own stream (the weight gradients have theirs), ordered by events; every launch uses the block decomposition of the
# unsplit call, so the results are bit-identical to the sequential schedule (tests/test_model_gpu.py).
# OPT-IN, measured negative on MI355X at the bench shape (profiles/r02_ab_experiments.txt): 43.05 ms/step sequential vs
# 43.6-44.1 pipelined (any size threshold), 44.6 vs 45.0 without the weight-gradient stream.  The data-gradients of the
# full-resolution 64-channel layers are themselves half HBM-bound and all MFMA kernels are power-limited, so a BatchNorm
# pass running beside them takes from them what it hides; the weight-gradient stream (+3.6 %) already fills the gaps.
BWD_PIPELINE = os.environ.get("IM2IM_BWD_PIPELINE", "0") == "1"
BWD_PIPELINE_MIN_BYTES = int(os.environ.get("IM2IM_BWD_PIPELINE_MIN_BYTES", str(64 << 20)))   # per tensor; below this the extra launches cost more
_bn_streams = {}
_halves = {}             # data_ptr -> Halves of a gradient tensor whose two batch halves become ready at different times


class Halves:
    """a gradient tensor [B,H,W,C] (NHWC storage) whose images [0,b_first) are complete at `first` and all of it at `rest`
    (an event, or None = in the order of the current stream).  Holds the tensor so its memory cannot be reused while
    the record exists."""
    __slots__ = ("tensor", "b_first", "first", "rest")

    def __init__(self, tensor, b_first, first, rest):
        self.tensor, self.b_first, self.first, self.rest = tensor, b_first, first, rest


def bn_stream(device):
    idx = torch.device(device).index
    st = _bn_streams.get(idx)
    if st is None:
        st = _bn_streams[idx] = torch.cuda.Stream(device=device)
    return st


def _publish_halves(t, b_first, first, rest=None):
    _halves[t.data_ptr()] = Halves(t, b_first, first, rest)


def _take_halves(t):
    h = _halves.pop(t.data_ptr(), None)
    if h is not None and (h.tensor.numel() != t.numel() or h.tensor.shape[0] != (t.shape[0])):
        raise _lib.Im2ImError("backward pipeline: a gradient tensor changed shape between producer and consumer")
    return h


def _pipeline_split(t):
    """images in the first half of NHWC tensor t, or 0 when the tensor is too small to be worth two launches."""
    if not BWD_PIPELINE or torch.is_grad_enabled() or t.shape[0] < 2 or t.numel() * t.element_size() < BWD_PIPELINE_MIN_BYTES:
        return 0
    return t.shape[0] // 2


def _on_side_stream(device, tensors, fn, after=None):
    """run fn() on the device's side stream after everything already queued on the current stream.

    Memory discipline: every tensor the side stream touches (inputs AND the outputs, which the caller allocates on the
    current stream before calling) is owned by the current stream's allocator pool and kept referenced until the join, so
    a block is only ever reused in current-stream order after the join.  Tensor.record_stream is deliberately not used:
    it parks a freed block until an event on the other stream has COMPLETED, and a host that runs several steps ahead of
    the GPU (no sync in a training loop) then never gets a block back -- the caching allocator grew to 190 GB for a
    12 GB working set, and a process starting while the driver was still releasing that memory ran 2x slower
    (profiles/r02_allocator_stall.txt)."""
    main = torch.cuda.current_stream(device)
    idx = torch.device(device).index
    _side_keep.setdefault(idx, []).extend(t for t in tensors if t is not None)
    _queue_join(idx)
    side = side_stream(device)
    side.wait_stream(main)
    if after is not None:                                # an operand produced on a third stream (pipelined BatchNorm backward)
        side.wait_event(after)
    with torch.cuda.stream(side):
        fn()


# [r5] width of a bf16 3x3 weight-gradient launch (split-K slabs = workgroups / channel blocks, one 768-thread workgroup per CU):
# ALONE on the chip it wants every CU (256 workgroups: 1,070 TF over the 17 layers against 720 at 128); issued from the side stream
# it shares the chip with the main stream's data-gradients and BatchNorm passes, and half the width is the faster STEP -- the other
# kernels keep half the CUs, the fp32 partial slabs (written by the kernel, read back by wgrad_reduce) halve: batch 78 39.07 ->
# 38.59 ms, per-GPU batch 10 6.68 -> 6.35 ms (64: 6.97, 96: 6.35, 160: 6.50, 192: 6.30 / 38.95; profiles/r05_ab_experiments.txt
# section 5).  IM2IM_WGRAD_WGS pins one width for both cases (A/B runs).
_WGS_PIN = os.environ.get("IM2IM_WGRAD_WGS")
WGRAD_WGS_ALONE = int(_WGS_PIN) if _WGS_PIN else 256
WGRAD_WGS_SIDE = int(_WGS_PIN) if _WGS_PIN else int(os.environ.get("IM2IM_WGRAD_WGS_SIDE", "128"))


WGRAD_TAIL_FULL_FROM = int(os.environ.get("IM2IM_WGRAD_TAIL_FULL_FROM", "0"))   # A/B: 3x3 weight gradients number >= this of a backward pass (1-based) run at the alone-width
_wgrad_calls = [0]


def _wgrad_width(device) -> int:
    """workgroups this weight-gradient launch aims at -- an ARGUMENT of the launch since ABI 3 (it was a process-global option two autograd
    threads could race on).  The split-K slab count follows from it and sets the order of the fp32 partial sums, so a layer's dW is
    bit-identical from step to step, but not between a launch from the side stream (128) and one from the main stream (256):
    IM2IM_WGRAD_STREAM=0 / 1 agree to fp32 re-association noise, not bits."""
    side = _side_streams.get(torch.device(device).index)
    _wgrad_calls[0] += 1
    tail = WGRAD_TAIL_FULL_FROM > 0 and _wgrad_calls[0] >= WGRAD_TAIL_FULL_FROM
    return WGRAD_WGS_SIDE if (side is not None and torch.cuda.current_stream(device) == side and not tail) else WGRAD_WGS_ALONE


# [r6] split-K reduction of the weight gradients issued from the side stream: the slabs stay in per-layer workspaces and ONE multi-tensor
# launch (im2im_wgrad_reduce_multi) finishes every pending layer when the side stream is joined (end of the backward pass) or when a
# GradSync bucket is about to pack its gradients -- 18 launches of 8-50 us per step become one; per output the same slabs are added in the
# same order, so the same bits.  DEFAULT OFF (IM2IM_WGRAD_DEFER_REDUCE=1 enables): measured neutral at batch 78 (38.27 vs 38.28 ms) and
# 0.05 ms slower at batch 10 (6.46 vs 6.42) -- the 18 small launches sit on the side stream in the shadow of the main stream's kernels, and
# one late launch that reads every layer's slabs from HBM loses what the per-layer reductions found in L2 (profiles/r06_ab_experiments.txt 1).
WGRAD_DEFER_REDUCE = os.environ.get("IM2IM_WGRAD_DEFER_REDUCE", "0") != "0"
_pending_reduce = {}       # device index -> [(slab workspace, nsplit, Co, Ci, taps, dw)] in launch order


def flush_wgrad_reduce(device_index) -> None:
    """finish the pending weight gradients of a device on the CURRENT stream (call it on the stream that computed the slabs)"""
    items = _pending_reduce.pop(device_index, None)
    if not items:
        return
    n = len(items)
    arr, i32 = ctypes.c_void_p * n, ctypes.c_int32 * n
    check(lib.im2im_wgrad_reduce_multi(n, arr(*[it[0].data_ptr() for it in items]), i32(*[it[1] for it in items]), i32(*[it[2] for it in items]),
                                       i32(*[it[3] for it in items]), i32(*[it[4] for it in items]), arr(*[it[5].data_ptr() for it in items]),
                                       stream_ptr(torch.device("cuda", device_index))), "im2im_wgrad_reduce_multi")


def conv_wgrad(x, dz, taps, x_ss=None, x_hi=None, x_ss_hi=None, scratch_key="a", out=None, defer=False):
    """x [B,H,W,Ci], dz [B,H,W,Co] -> dw [Co,Ci,taps] fp32 (x_ss: lazy BatchNorm+ReLU of x, as in conv_fwd; x_hi: second
    half of the input channels as in conv_fwd).  out: a preallocated [Co,Ci,taps] fp32 result.
    defer=True: only the split-K slabs are computed (into the scratch buffer `scratch_key`, which must then be this layer's own);
    dw is filled by the next flush_wgrad_reduce on this stream."""
    b, h, w_, ci = x.shape
    ci_lo = ci
    if x_hi is not None:
        ci = 2 * ci
    co = dz.shape[3]
    nbytes = lib.im2im_conv_wgrad_workspace_bytes(b, h, w_, ci, co, taps)
    if nbytes < 0:
        raise _lib.Im2ImError(f"conv wgrad: unsupported channels Ci={ci} Co={co}")
    ws = _Scratch.get(nbytes, x.device, scratch_key)
    dw = torch.empty((co, ci, taps), dtype=F32, device=x.device) if out is None else out
    width = _wgrad_width(x.device)
    ev = TIMER.wrap(_tile_name("wgrad", h, w_, co, taps, x.dtype), 2.0 * b * h * w_ * co * ci * taps, x.device) if TIMER else None
    nsplit = ctypes.c_int32(0)
    check(lib.im2im_conv_wgrad_split(dptr(x), dptr(x_ss), dptr(x_hi), dptr(x_ss_hi), ci_lo, dptr(dz), None if defer else dptr(dw), dptr(ws),
                                     ws.numel(), b, h, w_, ci, co, taps, _DT[x.dtype], width, ctypes.addressof(nsplit), stream_ptr(x.device)),
          "im2im_conv_wgrad_split")
    if defer:
        _pending_reduce.setdefault(x.device.index, []).append((ws, nsplit.value, co, ci, taps, dw))
    if ev is not None:
        ev.record(torch.cuda.current_stream(x.device))
    return dw


def touched(*tensors):
    """tell torch that a kernel wrote these tensors through their raw pointers (the version counter is what the eval-mode
    weight cache below, and autograd's saved-tensor checks, go by)."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, centered=False, num_batches_tracked=None):
    rows, _, c = stats.shape
    dev = stats.device
    mean_invstd = torch.empty((2, c), dtype=F32, device=dev)
    scale_shift = torch.empty((2, c), dtype=F32, device=dev)
    ws = _Scratch.get(lib.im2im_reduce_workspace_bytes(3 * c), dev)
    check(lib.im2im_bn_finalize(dptr(stats), rows, c, count, dptr(gamma), dptr(beta), dptr(running_mean), dptr(running_var),
                                float(momentum), float(eps), int(centered), dptr(mean_invstd), dptr(scale_shift), dptr(ws),
                                dptr(_Scratch.counters(dev)), dptr(num_batches_tracked), stream_ptr(dev)),
          "im2im_bn_finalize")
    touched(running_mean, running_var, num_batches_tracked)
    return mean_invstd, scale_shift


def bn_fold_eval(gamma, beta, running_mean, running_var, conv_bias, eps):
    c = gamma.numel()
    out = torch.empty((2, c), dtype=F32, device=gamma.device)
    check(lib.im2im_bn_fold_eval(dptr(gamma), dptr(beta), dptr(running_mean), dptr(running_var), dptr(conv_bias), float(eps), c,
                                 dptr(out), stream_ptr(gamma.device)), "im2im_bn_fold_eval")
    return out


def bn_relu_apply(z, scale_shift):
    a = torch.empty_like(z)
    c = z.shape[-1]
    check(lib.im2im_bn_relu_apply(dptr(z), dptr(scale_shift), dptr(a), z.numel() // c, c, _DT[z.dtype], stream_ptr(z.device)),
          "im2im_bn_relu_apply")
    return a


_AB_SKIP_BN_APPLY = os.environ.get("IM2IM_AB_SKIP_BN_APPLY", "0") == "1"   # TIMING EXPERIMENT ONLY (wrong gradients): see below


def bn_relu_bwd(da, z, scale_shift, mean_invstd):
    c = z.shape[-1]
    m = z.numel() // c
    dev = z.device
    dgamma = torch.empty((c,), dtype=F32, device=dev)
    dbeta = torch.empty((c,), dtype=F32, device=dev)
    nbytes = lib.im2im_bn_bwd_workspace_bytes(m, c)
    ws = _Scratch.get(nbytes, dev)
    if _AB_SKIP_BN_APPLY:
        # upper bound of what an "apply on load" BatchNorm backward could gain (round-2 verdict #6): the reduction and its
        # finalisation run, the apply pass (read da, z; write dz) does not, consumers get da in place of dz -- i.e. the apply
        # pass is removed AND its replacement in the consumers is free.  profiles/r03_ab_experiments.txt
        for which, r1 in ((1, m), (2, 0)):
            check(lib.im2im_bn_relu_bwd_phase(dptr(da), dptr(z), dptr(scale_shift), dptr(mean_invstd), dptr(da), dptr(dgamma), dptr(dbeta),
                                              m, c, _DT[z.dtype], dptr(ws), ws.numel(), which, 0, r1, dptr(_Scratch.counters(dev)),
                                              stream_ptr(dev)), "im2im_bn_relu_bwd_phase")
        return da, dgamma, dbeta
    dz = torch.empty_like(z)
    check(lib.im2im_bn_relu_bwd(dptr(da), dptr(z), dptr(scale_shift), dptr(mean_invstd), dptr(dz), dptr(dgamma), dptr(dbeta), m, c,
                                _DT[z.dtype], dptr(ws), ws.numel(), dptr(_Scratch.counters(dev)), stream_ptr(dev)), "im2im_bn_relu_bwd")
    return dz, dgamma, dbeta


def bn_relu_bwd_pipelined(da, z, scale_shift, mean_invstd, halves, b_first):
    """bn_relu_bwd on the BatchNorm stream, in two halves of the batch (see BWD_PIPELINE): the reduction of images
    [0,b_first) starts at halves.first (or at once), the rest when the current stream's queued work is done; dz is
    published with the events at which its two halves are complete.  Bit-identical to bn_relu_bwd."""
    c = z.shape[-1]
    m = z.numel() // c
    dev = z.device
    idx = dev.index
    dz = torch.empty_like(z)
    dgamma = torch.empty((c,), dtype=F32, device=dev)
    dbeta = torch.empty((c,), dtype=F32, device=dev)
    ws = _Scratch.get(lib.im2im_bn_bwd_workspace_bytes(m, c), dev, "bn")      # this stream's own scratch
    ctr = _Scratch.counters(dev, "bn")                                        # ... and ticket counters
    rpb = lib.im2im_bn_bwd_rows_per_block(m)
    r_apply = b_first * (m // z.shape[0])
    r_reduce = (r_apply // rpb) * rpb                      # whole reduction blocks inside the first half
    main = torch.cuda.current_stream(dev)
    bn = bn_stream(dev)

    def phase(which, r0, r1):
        check(lib.im2im_bn_relu_bwd_phase(dptr(da), dptr(z), dptr(scale_shift), dptr(mean_invstd), dptr(dz), dptr(dgamma), dptr(dbeta),
                                          m, c, _DT[z.dtype], dptr(ws), ws.numel(), which, r0, r1, dptr(ctr), bn.cuda_stream), "im2im_bn_relu_bwd_phase")

    if halves is not None and halves.first is not None and r_reduce > 0:
        bn.wait_event(halves.first)
        phase(1, 0, r_reduce)
        bn.wait_stream(main)                               # the second half, and every earlier user of the new tensors' memory
        phase(1, r_reduce, m)
    else:
        bn.wait_stream(main)
        phase(1, 0, m)
    phase(2, 0, 0)
    phase(4, 0, r_apply)
    first = torch.cuda.Event()
    first.record(bn)
    phase(4, r_apply, m)
    rest = torch.cuda.Event()
    rest.record(bn)
    _publish_halves(dz, b_first, first, rest)
    _side_keep.setdefault(idx, []).extend((da, z, scale_shift, mean_invstd, dz))     # NOT dgamma/dbeta: autograd must be able to adopt them without a copy
    _queue_join(idx)
    return dz, dgamma, dbeta


def bn_relu_pool_bwd(da, dpool, z, scale_shift, mean_invstd):
    """BatchNorm+ReLU backward with the gradient g = da (or None) + maxpool2-backward(dpool) formed on the fly."""
    b, h, w_, c = z.shape
    dev = z.device
    dz = torch.empty_like(z)
    dgamma = torch.empty((c,), dtype=F32, device=dev)
    dbeta = torch.empty((c,), dtype=F32, device=dev)
    ws = _Scratch.get(lib.im2im_bn_relu_pool_bwd_workspace_bytes(b, h, w_, c), dev)
    check(lib.im2im_bn_relu_pool_bwd(dptr(da), dptr(dpool), dptr(z), dptr(scale_shift), dptr(mean_invstd), dptr(dz), dptr(dgamma),
                                     dptr(dbeta), b, h, w_, c, _DT[z.dtype], dptr(ws), ws.numel(), dptr(_Scratch.counters(dev)),
                                     stream_ptr(dev)),
          "im2im_bn_relu_pool_bwd")
    return dz, dgamma, dbeta


def colsum(x):
    c = x.shape[-1]
    m = x.numel() // c
    out = torch.empty((c,), dtype=F32, device=x.device)
    ws = _Scratch.get(lib.im2im_colsum_workspace_bytes(m, c), x.device)
    check(lib.im2im_colsum(dptr(x), dptr(out), m, c, _DT[x.dtype], dptr(ws), ws.numel(), stream_ptr(x.device)), "im2im_colsum")
    return out


def smallconv_s2l(x_nchw, w, bias, scale_shift, cl, dtype, relu=False, flip=False, want_stats=False, center=None):
    b, cs, h, w_ = x_nchw.shape
    out = torch.empty((b, h, w_, cl), dtype=dtype, device=x_nchw.device)
    stats = None
    if want_stats:
        stats = torch.empty((lib.im2im_smallconv_tiles(b, h, w_), 3, cl), dtype=F32, device=x_nchw.device)
    check(lib.im2im_smallconv_s2l_fwd(dptr(x_nchw), dptr(w), dptr(bias), dptr(center), dptr(scale_shift), dptr(out), dptr(stats), b, h, w_, cs, cl,
                                      int(relu), int(flip), _DT[dtype], stream_ptr(x_nchw.device)), "im2im_smallconv_s2l_fwd")
    return (out, stats) if want_stats else out


def smallconv_l2s(x, w, bias, cs):
    b, h, w_, cl = x.shape
    out = torch.empty((b, cs, h, w_), dtype=F32, device=x.device)
    check(lib.im2im_smallconv_l2s_fwd(dptr(x), dptr(w), dptr(bias), dptr(out), b, h, w_, cl, cs, _DT[x.dtype], stream_ptr(x.device)),
          "im2im_smallconv_l2s_fwd")
    return out


def smallconv_wgrad(s_nchw, l_nhwc, l_major, want_bias):
    b, cs, h, w_ = s_nchw.shape
    cl = l_nhwc.shape[3]
    dev = s_nchw.device
    dw = torch.empty((cl, cs, 9) if l_major else (cs, cl, 9), dtype=F32, device=dev)
    dbias = torch.empty((cs,), dtype=F32, device=dev) if want_bias else None
    ws = _Scratch.get(lib.im2im_smallconv_wgrad_workspace_bytes(b, h, w_, cs, cl), dev)
    check(lib.im2im_smallconv_wgrad(dptr(s_nchw), dptr(l_nhwc), dptr(dw), dptr(dbias), b, h, w_, cs, cl, int(l_major),
                                    _DT[l_nhwc.dtype], dptr(ws), ws.numel(), stream_ptr(dev)), "im2im_smallconv_wgrad")
    return dw, dbias


# ----------------------------------------------------------------------------------------- autograd
LAZY_ATTR = "_im2im_lazy_ss"
LINK_ATTR = "_im2im_bn_link"
FUSE_BN_REDUCE = os.environ.get("IM2IM_FUSE_BN_REDUCE", "1") != "0"    # a conv that is the only consumer of a lazy activation folds that layer's BatchNorm-backward
                          # reduction into its own data-gradient epilogue (im2im_conv_dgrad_bn): one pass over da and z fewer per layer
                          # and one launch fewer.  [r5] default ON: with the round-4 weight-gradient kernels the step gains 1.2 % at
                          # batch 78 (41.01 -> 40.54 ms, two alternating pairs on one box) and 2.0 % at the per-GPU batch of 10
                          # (7.04 -> 6.90 ms), profiles/r05_ab_experiments.txt section 3 -- rounds 1-4 measured +0.2..0.6 % and left it off.
                          # The price is visible in the conv kernel's own number: the data-gradients that carry the sums run ~5 %
                          # slower (the epilogue reads z and does ~8 VALU per value), so `roofline.achieved` of conv_igemm drops
                          # while img/s rises; IM2IM_FUSE_BN_REDUCE=0 restores the separate bandwidth-bound reduction


# ... but only where the normalised tensor has >= 128 channels, and not in OutConv's 1x1 data-gradient: on the 64-channel
# full-resolution layers the fused launch costs what the separate reduction did (rocprofv3, batch 78: conv_igemm<32,16,64,EPI 3>
# 896 us against 530 us for EPI 0 + a 0.32 ms reduction; the 1x1: 520 vs 273 us) -- the step is the same either way (39.03 / 39.01
# vs 39.02 / 39.08 ms, batch 10 6.44 / 6.43 vs 6.48 / 6.45) and the conv kernels stay leaner (conv roofline 929 -> 954 TF)
FUSE_BN_MIN_CH = int(os.environ.get("IM2IM_FUSE_BN_MIN_CH", "128"))
FUSE_BN_1X1 = os.environ.get("IM2IM_FUSE_BN_1X1", "0") != "0"


class BnLink:
    """what a lazy activation's consumers need to start its BatchNorm backward for it: the producer's z and coefficients,
    how many consumers read the activation in this forward, and (set by the consumer's backward) the partial sums."""
    __slots__ = ("z", "ss", "mi", "consumers", "partial")

    def __init__(self, z, ss, mi):
        self.z, self.ss, self.mi, self.consumers, self.partial = z, ss, mi, 0, None


def lazy_ss(x):
    """scale/shift [2,C] of a *lazy activation*: a tensor that physically holds the pre-BatchNorm conv output z and
    stands for a = relu(z*scale + shift).  Only the HIP consumers in this module understand it (they apply the
    transform while staging their operand); anything else must call materialize()."""
    link = getattr(x, LINK_ATTR, None)
    if link is not None:
        link.consumers += 1                    # every consumer of a lazy activation asks for its coefficients exactly once
    return getattr(x, LAZY_ATTR, None)


class ConvStats(torch.autograd.Function):
    """conv3x3(pad 1, bias) with BatchNorm batch statistics from the conv epilogue (unet_parts.py:16-17 / 19-20).
    Returns the pre-BN output z plus the BatchNorm coefficients; the normalisation itself is applied lazily by the
    consumers (BnReluLazy below carries its gradient).  The conv bias shifts the batch mean and cancels in the
    normalised output, so its gradient is identically zero here (the reference's autograd yields rounding noise)."""

    @staticmethod
    def forward(ctx, x, x_hi, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, cdt, nbt=None):
        """x_hi (or None): second half of the input channels -- the Up block's cat([skip, up]) without the copy."""
        _gpu(x, "input")
        co, ci = weight.shape[0], weight.shape[1]
        small = ci <= 8
        b, _, h, w_ = x.shape
        in_ss = lazy_ss(x)
        in_ss_hi = lazy_ss(x_hi) if x_hi is not None else None
        xin_hi = nhwc(x_hi.detach(), cdt) if x_hi is not None else None
        ctx.link = getattr(x, LINK_ATTR, None) if (x_hi is None and not small) else None
        fp8_d = None
        ctx.fp8_gs = None
        center = running_mean if (BF16_CENTERING and cdt == BF16 and running_mean is not None) else None
        if small:
            xin = x.detach().to(F32).contiguous()
            _, wd = pack_weight(weight, F32)
            z, stats = smallconv_s2l(xin, wd, bias.detach(), None, co, cdt, flip=True, want_stats=True, center=center)
            if not x.requires_grad:
                wd = None                                  # the network input: no data-gradient needed
        else:
            xin = nhwc(x.detach(), cdt)
            wd = None
            if center is None and fp8_eligible(ci, co, cdt, xin.shape[3] if xin_hi is not None else None):
                # (data-gradient on the fp8 kernel for 128-wide result tiles only: at 64 output channels it is no faster than the bf16 one)
                want_d = FP8_DGRAD and ci % 128 == 0 and (x.requires_grad or xin_hi is not None)
                (wq, wscale), fp8_d_packed = packed_fp8(weight, want_d)           # forward on the block-scaled fp8 MFMA
                z, stats = conv_fwd_fp8(xin, wq, wscale, bias.detach(), want_stats=True, in_ss=in_ss, x_hi=xin_hi, in_ss_hi=in_ss_hi)
                if want_d:
                    # ... and the data-gradient too (e5m2 dz under delayed scaling), and with it the weight gradient (FP8_WGRAD)
                    fp8_d = fp8_d_packed
                    gs = _fp8_grad_scales.get(weight)
                    if gs is None:
                        gs = Fp8GradScale()
                        if isinstance(weight, torch.nn.Parameter):
                            _fp8_grad_scales[weight] = gs
                    ctx.fp8_gs = gs
                else:
                    wd = packed_pair(weight, cdt)[1]
            else:
                wf, wd = packed_pair(weight, cdt)
                z, stats = conv_fwd(xin, wf, bias.detach(), want_stats=True, in_ss=in_ss, center=center, x_hi=xin_hi, in_ss_hi=in_ss_hi)
        if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
            nbt.add_(1)                                        # (a host-side counter: torch's own increment)
            nbt = None
        mean_invstd, scale_shift = bn_finalize(stats, b * h * w_, gamma.detach(), beta.detach(), running_mean, running_var,
                                               momentum, eps, centered=center is not None, num_batches_tracked=nbt)
        ctx.small = small
        ctx.weight_ref = weakref.ref(weight) if isinstance(weight, torch.nn.Parameter) or weight.is_leaf else None
        ctx.has = (in_ss is not None, xin_hi is not None, in_ss_hi is not None)
        ctx.set_materialize_grads(False)              # no zero tensors for the two non-differentiable outputs
        none = torch.empty(0)
        ctx.save_for_backward(xin, wd if wd is not None else none, in_ss if in_ss is not None else none,
                              xin_hi if xin_hi is not None else none, in_ss_hi if in_ss_hi is not None else none,
                              fp8_d[0] if fp8_d is not None else none, fp8_d[1] if fp8_d is not None else none)
        ctx.fp8_dgrad = fp8_d is not None
        zz = nchw(z)
        ctx.mark_non_differentiable(scale_shift, mean_invstd)
        return zz, scale_shift, mean_invstd

    @staticmethod
    def backward(ctx, dz, _g1, _g2):
        xin, wd, in_ss, xin_hi, in_ss_hi, wq_d, wscale_d = ctx.saved_tensors
        in_ss = in_ss if ctx.has[0] else None
        xin_hi = xin_hi if ctx.has[1] else None
        in_ss_hi = in_ss_hi if ctx.has[2] else None
        halves = _take_halves(dz)                     # dz from the pipelined BatchNorm backward lives on ITS stream until these events
        main = torch.cuda.current_stream(dz.device)
        if halves is not None and halves.rest is not None and (ctx.small or dz.dtype != xin.dtype):
            main.wait_event(halves.rest)
            halves = None
        dz = nhwc(dz, xin.dtype if not ctx.small else dz.dtype)
        dz_ready = halves.rest if halves is not None else None
        dx = dx_hi = None
        if ctx.small:
            dw, _ = smallconv_wgrad(xin, dz, l_major=True, want_bias=False)
            dw = dw.view(dz.shape[3], xin.shape[1], 3, 3)
            if ctx.needs_input_grad[0] and wd.numel():
                dx = smallconv_l2s(dz, wd, None, xin.shape[1])      # [B,Cin,H,W] fp32: correlation with the flipped taps
        else:
            ci = xin.shape[3] * (2 if xin_hi is not None else 1)
            # the weight gradient goes to the second stream (see _on_side_stream) unless autograd is about to ACCUMULATE it into
            # an existing .grad on this stream the moment we return (gradient accumulation over several backward passes)
            w_ref = ctx.weight_ref() if ctx.weight_ref is not None else None
            # fp8 mode: this layer's dz scale slots, taken ONCE per step (the weight gradient reads `previous`, the data-gradient
            # below reads it too and maintains the other two)
            fp8_slots = ctx.fp8_gs.slots(dz) if ctx.fp8_dgrad else None
            fp8_w = (ctx.fp8_dgrad and FP8_WGRAD and dz.dtype == BF16 and dz.shape[3] % 64 == 0 and ci % 64 == 0
                     and (FP8_WGRAD is True or dz.shape[3] % 128 != 0))

            def wgrad(out, key, defer=False):
                if fp8_w:
                    return conv_wgrad_fp8(xin, dz, fp8_slots[0], x_ss=in_ss, x_hi=xin_hi, x_ss_hi=in_ss_hi, scratch_key=key, out=out, defer=defer)
                return conv_wgrad(xin, dz, 9, x_ss=in_ss, x_hi=xin_hi, x_ss_hi=in_ss_hi, scratch_key=key, out=out, defer=defer)
            # ... or a tensor hook on the weight (wandb.watch, a user's register_hook) would read dW on THIS stream right away
            if (WGRAD_SIDE_STREAM and not torch.is_grad_enabled() and w_ref is not None and w_ref.grad is None
                    and not getattr(w_ref, "_backward_hooks", None)):
                dw_buf = torch.empty((dz.shape[3], ci, 9), dtype=F32, device=dz.device)     # owned by the current stream's pool
                # (the closure must not hold the tensor OBJECT handed to autograd: AccumulateGrad adopts an incoming gradient only
                # while nobody else references it, otherwise it clones it on the spot -- before the side stream has computed it)
                if WGRAD_DEFER_REDUCE:
                    # [r6] this layer's slabs stay in its own workspace ("side<n>": the n-th deferred weight gradient of the backward pass)
                    # until the side stream is joined; one multi-tensor launch reduces them all then (flush_wgrad_reduce)
                    slot = f"side{len(_pending_reduce.get(dz.device.index, ()))}"
                    _on_side_stream(dz.device, (xin, dz, in_ss, xin_hi, in_ss_hi, dw_buf), lambda dw_buf=dw_buf: wgrad(dw_buf, slot, True),
                                    after=dz_ready)
                else:
                    _on_side_stream(dz.device, (xin, dz, in_ss, xin_hi, in_ss_hi, dw_buf), lambda dw_buf=dw_buf: wgrad(dw_buf, "side"),
                                    after=dz_ready)
                dw = dw_buf.view(dz.shape[3], ci, 3, 3)
                del dw_buf
            else:
                if dz_ready is not None:
                    main.wait_event(dz_ready)
                    halves = dz_ready = None
                dw = wgrad(None, "a").view(dz.shape[3], ci, 3, 3)
            link = ctx.link
            fuse = (FUSE_BN_REDUCE and xin_hi is None and link is not None and link.consumers == 1 and link.z.dtype == dz.dtype
                    and not ctx.fp8_dgrad and link.z.shape[3] >= FUSE_BN_MIN_CH)
            b_first = 0
            if (xin_hi is not None or ctx.needs_input_grad[0]) and not fuse and not ctx.fp8_dgrad:
                b_first = halves.b_first if halves is not None else _pipeline_split(dz)
            if b_first:
                # data-gradient in two halves of the batch: the first waits only for ITS half of dz, and the BatchNorm
                # backward of the layer below may start on it while the second half is computed (see BWD_PIPELINE)
                b, h, w_ = dz.shape[0], dz.shape[1], dz.shape[2]
                co_in = wd.shape[0]
                split = xin.shape[3] if xin_hi is not None else 0
                dxn = torch.empty((b, h, w_, split if split else co_in), dtype=dz.dtype, device=dz.device)
                dxh = torch.empty((b, h, w_, co_in - split), dtype=dz.dtype, device=dz.device) if split else None
                if halves is not None and halves.first is not None:
                    main.wait_event(halves.first)
                conv_fwd(dz[:b_first], wd, split_out=split, out=dxn[:b_first], out_hi=dxh[:b_first] if split else None)
                first = torch.cuda.Event()
                first.record(main)
                if dz_ready is not None:
                    main.wait_event(dz_ready)
                conv_fwd(dz[b_first:], wd, split_out=split, out=dxn[b_first:], out_hi=dxh[b_first:] if split else None)
                if split:
                    dx, dx_hi = nchw(dxn), nchw(dxh)
                else:
                    _publish_halves(dxn, b_first, first)
                    dx = nchw(dxn)
            else:
                if dz_ready is not None:
                    main.wait_event(dz_ready)
                if ctx.fp8_dgrad and (xin_hi is not None or ctx.needs_input_grad[0]):
                    # fp8 mode: e5m2 dz x e4m3 weights on the block-scaled MFMA (conv_fp8.hip, GRAD form)
                    if xin_hi is not None:
                        dx, dx_hi = conv_dgrad_fp8(dz, wq_d, wscale_d, ctx.fp8_gs, split_out=xin.shape[3], slots=fp8_slots)
                        dx, dx_hi = nchw(dx), nchw(dx_hi)
                    else:
                        dx = nchw(conv_dgrad_fp8(dz, wq_d, wscale_d, ctx.fp8_gs, slots=fp8_slots))
                elif xin_hi is not None:
                    # the data-gradient lands directly in d(skip) and d(up): no concatenated gradient tensor
                    dx, dx_hi = conv_fwd(dz, wd, split_out=xin.shape[3])
                    dx, dx_hi = nchw(dx), nchw(dx_hi)
                elif ctx.needs_input_grad[0]:
                    if fuse:
                        # sole consumer of a lazy activation: its BatchNorm-backward sums ride on this data-gradient's epilogue
                        dx, link.partial = conv_dgrad_bn(dz, wd, link.z, link.ss, link.mi)
                        dx = nchw(dx)
                    else:
                        dx = nchw(conv_fwd(dz, wd))           # gradient w.r.t. the (lazy) input activation
        return dx, dx_hi, dw, None, None, None, None, None, None, None, None, None


class BnReluLazy(torch.autograd.Function):
    """a = relu(BatchNorm_train(z)) WITHOUT materialising a: forward returns an alias of z tagged with the BatchNorm
    scale/shift; backward is the full BatchNorm+ReLU backward (dz, dgamma, dbeta) from the summed gradient of all
    consumers (unet_parts.py:17-18 / 20-21)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, scale_shift, mean_invstd, link=None):
        ctx.save_for_backward(z, scale_shift, mean_invstd)
        ctx.link = link
        return z.detach().view_as(z)

    @staticmethod
    def backward(ctx, da):
        z, scale_shift, mean_invstd = ctx.saved_tensors
        zz = nhwc(z)
        link = ctx.link
        partial = link.partial if link is not None else None
        halves = _take_halves(da)                     # da always completes in the order of the current stream; `first` is a head start
        dan = nhwc(da, zz.dtype)
        if partial is not None:
            link.partial = None                       # the consumer's data-gradient already reduced g and g*xhat per tile
            dz, dgamma, dbeta = bn_relu_bwd_from_partial(dan, zz, scale_shift, mean_invstd, partial)
        else:
            b_first = _pipeline_split(zz) if dan.data_ptr() == da.data_ptr() else 0
            if b_first and halves is not None:
                b_first = halves.b_first
            if b_first:
                dz, dgamma, dbeta = bn_relu_bwd_pipelined(dan, zz, scale_shift, mean_invstd, halves, b_first)
            else:
                dz, dgamma, dbeta = bn_relu_bwd(dan, zz, scale_shift, mean_invstd)
        return nchw(dz), dgamma, dbeta, None, None, None


FUSE_POOL_BWD = True      # skip layers: max-pool backward + gradient add folded into the BatchNorm backward kernels


def can_fuse_pool_bwd(z) -> bool:
    vpr = z.shape[1] // (8 if z.dtype == BF16 else 4)
    return FUSE_POOL_BWD and vpr > 0 and (vpr & (vpr - 1)) == 0 and vpr <= 256 and z.shape[2] >= 2 and z.shape[3] >= 2


class BnReluLazyPool(torch.autograd.Function):
    """BnReluLazy for an activation that feeds a skip connection AND Down.maxpool (unet.py:35-38): returns the lazy
    activation and MaxPool2d(2) of it; backward takes both gradients and runs ONE fused BatchNorm+ReLU backward that
    scatters the pooled gradient to the window maxima and adds the skip gradient on the fly (no max-pool backward
    tensor, no gradient-accumulation add)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, scale_shift, mean_invstd):
        zz = nhwc(z.detach())
        b, h, w_, c = zz.shape
        y = torch.empty((b, h // 2, w_ // 2, c), dtype=zz.dtype, device=zz.device)
        check(lib.im2im_maxpool2_fwd(dptr(zz), dptr(scale_shift), dptr(y), b, h, w_, c, _DT[zz.dtype], stream_ptr(zz.device)),
              "im2im_maxpool2_fwd")
        ctx.save_for_backward(z, scale_shift, mean_invstd)
        ctx.set_materialize_grads(False)
        return z.detach().view_as(z), nchw(y)

    @staticmethod
    def backward(ctx, da, dpool):
        z, scale_shift, mean_invstd = ctx.saved_tensors
        zz = nhwc(z)
        if dpool is None:
            dz, dgamma, dbeta = bn_relu_bwd(nhwc(da, zz.dtype), zz, scale_shift, mean_invstd)
        else:
            dz, dgamma, dbeta = bn_relu_pool_bwd(nhwc(da, zz.dtype) if da is not None else None, nhwc(dpool, zz.dtype), zz,
                                                 scale_shift, mean_invstd)
        return nchw(dz), dgamma, dbeta, None, None


class Materialize(torch.autograd.Function):
    """turn a lazy activation into a plain tensor (one BatchNorm+ReLU pass); identity for the gradient."""

    @staticmethod
    def forward(ctx, x, scale_shift):
        if scale_shift.dim() == 3:                             # per-image coefficients: a GroupNorm producer
            return nchw(affine_relu_apply_per_image(nhwc(x.detach()), scale_shift))
        return nchw(bn_relu_apply(nhwc(x.detach()), scale_shift))

    @staticmethod
    def backward(ctx, g):
        return g, None


def materialize(x):
    ss = lazy_ss(x)
    return x if ss is None else Materialize.apply(x, ss)


def conv_bn_relu_train(x, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, cdt, lazy_out=False, x_hi=None,
                       pool=False, num_batches_tracked=None):
    """pool=True (lazy_out only): also return MaxPool2d(2) of the activation -> (a, pooled).
    num_batches_tracked: the BatchNorm module's counter, incremented by the statistics kernel (no separate launch)."""
    z, scale_shift, mean_invstd = ConvStats.apply(x, x_hi, weight, bias, gamma, beta, running_mean, running_var, momentum, eps, cdt,
                                                  num_batches_tracked)
    if pool and lazy_out and can_fuse_pool_bwd(z):
        a, pooled = BnReluLazyPool.apply(z, gamma, beta, scale_shift, mean_invstd)
        setattr(a, LAZY_ATTR, scale_shift)
        return a, pooled
    link = BnLink(nhwc(z.detach()), scale_shift, mean_invstd) if lazy_out else None
    a = BnReluLazy.apply(z, gamma, beta, scale_shift, mean_invstd, link)
    setattr(a, LAZY_ATTR, scale_shift)
    if link is not None:
        setattr(a, LINK_ATTR, link)
    a = a if lazy_out else materialize(a)
    return (a, MaxPool2.apply(a)) if pool else a


_EVAL_CACHE = weakref.WeakKeyDictionary()      # conv module -> (key, packed operands): see conv_bn_relu_eval


FUSE_EVAL_OUTCONV = os.environ.get("IM2IM_FUSE_EVAL_OUTCONV", "1") != "0"    # [r4] eval: OutConv's 1x1 in the last block's conv epilogue


def conv_bn_relu_eval(x, weight, bias, gamma, beta, running_mean, running_var, eps, cdt, owner=None, x_hi=None, pool=False, tail=None):
    """eval mode: BatchNorm folded into the conv epilogue (one kernel, no intermediate).
    pool=True [r4]: also return MaxPool2d(2) of the result -> (a, pooled), taken from the epilogue's LDS tile when the extent is
    even and the conv runs on the bf16 / fp32 MFMA kernel (otherwise the caller pools separately: a plain tensor is returned).
    tail=<OutConv's nn.Conv2d> [r4]: the block's result is consumed only by that 1x1 convolution (64 -> 32): return
    conv1x1(result) computed on the epilogue's LDS tile (same bits as Conv1x1 on the stored result; the 64-channel tensor never
    reaches HBM), tagged `_im2im_tail_done`; when the shapes are not the fused kernel's the plain result comes back untagged.

    owner (the conv module): the folded coefficients and the packed weight are kept for it and reused while none of the
    six tensors changed (storage and version counter) -- calibration and validation run hundreds of forwards over the
    same weights, and the 36 packing / folding launches per forward were 2 % of one."""
    _gpu(x, "input")
    co, ci = weight.shape[0], weight.shape[1]
    small = ci <= 8
    fp8 = not small and fp8_eligible(ci, co, cdt, x.shape[1] if x_hi is not None else None)
    key = (cdt, fp8, float(eps)) + tuple(v for t in (weight, bias, gamma, beta, running_mean, running_var) for v in (t.data_ptr(), t._version))
    hit = _EVAL_CACHE.get(owner) if owner is not None else None
    if hit is not None and hit[0] == key:
        fold, packed = hit[1]
    else:
        fold = bn_fold_eval(gamma.detach(), beta.detach(), running_mean, running_var, bias.detach(), eps)
        if small:
            packed = pack_weight(weight, F32)[1]
        elif fp8:
            packed = pack_weight_fp8(weight)
        else:
            packed = pack_weight(weight, cdt, want_wd=False)[0]
        if owner is not None:
            _EVAL_CACHE[owner] = (key, (fold, packed))
    if small:
        xin = x.detach().to(F32).contiguous()
        return nchw(smallconv_s2l(xin, packed, None, fold, co, cdt, relu=True, flip=True))
    xin = nhwc(x.detach(), cdt)
    xin_hi = nhwc(x_hi.detach(), cdt) if x_hi is not None else None
    if fp8:
        return nchw(conv_fwd_fp8(xin, packed[0], packed[1], None, fold, relu=True, x_hi=xin_hi))
    b, h, w_, cin = xin.shape
    if (tail is not None and FUSE_EVAL_OUTCONV and not torch.is_grad_enabled() and co == 64 and tuple(tail.weight.shape) == (32, 64, 1, 1)
            and tail.weight.is_cuda):
        tkey = (cdt,) + tuple(v for t in (tail.weight, tail.bias) for v in (t.data_ptr(), t._version))
        thit = _EVAL_CACHE.get(tail)
        if thit is not None and thit[0] == tkey:
            w1, b1 = thit[1]
        else:
            w1 = pack_weight(tail.weight, cdt, want_wd=False)[0]              # [32][1][64]
            b1 = tail.bias.detach().to(F32).contiguous()
            _EVAL_CACHE[tail] = (tkey, (w1, b1))
        f = torch.empty((b, h, w_, 32), dtype=cdt, device=xin.device)
        ev = TIMER.wrap(_tile_name("igemm", h, w_, co, 9, cdt), 2.0 * b * h * w_ * co * ci * 9, xin.device) if TIMER else None
        check(lib.im2im_conv_fwd_eval_tail(dptr(xin), dptr(xin_hi), cin, dptr(packed), dptr(fold[0]), dptr(fold[1]), dptr(w1), dptr(b1), dptr(f),
                                           b, h, w_, ci, 32, _DT[cdt], stream_ptr(xin.device)), "im2im_conv_fwd_eval_tail")
        if ev is not None:
            ev.record(torch.cuda.current_stream(xin.device))
        out = nchw(f)
        out._im2im_tail_done = True
        return out
    if pool and not torch.is_grad_enabled() and h % 2 == 0 and w_ % 2 == 0 and h >= 2 and w_ >= 2:
        y = torch.empty((b, h, w_, co), dtype=cdt, device=xin.device)
        pooled = torch.empty((b, h // 2, w_ // 2, co), dtype=cdt, device=xin.device)
        ev = TIMER.wrap(_tile_name("igemm", h, w_, co, 9, cdt), 2.0 * b * h * w_ * co * ci * 9, xin.device) if TIMER else None
        check(lib.im2im_conv_fwd_eval_pool(dptr(xin), dptr(xin_hi), cin, dptr(packed), dptr(fold[0]), dptr(fold[1]), dptr(y), dptr(pooled),
                                           b, h, w_, ci, co, _DT[cdt], stream_ptr(xin.device)), "im2im_conv_fwd_eval_pool")
        if ev is not None:
            ev.record(torch.cuda.current_stream(xin.device))
        return nchw(y), nchw(pooled)
    return nchw(conv_fwd(xin, packed, None, fold, relu=True, x_hi=xin_hi))


_HEAD_ACT = {"relu": 0, "abs": 1}


# ----------------------------------------------------------------------------------------- GroupNorm (north-star extra)
def conv_fwd_per_image(x, wf, bias, in_ss_img=None):
    """conv with one image per tile: y [B,H,W,Co] + per-tile statistics rows grouped by image -> (y, stats, rows_per_image).
    in_ss_img [B,2,Ci]: x is a GroupNorm producer's pre-norm z, max(z*scale+shift, 0) is applied while staging."""
    b, h, w_, ci = x.shape
    co, taps = wf.shape[0], wf.shape[1]
    rpi = lib.im2im_conv_tiles_per_image(h, w_)
    y = torch.empty((b, h, w_, co), dtype=x.dtype, device=x.device)
    stats = torch.empty((b * rpi, 3, co), dtype=F32, device=x.device)
    ev = TIMER.wrap(f"conv_igemm_kernel<{'bf16' if x.dtype == BF16 else 'f32'},1x16x16,gn,taps={taps}>", 2.0 * b * h * w_ * co * ci * taps,
                    x.device) if TIMER else None
    check(lib.im2im_conv_fwd_per_image(dptr(x), dptr(in_ss_img), dptr(wf), dptr(bias), dptr(y), dptr(stats), b, h, w_, ci, co, taps,
                                       _DT[x.dtype], stream_ptr(x.device)), "im2im_conv_fwd_per_image")
    if ev is not None:
        ev.record(torch.cuda.current_stream(x.device))
    return y, stats, rpi


def groupnorm_stats(z):
    """(mean, M2, n) partial rows of an arbitrary NHWC tensor [B,H,W,C] -> (stats, rows_per_image)."""
    b, c = z.shape[0], z.shape[-1]
    hw = z.numel() // (b * c)
    rows = lib.im2im_groupnorm_stats_rows(b, hw)
    stats = torch.empty((rows, 3, c), dtype=F32, device=z.device)
    check(lib.im2im_groupnorm_stats(dptr(z), b, hw, c, _DT[z.dtype], dptr(stats), stream_ptr(z.device)), "im2im_groupnorm_stats")
    return stats, rows // b


def groupnorm_finalize(stats, b, rows_per_image, gamma, beta, groups, eps):
    c = stats.shape[-1]
    mean_rstd = torch.empty((b, 2, c), dtype=F32, device=stats.device)
    scale_shift = torch.empty((b, 2, c), dtype=F32, device=stats.device)
    check(lib.im2im_groupnorm_finalize(dptr(stats), b, rows_per_image, c, int(groups), dptr(gamma), dptr(beta), float(eps),
                                       dptr(mean_rstd), dptr(scale_shift), stream_ptr(stats.device)), "im2im_groupnorm_finalize")
    return mean_rstd, scale_shift


def affine_relu_apply_per_image(z, ss):
    a = torch.empty_like(z)
    b, c = z.shape[0], z.shape[-1]
    check(lib.im2im_affine_relu_apply_per_image(dptr(z), dptr(ss), dptr(a), b, z.numel() // (b * c), c, _DT[z.dtype], stream_ptr(z.device)),
          "im2im_affine_relu_apply_per_image")
    return a


def groupnorm_relu_bwd(da, z, ss, mean_rstd, gamma, groups):
    b, c = z.shape[0], z.shape[-1]
    hw = z.numel() // (b * c)
    dev = z.device
    dz = torch.empty_like(z)
    dgamma = torch.empty((c,), dtype=F32, device=dev)
    dbeta = torch.empty((c,), dtype=F32, device=dev)
    ws = _Scratch.get(lib.im2im_groupnorm_relu_bwd_workspace_bytes(b, hw, c), dev)
    check(lib.im2im_groupnorm_relu_bwd(dptr(da), dptr(z), dptr(ss), dptr(mean_rstd), dptr(gamma), dptr(dz), dptr(dgamma), dptr(dbeta),
                                       b, hw, c, int(groups), _DT[z.dtype], dptr(ws), ws.numel(), stream_ptr(dev)),
          "im2im_groupnorm_relu_bwd")
    return dz, dgamma, dbeta


class ConvGroupStats(torch.autograd.Function):
    """conv3x3(pad 1, bias) whose epilogue statistics are finalised per image and channel group (GroupNorm): returns the
    pre-norm output z and per-(image, channel) coefficients; the normalisation is applied by the consumer (GnReluLazy).
    The input may itself be a GroupNorm-lazy activation (coefficients [B,2,Ci]) -- the DoubleConv's second conv."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, groups, eps, cdt):
        _gpu(x, "input")
        co, ci = weight.shape[0], weight.shape[1]
        small = ci <= 8
        b, _, h, w_ = x.shape
        in_ss = getattr(x, LAZY_ATTR, None)
        if in_ss is not None and in_ss.dim() != 3:
            raise _lib.Im2ImError("ConvGroupStats: a BatchNorm-lazy input must be materialised first")
        if small:
            xin = x.detach().to(F32).contiguous()
            _, wd = pack_weight(weight, F32)
            z, stats = smallconv_s2l(xin, wd, bias.detach(), None, co, cdt, flip=True, want_stats=True)
            rpi = lib.im2im_smallconv_tiles(1, h, w_)
            if not x.requires_grad:
                wd = None
        else:
            xin = nhwc(x.detach(), cdt)
            wf, wd = pack_weight(weight, cdt)
            z, stats, rpi = conv_fwd_per_image(xin, wf, bias.detach(), in_ss_img=in_ss)
        mean_rstd, scale_shift = groupnorm_finalize(stats, b, rpi, gamma.detach(), beta.detach(), groups, eps)
        ctx.small, ctx.lazy_in = small, in_ss is not None
        ctx.set_materialize_grads(False)
        none = torch.empty(0)
        ctx.save_for_backward(xin, wd if wd is not None else none, in_ss if in_ss is not None else none)
        ctx.mark_non_differentiable(scale_shift, mean_rstd)
        return nchw(z), scale_shift, mean_rstd

    @staticmethod
    def backward(ctx, dz, _g1, _g2):
        xin, wd, in_ss = ctx.saved_tensors
        dz = nhwc(dz, xin.dtype if not ctx.small else dz.dtype)
        dx = None
        if ctx.small:
            dw, _ = smallconv_wgrad(xin, dz, l_major=True, want_bias=False)
            dw = dw.view(dz.shape[3], xin.shape[1], 3, 3)
            db = colsum(dz)
            if ctx.needs_input_grad[0] and wd.numel():
                dx = smallconv_l2s(dz, wd, None, xin.shape[1])
        else:
            a = affine_relu_apply_per_image(xin, in_ss) if ctx.lazy_in else xin      # recomputed operand of the weight gradient
            dw = conv_wgrad(a, dz, 9).view(dz.shape[3], xin.shape[3], 3, 3)
            db = colsum(dz)                                   # the bias does NOT cancel under GroupNorm (the mean is per image/group)
            if ctx.needs_input_grad[0]:
                dx = nchw(conv_fwd(dz, wd))
        return dx, dw, db, None, None, None, None, None


class GnReluLazy(torch.autograd.Function):
    """a = relu(GroupNorm(z)) without materialising a: forward returns an alias of z (the caller tags it with the per-image
    coefficients); backward is the GroupNorm+ReLU backward (dz, dgamma, dbeta)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, scale_shift, mean_rstd, groups):
        ctx.save_for_backward(z, gamma, scale_shift, mean_rstd)
        ctx.groups = groups
        return z.detach().view_as(z)

    @staticmethod
    def backward(ctx, da):
        z, gamma, scale_shift, mean_rstd = ctx.saved_tensors
        zz = nhwc(z)
        dz, dgamma, dbeta = groupnorm_relu_bwd(nhwc(da, zz.dtype), zz, scale_shift, mean_rstd, gamma.detach(), ctx.groups)
        return nchw(dz), dgamma, dbeta, None, None, None


def conv_gn_relu(x, weight, bias, gamma, beta, groups, eps, cdt, lazy_out=False):
    """conv3x3 -> GroupNorm -> ReLU (DoubleConv(norm="group")); GroupNorm has no train/eval distinction."""
    ss_in = getattr(x, LAZY_ATTR, None)
    if ss_in is not None and ss_in.dim() != 3:
        x = materialize(x)
    z, scale_shift, mean_rstd = ConvGroupStats.apply(x, weight, bias, gamma, beta, groups, eps, cdt)
    a = GnReluLazy.apply(z, gamma, beta, scale_shift, mean_rstd, groups)
    setattr(a, LAZY_ATTR, scale_shift)
    return a if lazy_out else materialize(a)


def group_norm_relu(x, gamma, beta, groups, eps=1e-5):
    """standalone relu(GroupNorm(x)) of an arbitrary [B,C,H,W] GPU tensor (channels-last in its own dtype): statistics pass,
    finalize, one apply pass; differentiable (same backward kernels)."""
    return _GroupNormRelu.apply(x, gamma, beta, groups, eps)


class _GroupNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        z = nhwc(x.detach(), x.dtype if x.dtype in (F32, BF16) else F32)
        stats, rpi = groupnorm_stats(z)
        mean_rstd, scale_shift = groupnorm_finalize(stats, z.shape[0], rpi, gamma.detach(), beta.detach(), groups, eps)
        ctx.save_for_backward(z, gamma, scale_shift, mean_rstd)
        ctx.groups = groups
        return nchw(affine_relu_apply_per_image(z, scale_shift))

    @staticmethod
    def backward(ctx, da):
        z, gamma, scale_shift, mean_rstd = ctx.saved_tensors
        dz, dgamma, dbeta = groupnorm_relu_bwd(nhwc(da, z.dtype), z, scale_shift, mean_rstd, gamma.detach(), ctx.groups)
        return nchw(dz), dgamma, dbeta, None, None


class EvalModeBarrier(torch.autograd.Function):
    """identity in the forward; its backward raises: marks the output of a fused eval-mode conv+BatchNorm (which records no
    autograd graph) so that back-propagating through it fails loudly instead of yielding no / partial gradients."""

    @staticmethod
    def forward(ctx, y, *deps):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("eval-mode backward through the fused conv+BatchNorm kernel is not implemented: "
                                  "run inference under torch.no_grad(), or put the block in train() mode to fine-tune")


class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ss = lazy_ss(x)
        xin = nhwc(x.detach())
        b, h, w_, c = xin.shape
        y = torch.empty((b, h // 2, w_ // 2, c), dtype=xin.dtype, device=xin.device)
        check(lib.im2im_maxpool2_fwd(dptr(xin), dptr(ss), dptr(y), b, h, w_, c, _DT[xin.dtype], stream_ptr(xin.device)),
              "im2im_maxpool2_fwd")
        ctx.has_ss = ss is not None
        ctx.save_for_backward(xin, ss if ss is not None else torch.empty(0))
        return nchw(y)

    @staticmethod
    def backward(ctx, dy):
        xin, ss = ctx.saved_tensors
        ss = ss if ctx.has_ss else None
        dy = nhwc(dy, xin.dtype)
        b, h, w_, c = xin.shape
        dx = torch.empty_like(xin)
        check(lib.im2im_maxpool2_bwd(dptr(xin), dptr(ss), dptr(dy), dptr(dx), b, h, w_, c, _DT[xin.dtype], stream_ptr(xin.device)),
              "im2im_maxpool2_bwd")
        return nchw(dx)


class UpsampleConcat(torch.autograd.Function):
    """cat([skip, zero_pad(bilinear x2 align_corners(deep))], dim=1)  (unet_parts.py:58-68) in one pass; both inputs
    may be lazy activations."""

    @staticmethod
    def forward(ctx, deep, skip):
        dss, sss = lazy_ss(deep), lazy_ss(skip)
        d = nhwc(deep.detach())
        s = nhwc(skip.detach(), d.dtype)
        b, h, w_, cd = d.shape
        _, hh, ww, cs = s.shape
        out = torch.empty((b, hh, ww, cs + cd), dtype=d.dtype, device=d.device)
        check(lib.im2im_upsample2x_concat_fwd(dptr(d), dptr(dss), dptr(s), dptr(sss), dptr(out), b, h, w_, cd, hh, ww, cs,
                                              _DT[d.dtype], stream_ptr(d.device)), "im2im_upsample2x_concat_fwd")
        ctx.shape = (b, h, w_, cd, hh, ww, cs)
        return nchw(out)

    @staticmethod
    def backward(ctx, dout):
        b, h, w_, cd, hh, ww, cs = ctx.shape
        dout = nhwc(dout)
        ddeep = torch.empty((b, h, w_, cd), dtype=dout.dtype, device=dout.device)
        dskip = torch.empty((b, hh, ww, cs), dtype=dout.dtype, device=dout.device)
        check(lib.im2im_upsample2x_concat_bwd(dptr(dout), dptr(ddeep), dptr(dskip), b, h, w_, cd, hh, ww, cs, _DT[dout.dtype],
                                              stream_ptr(dout.device)), "im2im_upsample2x_concat_bwd")
        return nchw(ddeep), nchw(dskip)


MATERIALIZE_BEFORE_UPSAMPLE = True
SPLIT_CONCAT = True       # Up blocks: the first conv reads [skip, upsampled] from two tensors instead of a concatenated copy


def can_split_concat(deep, skip) -> bool:
    """the split-operand kernels need both halves equally wide and a multiple of 64 channels (true for the reference
    UNet with bilinear=True: 512+512, 256+256, 128+128, 64+64)."""
    return SPLIT_CONCAT and deep.shape[1] == skip.shape[1] and skip.shape[1] % 64 == 0


class Upsample2x(torch.autograd.Function):
    """zero_pad(bilinear x2 align_corners(deep)) to the skip extent (unet_parts.py:58-66) -- the half of the Up-block
    concatenation that has to be computed; the skip half is read in place by the consumer conv (conv_fwd x_hi=...)."""

    @staticmethod
    def forward(ctx, deep, hh, ww):
        dss = lazy_ss(deep)
        d = nhwc(deep.detach())
        if dss is not None and MATERIALIZE_BEFORE_UPSAMPLE:
            # the interpolation kernel is VALU-bound when it applies BatchNorm+ReLU to each of its four taps; the deep map
            # is 1/4 the size of the result, so normalising it once first is cheaper (same values: the in-kernel path rounds
            # each tap to the storage type exactly like this pass does)
            d, dss = bn_relu_apply(d, dss), None
        b, h, w_, cd = d.shape
        out = torch.empty((b, hh, ww, cd), dtype=d.dtype, device=d.device)
        check(lib.im2im_upsample2x_concat_fwd(dptr(d), dptr(dss), None, None, dptr(out), b, h, w_, cd, hh, ww, 0,
                                              _DT[d.dtype], stream_ptr(d.device)), "im2im_upsample2x_concat_fwd")
        ctx.shape = (b, h, w_, cd, hh, ww)
        return nchw(out)

    @staticmethod
    def backward(ctx, dout):
        b, h, w_, cd, hh, ww = ctx.shape
        dout = nhwc(dout)
        ddeep = torch.empty((b, h, w_, cd), dtype=dout.dtype, device=dout.device)
        check(lib.im2im_upsample2x_concat_bwd(dptr(dout), dptr(ddeep), None, b, h, w_, cd, hh, ww, 0, _DT[dout.dtype],
                                              stream_ptr(dout.device)), "im2im_upsample2x_concat_bwd")
        return nchw(ddeep), None, None


def depth_space2(t, co, to_space):
    """pixel shuffle of the 2x2 transposed conv: [B,h,w,4*co] -> [B,2h,2w,co] (to_space) or back."""
    if to_space:
        b, h, w_, _ = t.shape
        out = torch.empty((b, 2 * h, 2 * w_, co), dtype=t.dtype, device=t.device)
    else:
        b, h2, w2, _ = t.shape
        h, w_ = h2 // 2, w2 // 2
        out = torch.empty((b, h, w_, 4 * co), dtype=t.dtype, device=t.device)
    check(lib.im2im_depth_space2(dptr(t), dptr(out), b, h, w_, co, int(to_space), _DT[t.dtype], stream_ptr(t.device)), "im2im_depth_space2")
    return out


class ConvTranspose2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2) (Up with bilinear=False, unet_parts.py:53): every input pixel
    produces a 2x2 output patch and patches do not overlap, so it is ONE 1x1 convolution to 4*Co channels on the MFMA
    kernel followed by a depth-to-space rearrangement; the backward is the mirror image (space-to-depth, 1x1 data- and
    weight-gradient).  The input may be a lazy activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt):
        ss = lazy_ss(x)
        xin = nhwc(x.detach(), cdt)
        b, h, w_, ci = xin.shape
        co = weight.shape[1]
        w1 = weight.detach().permute(2, 3, 1, 0).reshape(4 * co, ci, 1, 1)        # row (a, b, co) <- weight[ci, co, a, b]
        wf, wd = pack_weight(w1, cdt)
        y4 = conv_fwd(xin, wf, bias.detach().to(F32).repeat(4), in_ss=ss)          # [B,h,w,(a,b,co)]
        y = depth_space2(y4, co, to_space=True)
        ctx.has_ss = ss is not None
        ctx.save_for_backward(xin, wd, ss if ss is not None else torch.empty(0))
        return nchw(y)

    @staticmethod
    def backward(ctx, dy):
        xin, wd, ss = ctx.saved_tensors
        ss = ss if ctx.has_ss else None
        b, h, w_, ci = xin.shape
        dy = nhwc(dy, xin.dtype)
        co = dy.shape[3]
        d4 = depth_space2(dy.contiguous(), co, to_space=False)
        dx = nchw(conv_fwd(d4, wd)) if ctx.needs_input_grad[0] else None
        dw = conv_wgrad(xin, d4, 1, x_ss=ss).view(2, 2, co, ci).permute(3, 2, 0, 1).contiguous()
        db = colsum(dy.contiguous())                           # every output pixel gets the bias once
        return dx, dw, db, None


class Conv1x1(torch.autograd.Function):
    """OutConv (unet_parts.py:87-94): 1x1 conv with bias on the MFMA kernel (taps = 1); input may be lazy."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt):
        ss = lazy_ss(x)
        xin = nhwc(x.detach(), cdt)
        wf, wd = packed_pair(weight, cdt)                     # with the 3x3 weights' per-step batch (one launch for the model)
        y = conv_fwd(xin, wf, bias.detach(), in_ss=ss)
        ctx.has_ss = ss is not None
        ctx.link = getattr(x, LINK_ATTR, None)
        ctx.save_for_backward(xin, wd, ss if ss is not None else torch.empty(0))
        return nchw(y)

    @staticmethod
    def backward(ctx, dy):
        xin, wd, ss = ctx.saved_tensors
        ss = ss if ctx.has_ss else None
        dy = nhwc(dy, xin.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            link = ctx.link
            if FUSE_BN_REDUCE and FUSE_BN_1X1 and link is not None and link.consumers == 1 and link.z.dtype == dy.dtype:
                dx, link.partial = conv_dgrad_bn(dy, wd, link.z, link.ss, link.mi)     # as in ConvStats.backward
                dx = nchw(dx)
            else:
                dx = nchw(conv_fwd(dy, wd))
        dw = conv_wgrad(xin, dy, 1, x_ss=ss).view(dy.shape[3], xin.shape[3], 1, 1)
        return dx, dw, colsum(dy), None


_HEADS_CACHE = WeakTensorKeyDictionary()     # first head weight -> (key, packed weights, concatenated biases): inference only


def _packed_heads(ws, bs):
    """([K*C][9][Cmid] fp32 packed head weights, [K*C] fp32 biases) of K 3x3 heads.  Under torch.no_grad() (calibration and
    validation run hundreds of forwards over the same weights) the pair is kept while none of the tensors changed -- the two
    concatenations and the packing were three launches per forward."""
    key = None
    if not torch.is_grad_enabled():
        key = tuple(v for t in (*ws, *bs) for v in (t.data_ptr(), t._version))
        hit = _HEADS_CACHE.get(ws[0])
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
    w_all = torch.cat([w.detach() for w in ws], dim=0)
    b_all = torch.cat([b.detach() for b in bs], dim=0).to(F32).contiguous()
    wf, _ = pack_weight(w_all, F32, want_wd=False)
    if key is not None:
        _HEADS_CACHE[ws[0]] = (key, wf, b_all)
    return wf, b_all


class QuantileHeads(torch.autograd.Function):
    """Three 3x3 heads (lower, prediction, upper) in one kernel, output written directly as [B,3,C,H,W] fp32
    (finallayers/quantile_layer.py:15-17,19-21)."""

    @staticmethod
    def forward(ctx, feat, w_lo, b_lo, w_mid, b_mid, w_hi, b_hi, cdt):
        x = nhwc(feat.detach(), cdt)
        c_out = w_lo.shape[0]
        wf, b_all = _packed_heads((w_lo, w_mid, w_hi), (b_lo, b_mid, b_hi))     # [3C][9][Cmid] fp32
        b, h, w_, _ = x.shape
        out = smallconv_l2s(x, wf, b_all, 3 * c_out)                # [B,3C,H,W]
        ctx.save_for_backward(x, wf)
        ctx.c_out = c_out
        return out.view(b, 3, c_out, h, w_)

    @staticmethod
    def backward(ctx, dout):
        x, wf = ctx.saved_tensors
        c = ctx.c_out
        b, h, w_, cmid = x.shape
        dout = dout.to(F32).contiguous().view(b, 3 * c, h, w_)
        dfeat = None
        if ctx.needs_input_grad[0]:
            dfeat = nchw(smallconv_s2l(dout, wf, None, None, cmid, x.dtype, flip=True))
        dw, db = smallconv_wgrad(dout, x, l_major=False, want_bias=True)
        dw = dw.view(3, c, cmid, 3, 3)
        db = db.view(3, c)
        return dfeat, dw[0], db[0], dw[1], db[1], dw[2], db[2], None


class Heads(torch.autograd.Function):
    """K 3x3 heads (K = 2 or 3) in one kernel, output [B,K,C,H,W] fp32, with an optional activation on head 1:
    'relu' (GaussianRegressionLayer's variance, gaussian_layer.py:15-17) or 'abs' (ResidualMagnitude*Layer's magnitude,
    residual_magnitude_layer.py:15-17).  The activation and its gradient mask act on one small fp32 plane."""

    @staticmethod
    def forward(ctx, feat, cdt, act, *wb):
        x = nhwc(feat.detach(), cdt)
        k = len(wb) // 2
        ws, bs = wb[0::2], wb[1::2]
        c_out = ws[0].shape[0]
        wf, b_all = _packed_heads(ws, bs)                            # [K*C][9][Cmid] fp32
        b, h, w_, _ = x.shape
        out = smallconv_l2s(x, wf, b_all, k * c_out).view(b, k, c_out, h, w_)
        pre = torch.empty(0)
        if act is not None:
            if act not in _HEAD_ACT:
                raise ValueError(f"unknown head activation {act!r}")
            p = c_out * h * w_
            pre = torch.empty((b, p), dtype=F32, device=out.device)
            check(lib.im2im_head_activation_fwd(dptr(out), dptr(pre), b, p, k * p, p, _HEAD_ACT[act], stream_ptr(out.device)),
                  "im2im_head_activation_fwd")
        ctx.save_for_backward(x, wf, pre)
        ctx.cfg = (c_out, k, act)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wf, pre = ctx.saved_tensors
        c, k, act = ctx.cfg
        b, h, w_, cmid = x.shape
        dout = dout.to(F32).contiguous()
        if act is not None:
            dout = dout.clone()                    # the incoming gradient tensor belongs to autograd
            p = c * h * w_
            check(lib.im2im_head_activation_bwd(dptr(dout), dptr(pre), b, p, k * p, p, _HEAD_ACT[act], stream_ptr(dout.device)),
                  "im2im_head_activation_bwd")
        dout = dout.view(b, k * c, h, w_)
        dfeat = None
        if ctx.needs_input_grad[0]:
            dfeat = nchw(smallconv_s2l(dout, wf, None, None, cmid, x.dtype, flip=True))
        dw, db = smallconv_wgrad(dout, x, l_major=False, want_bias=True)
        dw = dw.view(k, c, cmid, 3, 3)
        db = db.view(k, c)
        grads = []
        for i in range(k):
            grads += [dw[i], db[i]]
        return (dfeat, None, None) + tuple(grads)


class SoftmaxHead(torch.autograd.Function):
    """SoftmaxLayer's 3x3 conv to num_softmax class logits (softmax_layer.py:11-14) on the MFMA conv kernel: the class
    axis is zero-padded to a multiple of 32 channels and the result stays NHWC [B,H,W,S] in the compute dtype (the
    [B,K,1,H,W] tensor the reference returns is a strided view of it, made by the caller)."""

    @staticmethod
    def forward(ctx, feat, cdt, weight, bias):
        x = nhwc(feat.detach(), cdt)
        k, cmid = weight.shape[0], weight.shape[1]
        s_ch = (k + 31) // 32 * 32
        wpad = torch.zeros((s_ch, cmid, 3, 3), dtype=F32, device=x.device)
        wpad[:k] = weight.detach()
        bpad = torch.zeros((s_ch,), dtype=F32, device=x.device)
        bpad[:k] = bias.detach()
        wf, wd = pack_weight(wpad, cdt)
        y = conv_fwd(x, wf, bpad)
        ctx.save_for_backward(x, wd)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wd = ctx.saved_tensors
        k = ctx.k
        dy = dy.contiguous()
        dfeat = nchw(conv_fwd(dy, wd)) if ctx.needs_input_grad[0] else None
        dw = conv_wgrad(x, dy, 9).view(dy.shape[3], x.shape[3], 3, 3)[:k]
        db = colsum(dy)[:k]
        return dfeat, None, dw, db


def logits_nhwc(pred):
    """(NHWC class-padded logits [B,H,W,S], K) of a softmax-layer output [B,K,1,H,W]: zero-copy when `pred` is this
    package's own SoftmaxLayer output, else one padded copy (plumbing for user-supplied tensors)."""
    base = getattr(pred, "_im2im_nhwc", None)
    k = pred.shape[1]
    if base is not None:
        return base, k
    if pred.dim() != 5 or pred.shape[2] != 1:
        raise ValueError(f"softmax layer output must be [B,K,1,H,W], got {tuple(pred.shape)}")
    b, _, _, h, w_ = pred.shape
    s_ch = (k + 7) // 8 * 8
    dt = pred.dtype if pred.dtype in (F32, BF16) else F32
    buf = torch.zeros((b, h, w_, s_ch), dtype=dt, device=pred.device)
    buf[..., :k] = pred[:, :, 0].permute(0, 2, 3, 1)
    return buf, k


class SoftmaxCE(torch.autograd.Function):
    """softmax_loss_fn (softmax_layer.py:15-25): cross entropy of the class logits against the bucketised target, mean
    over pixels; one reduction kernel forward, one elementwise kernel backward (gradient in the logits' NHWC layout)."""

    @staticmethod
    def forward(ctx, logits, target, k, bounds):
        dev = logits.device
        m = logits.numel() // logits.shape[-1]
        loss = torch.empty((), dtype=F32, device=dev)
        ws = _Scratch.get(lib.im2im_quantile_loss_workspace_bytes(), dev)
        check(lib.im2im_softmax_ce_fwd(dptr(logits), dptr(target), dptr(bounds), m, k, logits.shape[-1], _DT[logits.dtype],
                                       dptr(loss), dptr(ws), stream_ptr(dev)), "im2im_softmax_ce_fwd")
        ctx.save_for_backward(logits, target, bounds)
        ctx.k = k
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, target, bounds = ctx.saved_tensors
        m = logits.numel() // logits.shape[-1]
        gout = gout.to(F32).contiguous()
        d = torch.empty_like(logits)
        check(lib.im2im_softmax_ce_bwd(dptr(logits), dptr(target), dptr(bounds), m, ctx.k, logits.shape[-1], _DT[logits.dtype],
                                       dptr(gout), dptr(d), stream_ptr(logits.device)), "im2im_softmax_ce_bwd")
        return d, None, None, None


def softmax_sets_summary(pred):
    """[B,K,1,H,W] class logits -> [B,3,1,H,W] fp32 planes (lower quantile, prediction, upper quantile): the
    lambda-independent part of softmax_nested_sets_from_output (softmax_layer.py:33-47)."""
    logits, k = logits_nhwc(pred.detach())
    b, h, w_, s_ch = logits.shape
    out = torch.empty((b, 3, 1, h, w_), dtype=F32, device=logits.device)
    check(lib.im2im_softmax_sets_summary(dptr(logits), b, h * w_, k, s_ch, _DT[logits.dtype], dptr(out), stream_ptr(logits.device)),
          "im2im_softmax_sets_summary")
    return out


LOSS_QUANTILE, LOSS_QUANTILE_L1, LOSS_GAUSSIAN, LOSS_RESIDUAL, LOSS_RESIDUAL_L1, LOSS_INN = 0, 1, 2, 3, 4, 5


class UQLossPacked(torch.autograd.Function):
    """the training loss of a final layer on its packed [B,K,C,H,W] output (include/im2im_uq.h IM2IM_LOSS_*): one fused
    reduction forward, one elementwise kernel backward writing d(pred) as one [B,K,C,H,W] tensor."""

    @staticmethod
    def forward(ctx, pred, target, kind, q_lo, q_hi, w0, w1, w2):
        n, k = pred.shape[0], pred.shape[1]
        p = pred[0, 0].numel()
        dev = pred.device
        loss = torch.empty((), dtype=F32, device=dev)
        ws = _Scratch.get(lib.im2im_quantile_loss_workspace_bytes(), dev)
        base = pred.data_ptr()
        check(lib.im2im_uq_loss_fwd(kind, base, base + 4 * p, (base + 8 * p) if k == 3 else None, dptr(target), n, p, k * p,
                                    q_lo, q_hi, w0, w1, w2, dptr(loss), dptr(ws), stream_ptr(dev)), "im2im_uq_loss_fwd")
        ctx.save_for_backward(pred, target)
        ctx.cfg = (n, k, p, kind, q_lo, q_hi, w0, w1, w2)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        n, k, p, kind, q_lo, q_hi, w0, w1, w2 = ctx.cfg
        gout = gout.to(F32).contiguous()
        d = torch.empty_like(pred)
        base, dbase = pred.data_ptr(), d.data_ptr()
        check(lib.im2im_uq_loss_bwd(kind, base, base + 4 * p, (base + 8 * p) if k == 3 else None, dptr(target), n, p, k * p,
                                    q_lo, q_hi, w0, w1, w2, dptr(gout), dbase, dbase + 4 * p, (dbase + 8 * p) if k == 3 else None,
                                    k * p, stream_ptr(pred.device)), "im2im_uq_loss_bwd")
        return d, None, None, None, None, None, None, None


class QuantileLoss(torch.autograd.Function):
    """w_lo*pinball(q_lo) + w_hi*pinball(q_hi) + w_mse*MSE, each mean-reduced, in one fused reduction."""

    @staticmethod
    def forward(ctx, lo, mid, hi, target, stride, n, p, q_lo, q_hi, w_lo, w_hi, w_mse):
        dev = target.device
        loss = torch.empty((), dtype=F32, device=dev)
        ws = _Scratch.get(lib.im2im_quantile_loss_workspace_bytes(), dev)
        check(lib.im2im_quantile_loss_fwd(dptr(lo), dptr(mid), dptr(hi), dptr(target), n, p, stride, q_lo, q_hi, w_lo, w_hi, w_mse,
                                          dptr(loss), dptr(ws), stream_ptr(dev)), "im2im_quantile_loss_fwd")
        ctx.save_for_backward(lo, mid, hi, target)
        ctx.cfg = (stride, n, p, q_lo, q_hi, w_lo, w_hi, w_mse)
        return loss

    @staticmethod
    def backward(ctx, gout):
        lo, mid, hi, target = ctx.saved_tensors
        stride, n, p, q_lo, q_hi, w_lo, w_hi, w_mse = ctx.cfg
        gout = gout.to(F32).contiguous()
        grads = [None, None, None]
        ptrs = []
        for i in range(3):
            if ctx.needs_input_grad[i]:
                grads[i] = torch.empty((n, p), dtype=F32, device=target.device)
            ptrs.append(dptr(grads[i]))
        check(lib.im2im_quantile_loss_bwd(dptr(lo), dptr(mid), dptr(hi), dptr(target), n, p, stride, q_lo, q_hi, w_lo, w_hi, w_mse,
                                          dptr(gout), ptrs[0], ptrs[1], ptrs[2], p, stream_ptr(target.device)),
              "im2im_quantile_loss_bwd")
        for i, t in enumerate((lo, mid, hi)):
            if grads[i] is not None:
                grads[i] = grads[i].view(t.shape)
        return grads[0], grads[1], grads[2], None, None, None, None, None, None, None, None, None


class QuantileLossPacked(torch.autograd.Function):
    """same loss on the packed [B,3,C,H,W] model output (gradient written in place into one [B,3,C,H,W])."""

    @staticmethod
    def forward(ctx, pred, target, q_lo, q_hi, w_lo, w_hi, w_mse):
        n = pred.shape[0]
        p = pred[0, 0].numel()
        dev = pred.device
        loss = torch.empty((), dtype=F32, device=dev)
        ws = _Scratch.get(lib.im2im_quantile_loss_workspace_bytes(), dev)
        base = pred.data_ptr()
        check(lib.im2im_quantile_loss_fwd(base, base + 4 * p, base + 8 * p, dptr(target), n, p, 3 * p, q_lo, q_hi, w_lo, w_hi, w_mse,
                                          dptr(loss), dptr(ws), stream_ptr(dev)), "im2im_quantile_loss_fwd")
        ctx.save_for_backward(pred, target)
        ctx.cfg = (n, p, q_lo, q_hi, w_lo, w_hi, w_mse)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        n, p, q_lo, q_hi, w_lo, w_hi, w_mse = ctx.cfg
        gout = gout.to(F32).contiguous()
        d = torch.empty_like(pred)
        base, dbase = pred.data_ptr(), d.data_ptr()
        check(lib.im2im_quantile_loss_bwd(base, base + 4 * p, base + 8 * p, dptr(target), n, p, 3 * p, q_lo, q_hi, w_lo, w_hi, w_mse,
                                          dptr(gout), dbase, dbase + 4 * p, dbase + 8 * p, 3 * p, stream_ptr(pred.device)),
              "im2im_quantile_loss_bwd")
        return d, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------- optimizer
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics (defaults betas=(0.9,0.999), eps=1e-8, no weight decay) in one
    multi-tensor HIP launch per 24 tensors.  Drop-in for `optim.Adam(net.parameters(), lr=lr)` at train.py:120.

    capturable=True (torch.optim.Adam's flag of the same name): the step count that sets the bias correction lives on the
    device and is advanced by the update itself (im2im_adam_step_dev), so `step()` may be captured in a HIP graph and replayed
    (core/scripts/train.py GraphedStep); a replay is followed by `advance(params)`, the host-side bookkeeping of that step.
    `state[p]["step"]` stays a host int either way (what state_dict() holds)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._ctrs = {}            # (param group, device index, steps taken so far) -> (int64[1] device counter, float32[2] coefficients)
        self._ctr_allocs = 0       # counters created so far (GraphedStep: none may be created inside a capture)

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("capturable", False)
        self.__dict__.setdefault("_ctrs", {})
        self.__dict__.setdefault("_ctr_allocs", 0)

    def hyper_key(self):
        """what a captured step() froze by value (im2im_adam_step_dev takes lr, betas, eps as scalars): a graph holder compares
        this between replays and re-captures when an LR scheduler / a manual edit of `group['lr']` changed it."""
        return tuple((float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g in self.param_groups)

    def _counter(self, gi, device, taken):
        # keyed per PARAM GROUP: two groups on one device at the same step count each own a counter (sharing the key, the second
        # group's lookup missed, allocated a fresh one and dropped the first -- which a captured graph could still be writing)
        key = (int(gi), torch.device(device).index, int(taken))
        ctr = self._ctrs.pop(key, None)
        if ctr is None:
            if torch.cuda.is_current_stream_capturing():
                # a torch.full captured here would reset the step count at every replay
                raise _lib.Im2ImError("FusedAdam(capturable): a step counter would be created inside a graph capture; "
                                      "run one eager step() with the same parameters first")
            ctr = (torch.full((1,), int(taken), dtype=torch.int64, device=device), torch.zeros(2, dtype=F32, device=device))
            self._ctr_allocs += 1
        self._ctrs[(key[0], key[1], key[2] + 1)] = ctr        # where the next step will look for it
        return ctr

    def advance(self, params):
        """host-side bookkeeping of ONE optimizer step that a replayed HIP graph performed on the device: step counts, the
        counter table, and the parameters' version counters (the packed-weight caches key on them)."""
        group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        moved = set()
        for p in params:
            st = self.state.get(p)
            if st:
                moved.add((group_of.get(id(p), 0), p.device.index, int(st["step"])))
                st["step"] += 1
        for key in moved:
            ctr = self._ctrs.pop(key, None)
            if ctr is not None:
                self._ctrs[(key[0], key[1], key[2] + 1)] = ctr
        touched(*params)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        join_side_streams()                       # weight gradients computed on the second stream (no-op when none are pending)
        for gi, group in enumerate(self.param_groups):
            by_step = {}                      # bias correction is per parameter (torch.optim.Adam): one launch per distinct step count
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != F32 or not p.is_cuda:
                    raise _lib.Im2ImError("FusedAdam: parameters must be fp32 tensors on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad
                if g.dtype != F32 or not g.is_contiguous():
                    g = g.to(F32).contiguous()
                if not p.is_contiguous():
                    raise _lib.Im2ImError("FusedAdam: non-contiguous parameter")
                by_step.setdefault((p.device, int(st["step"])), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            for (dev, step), items in by_step.items():
                n = len(items)
                arr = ctypes.c_void_p * n
                sizes = (ctypes.c_int64 * n)(*[it[0].numel() for it in items])
                ptrs = (arr(*[it[0].data_ptr() for it in items]), arr(*[it[1].data_ptr() for it in items]),
                        arr(*[it[2].data_ptr() for it in items]), arr(*[it[3].data_ptr() for it in items]))
                if self.capturable:
                    ctr, coef = self._counter(gi, dev, step - 1)
                    check(lib.im2im_adam_step_dev(n, *ptrs, sizes, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                  dptr(ctr), dptr(coef), stream_ptr(dev)), "im2im_adam_step_dev")
                else:
                    check(lib.im2im_adam_step(n, *ptrs, sizes, float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(step),
                                              stream_ptr(dev)), "im2im_adam_step")
                touched(*(it[0] for it in items))
        return loss
