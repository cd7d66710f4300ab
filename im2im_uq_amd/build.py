"""Build libim2im_uq.so (gfx950 HIP kernels + host C++ behind the C ABI in include/im2im_uq.h).

In-tree build with explicit hipcc calls: objects under im2im_uq_amd/build/, the shared library at
im2im_uq_amd/lib/libim2im_uq.so (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles for gfx950 without a GPU.

    python -m im2im_uq_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "lib", "libim2im_uq.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# [r6] compiles a file without any packed fp32 instruction (the host pass prints "not a recognized feature": expected).  Not used by default:
# the BatchNorm kernels of elementwise.hip lose 1.1 % of the step without v_pk_* (37.44 -> 37.90 ms, same box, alternating) -- the unreliable
# op_sel forms (profiles/r06_multiprocess_determinism.txt) are kept out at their sources and by tests/test_abi.py's scan of the built library
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# per-source extra flags.  rcps.hip: the reference's CPU path never fuses mul+add.
SOURCES = {
    "common.cpp": [],
    "hb_bound.cpp": ["-ffp-contract=off"],
    "rcps.hip": ["-ffp-contract=off"],
    "conv_mfma.hip": ["-ffp-contract=off"],   # lazy BatchNorm+ReLU in the operand staging must round exactly like bn_relu_apply
    # (the same lazy transform on the weight gradient's x operand); max-ILP machine scheduling: +1.2 % on the 13 BASELINE weight
    # gradients (949-951 -> 957-964 TF, profiles/r04_ab_experiments.txt section 9) -- and -15 % on conv_mfma.hip, which keeps the default
    "conv_wgrad.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "elementwise.hip": ["-ffp-contract=off"],
    "smallconv.hip": ["-ffp-contract=off"],
    "fastmri.hip": ["-ffp-contract=off"],
    "conv_fp8.hip": ["-ffp-contract=off"],
}


# [r6] measured-and-shelved kernels stay out of the default library: IM2IM_BUILD_EXPERIMENTAL=1 compiles them in (and defines the
# macro of the same name for the dispatchers that route to them); their objects carry an ".exp" tag so the two builds never mix.
#   conv_roll.hip: conv_roll64_kernel [r5], the persistent kernel of the 64-output-channel full-resolution layers -- correct, tested,
#   same time as conv_igemm_kernel on this power-limited part (profiles/r05_ab_experiments.txt section 1), option "conv_roll"
EXPERIMENTAL = os.environ.get("IM2IM_BUILD_EXPERIMENTAL", "0") == "1"
EXPERIMENTAL_SOURCES = {
    "conv_roll.hip": ["-ffp-contract=off"],   # same rounding rule for its lazy BatchNorm+ReLU staging
}
EXPERIMENTAL_USERS = ("conv_mfma.hip", "conv_wgrad.hip", "smallconv.hip")     # translation units that test the macro
#   smallconv.hip: the multiply-add forms 0, 2..5 of smallconv_wgrad_vec_kernel (the record of the v_pk_fma_f32 op_sel bisect, IM2IM_SWG_DBG)


def active_sources() -> dict:
    if not EXPERIMENTAL:
        return dict(SOURCES)
    out = {k: (v + ["-DIM2IM_BUILD_EXPERIMENTAL=1"] if k in EXPERIMENTAL_USERS else v) for k, v in SOURCES.items()}
    out.update({k: v + ["-DIM2IM_BUILD_EXPERIMENTAL=1"] for k, v in EXPERIMENTAL_SOURCES.items()})
    return out


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps_mtime() -> float:
    t = os.path.getmtime(os.path.join(os.path.dirname(PKG), "include", "im2im_uq.h"))
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _compile(src: str, flags, force: bool, hdr_t: float) -> str:
    path = os.path.join(CSRC, src)
    tag = ".exp" if EXPERIMENTAL and (src in EXPERIMENTAL_USERS or src in EXPERIMENTAL_SOURCES) else ""
    obj = os.path.join(OBJ, src + tag + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_t):
        return obj
    cmd = [_hipcc(), f"--offload-arch={ARCH}", *COMMON, *flags, "-c", path, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x"); cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)   # NO_PACKED_FP32 on the host pass
    if err.strip():
        sys.stderr.write(err + "\n")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdr_t = _deps_mtime()
    sources = active_sources()
    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], force, hdr_t), sources.items()))
    # the library remembers which object set it was linked from (a default build after an experimental one must relink)
    stamp = LIB + ".objs"
    want = "\n".join(sorted(os.path.basename(o) for o in objs))
    have = open(stamp).read() if os.path.exists(stamp) else None
    if force or not os.path.exists(LIB) or have != want or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
        if verbose:
            print(f"[im2im_uq_amd.build] linked {LIB}" + (" (with the experimental kernels)" if EXPERIMENTAL else ""))
    elif verbose:
        print(f"[im2im_uq_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
