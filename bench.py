#!/usr/bin/env python3
"""Benchmark of the im2im-uq hot path on MI355X: quantile-regression UNet training + RCPS calibration.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched under torch.distributed.run)

Default workload = BASELINE.json configs[1] (configs[2] when N > 1): fastMRI-shaped synthetic data, 320x320, n_in = 1, the
reference's 4-level UNet (17.27 M parameters) + quantile head, bf16 compute mode, random-init weights, inputs resident
in HBM.  `--config` selects the other BASELINE configs (see CONFIGS below); they are parity-test cases, not the bench line.
  * a train "step" = forward + fused quantile loss + backward + (N > 1: bucketed RCCL all-reduce of the 17.27 M
    gradients, launched from backward hooks) + fused Adam on a per-GPU batch         -> `value` = train imgs/s (whole job)
  * the calibration leg (`calib`) = calibrate_model on the calibration split (3,474 images at N = 1, 3474/N per GPU):
    eval forward, ONE pass of the scoring kernel for all 1000 lambdas, (N > 1: all-gather of the loss-table rows),
    host Hoeffding-Bentkus scan that stops mid-grid; plus the scoring kernel alone on that set (5.7 GB >> Infinity Cache).
  * `fp32` = the same train step in the parity mode (exact-fp32 MFMA), so the reference-precision number exists
    beside the bf16 one.
`--scaling weak` (default) keeps per-GPU work fixed as N grows; `--scaling strong` splits the reference's global batch
of 78 (and the 3,474 calibration images) over the ranks.  One JSON line is printed by rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3           # v_mfma_f32_32x32x2_f32 = the fp32 vector rate
PEAK_FP8_TFLOPS = 5000.0           # dense MX-scaled fp8 MFMA peak (K = 64 / 128 forms), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MEASURED_BF16_RANDOM_TFLOPS = 1981.0   # register-only v_mfma_f32_32x32x16_bf16 loop on random operands, this part, profiles/r01_hwprobe.txt

PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=1000, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              dataset="fastmri-synthetic", lr=1e-4, input_normalization="standard", output_normalization="min-max")

# BASELINE.json configs -> workloads.  batch = per-GPU batch in weak scaling / global batch in strong scaling.
CONFIGS = {
    "fastmri": dict(label="fastMRI knee singlecoil 320x320, 4-level UNet (BASELINE configs[1]; configs[2] when n_gpus > 1)",
                    size=320, n_in=1, depth=4, batch=78, calib_total=3474, num_lambdas=1000, lam=(0.0, 6.0), dtype="bf16"),
    "denoise32": dict(label="32x32 synthetic Gaussian-denoise, 2-level UNet (BASELINE configs[0])",
                      size=32, n_in=1, depth=2, batch=32, calib_total=64, num_lambdas=50, lam=(0.0, 6.0), dtype="bf16"),
    "temca1024": dict(label="TEMCA2-shaped 1024x1024 tiles, 5-level (deeper) UNet (BASELINE configs[3])",
                      size=1024, n_in=1, depth=5, batch=4, calib_total=256, num_lambdas=100, lam=(7.0, 10.0), dtype="bf16"),
    "bsbcm512": dict(label="BSBCM-shaped 512x512, 2 input channels, 4-level UNet (BASELINE configs[4])",
                     size=512, n_in=2, depth=4, batch=16, calib_total=256, num_lambdas=2000, lam=(0.0, 6.0), dtype="fp8"),
}


def conv_flops_per_image(hw, n_in, depth, base=64, n_mid=32, heads=3):
    """algorithmic conv FLOPs (2 per MAC) of one image: (forward, forward + backward).  Reproduces SURVEY 8(d)'s
    measured 125.285 / 375.738 GFLOP at 320x320, n_in = 1, depth 4 (the first conv has no data-gradient)."""
    from im2im_uq_amd.core.models.trunks.unet import unet_plan
    fwd = 0.0
    first = 0.0
    for name, kind, cin, cout in unet_plan(n_in, depth, base):
        if kind == "inc":
            lv, mid = 0, cout
        elif kind == "down":
            lv, mid = int(name[4:]), cout
        else:
            lv, mid = depth - int(name[2:]), cin // 2
        px = (hw >> lv) ** 2
        f1, f2 = 2.0 * 9 * cin * mid * px, 2.0 * 9 * mid * cout * px
        fwd += f1 + f2
        if kind == "inc":
            first = f1
    fwd += 2.0 * base * n_mid * hw * hw + heads * 2.0 * 9 * n_mid * hw * hw
    return fwd, 3.0 * fwd - first


def cpu_baseline(hw, legs=((4, 5), (8, 3)), n_cal_e2e=48):
    """The CPU leg: the oracle (oracle/, the repo's PyTorch-CPU restatement of the reference path, kind "port") timed
    on this box's host cores on a bounded sample of the same workload (train at B = 4 and 8, the
    per-lambda scoring loop, and calibrate end to end).  Baseline only -- never the target."""
    from oracle import calibration as oc
    from oracle import model as om
    cores = min(os.cpu_count() or 1, 32)      # torch-CPU conv peaks at ~32 threads on the 256-core host (profiles/r02_cpu_threads_probe.txt)
    torch.set_num_threads(cores)
    st = {k: (torch.randn(s) * 0.05 if len(s) == 4 else torch.ones(s) if k.endswith(("weight", "running_var")) else torch.zeros(s))
          for k, s in om.state_spec(1, 1)}
    for k in st:
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
    train = []
    for batch, steps in legs:
        x = torch.randn(batch, 1, hw, hw)
        y = torch.rand(batch, 1, hw, hw)
        om.train_steps(st, [(x, y)], PARAMS, lr=1e-4)                       # warm-up step
        t0 = time.perf_counter()
        om.train_steps(st, [(x, y)] * steps, PARAMS, lr=1e-4)
        dt = time.perf_counter() - t0
        train.append({"batch": batch, "steps": steps, "imgs_per_s": batch * steps / dt, "seconds": dt})
    best = max(train, key=lambda r: r["imgs_per_s"])
    # calibration inner loop on pre-materialised outputs: per-lambda batched-64 loop as the reference runs it
    n_cal, n_lam = 256, 100
    out, lab = oc.synth_outputs(n_cal, 1, hw, hw, seed=0)
    lambdas = torch.linspace(0, 6, n_lam)
    t0 = time.perf_counter()
    for lam in lambdas:
        oc.losses_at(out, lab, lam)
    dt_cal = time.perf_counter() - t0
    per_img_lambda = dt_cal / (n_cal * n_lam)
    # calibrate end to end as the reference does it: eval forward over the set, then the per-lambda loop from the top of
    # the grid down to where the scan stops (here ~550 of 1000 lambdas, the same stop point as the GPU leg's data)
    xe = torch.randn(n_cal_e2e, 1, hw, hw)
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(0, n_cal_e2e, 16):
            om.model_forward(xe[s:s + 16], st, training=False)
    dt_fwd = time.perf_counter() - t0
    visited = 550
    e2e_per_img = dt_fwd / n_cal_e2e + visited * per_img_lambda
    return {
        "value": best["imgs_per_s"], "unit": "train imgs/s", "cores": cores, "kind": "port",
        "sample": "; ".join(f"{r['steps']} Adam steps at batch {r['batch']} = {r['imgs_per_s']:.2f} img/s ({r['seconds']:.1f} s)" for r in train)
                  + f"; {hw}x{hw}, fp32, torch-CPU {torch.get_num_threads()} threads",
        "train_legs": train,
        "calib_scoring": {"value": 1.0 / per_img_lambda, "unit": "image*lambda/s",
                          "imgs_per_s_at_1000_lambdas": 1.0 / per_img_lambda / 1000.0,
                          "sample": f"{n_cal} images x {n_lam} lambdas ({dt_cal:.2f} s); the reference re-reads 16 B/px "
                                    f"per lambda, so 1000 lambdas cost 1000x one"},
        "calib_end_to_end": {"value": 1.0 / e2e_per_img, "unit": "calib imgs/s",
                             "sample": f"eval forward of {n_cal_e2e} images ({dt_fwd:.1f} s) + {visited} visited lambdas x the measured "
                                       f"per-lambda scoring cost (extrapolated from {n_lam} lambdas on {n_cal} images)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="fastmri", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU training batch in weak scaling (default: the config's; 78 = "
                    "the reference's fastMRI batch_size); the GLOBAL batch with --scaling strong")
    ap.add_argument("--calib-images", type=int, default=None, help="calibration images per GPU (default: the config's total / N)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32", "fp8"],
                    help="bf16 (default) | fp32 (parity mode) | fp8 (bf16 storage, e4m3 operands in the forward 3x3 convs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 (parity mode) companion number")
    ap.add_argument("--legs", default="train,calib", help="which legs to run (profiling: --legs train / --legs calib)")
    ap.add_argument("--uncertainty-type", default="quantiles",
                    choices=["quantiles", "quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "softmax"],
                    help="final layer (the headline metric is 'quantiles'; the others are the SURVEY 8f rank-1 rows)")
    args = ap.parse_args()
    conf = dict(CONFIGS[args.config])
    custom = []
    for key, val in (("size", args.size), ("depth", args.depth), ("dtype", args.dtype)):
        if val is not None and val != conf[key]:
            conf[key] = val
            custom.append(f"{key}={val}")
    if args.batch is not None:
        conf["batch"] = args.batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # IM2IM_DIST_BACKEND=gloo lets the N > 1 code path be exercised with several ranks sharing ONE GPU (RCCL refuses
    # duplicate devices); the measured configuration is always nccl (= RCCL), one rank per GPU
    backend = os.environ.get("IM2IM_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from im2im_uq_amd import hip_ops, nn_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import GlobalBatchSampler, GradSync, broadcast_module_state
    from torch.utils.data import TensorDataset

    hw, n_in, depth = conf["size"], conf["n_in"], conf["depth"]
    strong = args.scaling == "strong"
    if strong:
        lo, hi = GlobalBatchSampler.share(conf["batch"], rank, world)
        B, global_batch = hi - lo, conf["batch"]
    else:
        B, global_batch = conf["batch"], conf["batch"] * world
    if B < 1:
        raise SystemExit("strong scaling: fewer images in the global batch than ranks")
    lo, hi = GlobalBatchSampler.share(conf["calib_total"], rank, world)
    M = args.calib_images if args.calib_images is not None else (hi - lo)
    fwd_flop, train_flop = conv_flops_per_image(hw, n_in, depth)

    nn_ops.set_compute_dtype(conf["dtype"])
    torch.manual_seed(0)                                                  # same init on every rank (and broadcast below)
    cfg = dict(PARAMS, device=str(dev), batch_size=B, uncertainty_type=args.uncertainty_type, num_lambdas=conf["num_lambdas"],
               minimum_lambda=conf["lam"][0], maximum_lambda=conf["lam"][1],
               num_softmax=50, minimum_lambda_softmax=0, maximum_lambda_softmax=1.2)      # fastmri_test/config.yml
    two_plane = args.uncertainty_type in ("gaussian", "residual_magnitude", "residual_magnitude_l1")
    form = {"gaussian": hip_ops.SETS_SQRT, "residual_magnitude": hip_ops.SETS_SCALE,
            "residual_magnitude_l1": hip_ops.SETS_SCALE, "softmax": hip_ops.SETS_SOFTMAX}.get(args.uncertainty_type, hip_ops.SETS_QUANTILE)
    calib_bytes_per_img = (12 if two_plane else 16) * hw * hw            # 2 or 3 fp32 output planes + the label, read once
    model = add_uncertainty(UNet(n_in, 1, depth=depth), cfg).to(dev)
    broadcast_module_state(model)
    opt = nn_ops.FusedAdam(model.parameters(), lr=cfg["lr"])
    sync = GradSync(model.parameters()) if world > 1 else None
    loss_weight = B / global_batch if strong else 1.0 / world            # summed over ranks = the global-batch mean loss
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, n_in, hw, hw, device=dev, generator=g)            # input_normalization: standard
    y = torch.rand(B, 1, hw, hw, device=dev, generator=g)                # output_normalization: min-max

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def train_step():
        pred = model(x)
        loss = model.loss_fn(pred, y)
        if sync is None:
            opt.zero_grad()
            loss.backward()
        else:
            sync.zero_grad()
            (loss * loss_weight).backward()
            sync.finish()
        opt.step()
        return loss

    host_dt = [0.0]

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        host_dt[0] = time.perf_counter() - t0        # time the host needed to ENQUEUE the steps (launch-bound when ~ the total)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    legs = set(args.legs.split(","))
    peak = PEAK_FP32_TFLOPS if conf["dtype"] == "fp32" else PEAK_BF16_TFLOPS
    # ---------------------------------------------------------------- train leg
    model.train()
    host_train = None
    if "train" in legs:
        dt_train = timed(train_step, args.steps, args.warmup)
        host_train = host_dt[0] / args.steps * 1e3
    else:
        dt_train = float("nan")
        args.no_roofline = True
    train_ips = global_batch * args.steps / dt_train

    # ---------------------------------------------------------------- roofline leg (HIP events per conv launch)
    roof = roof_w = per_kernel = roof_dgrad = None
    if not args.no_roofline:
        # per-launch durations of the conv kernels, HIP events on the launch stream.  The weight gradients are put back on the
        # main stream for these two steps: a kernel's roofline is about the kernel alone, not about what it shares the chip with
        side_stream_was, pipeline_was = nn_ops.WGRAD_SIDE_STREAM, nn_ops.BWD_PIPELINE
        nn_ops.WGRAD_SIDE_STREAM = nn_ops.BWD_PIPELINE = False
        nn_ops.TIMER = nn_ops.KernelTimer()
        for _ in range(2):
            train_step()
        rows = nn_ops.TIMER.collect()
        nn_ops.TIMER = None
        nn_ops.WGRAD_SIDE_STREAM, nn_ops.BWD_PIPELINE = side_stream_was, pipeline_was
        per_kernel = {k: {"launches": n, "avg_ms": t / n, "tflops": f / t / 1e9} for k, (n, f, t) in sorted(rows.items())}
        ig = [(n, f, t) for k, (n, f, t) in rows.items() if k.startswith("conv_igemm")]
        wg = [(n, f, t) for k, (n, f, t) in rows.items() if k.startswith("conv_wgrad")]

        def agg(v):
            n, f, t = sum(a for a, _, _ in v), sum(b for _, b, _ in v), sum(c for _, _, c in v)
            return {"bound": "mfma", "achieved": f / t / 1e9, "peak": peak, "unit": "TFLOP/s", "frac": f / t / 1e9 / peak,
                    "traffic": None, "launches_per_step": n // 2, "avg_launch_ms": t / n,
                    "algorithmic_gflop_per_launch": f / n / 1e9}
        roof = dict(agg(ig), kernel="conv_igemm_kernel (forward + data-gradient launches, all tile variants)")
        f8 = [(n, f, t) for k, (n, f, t) in rows.items() if k.startswith("conv_fp8")]
        if f8:                          # fp8 mode: the forward convs run on the block-scaled fp8 MFMA -> priced against ITS peak
            n8, fl8, t8 = sum(a for a, _, _ in f8), sum(b for _, b, _ in f8), sum(c for _, _, c in f8)
            roof_fp8 = {"bound": "mfma", "achieved": fl8 / t8 / 1e9, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                        "frac": fl8 / t8 / 1e9 / PEAK_FP8_TFLOPS, "traffic": None, "launches_per_step": n8 // 2, "avg_launch_ms": t8 / n8,
                        "algorithmic_gflop_per_launch": fl8 / n8 / 1e9, "kernel": "conv_fp8_kernel (forward 3x3 convs, e4m3 operands)"}
            roof = dict(roof, kernel="conv_igemm_kernel (bf16: data-gradient launches + the non-eligible forward convs)")
            roof, roof_dgrad = roof_fp8, roof
        else:
            roof_dgrad = None
        if conf["dtype"] == "bf16":
            # second stated peak: what the MFMA pipe sustains on random (non-zero) operands on this power-limited part
            roof["peak_random_operands"] = MEASURED_BF16_RANDOM_TFLOPS
            roof["frac_of_random_operand_peak"] = roof["achieved"] / MEASURED_BF16_RANDOM_TFLOPS
        try:                            # HBM bytes per launch from the committed PMC passes (same batch and size only)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                rec = json.load(f)["conv_igemm_kernel"]
            if rec["per_gpu_batch"] == B and rec["hw"] == hw and args.uncertainty_type == "quantiles" and args.config == "fastmri":
                roof["traffic"] = rec["traffic_bytes_per_launch"]
        except Exception:  # noqa: BLE001
            pass
        try:
            # counter-based utilisation from the committed PMC passes (busy MFMA cycles / elapsed SIMD cycles at the clock the chip
            # actually ran): a second reading beside `frac`, which divides FLOP/s by the nominal-clock peak
            with open(os.path.join(ROOT, "profiles", "r02_pmc_mfma_busy.json")) as f:
                busy = json.load(f)
            if conf["dtype"] == "bf16" and args.config == "fastmri":
                roof["mfma_busy_cycle_frac_pmc"] = {k: round(v["mfma_busy_frac"], 3) for k, v in busy.items() if isinstance(v, dict) and k.startswith("conv_igemm")}
        except Exception:  # noqa: BLE001
            pass
        roof_w = dict(agg(wg), kernel="conv_wgrad_kernel (+ its split-K reduce)")

    # ---------------------------------------------------------------- fp32 companion (parity mode, exact-fp32 MFMA)
    fp32 = None
    if "train" in legs and conf["dtype"] == "bf16" and not args.no_fp32 and world == 1 and args.config == "fastmri":
        nn_ops.set_compute_dtype("fp32")
        s32 = max(2, args.steps // 5)
        dt32 = timed(train_step, s32, 1)
        nn_ops.set_compute_dtype(conf["dtype"])
        ips32 = global_batch * s32 / dt32
        fp32 = {"value": ips32, "unit": "imgs/s", "steps": s32, "warmup": 1, "ms_per_step": dt32 / s32 * 1e3, "dtype": "fp32",
                "train_tflops": ips32 * train_flop / 1e12, "peak": PEAK_FP32_TFLOPS,
                "frac_of_fp32_peak_whole_step": ips32 * train_flop / 1e12 / PEAK_FP32_TFLOPS,
                "note": "same step in the parity mode (v_mfma_f32_32x32x2_f32, fp32 storage): the reference's own precision"}

    if "calib" not in legs:
        if rank == 0:
            print(json.dumps({"metric": "train imgs/sec (train leg only)", "value": train_ips, "unit": "imgs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_train / args.steps * 1e3,
                              "host_enqueue_ms_per_step": host_train, "dtype": conf["dtype"], "roofline": roof, "roofline_wgrad": roof_w, "fp32": fp32, "per_kernel": per_kernel}))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- calibration leg
    # Labels are built from the model's own eval outputs (untimed) so that the scan stops mid-grid like a trained model's
    # does: y = pred + s*z*(half-width on that side), z ~ N(0,1)  =>  the miss rate at lambda is P(|z| > lambda/s); with
    # s = mid-grid / 1.96 (5 % missed at mid-grid) the Hoeffding-Bentkus bound crosses alpha = 0.1 below the middle of the grid (~55-60 % visited).
    model.eval()
    xc = torch.randn(M, n_in, hw, hw, device=dev, generator=g)
    yc = torch.empty(M, 1, hw, hw, device=dev)
    s_lab = (conf["lam"][0] + conf["lam"][1]) / 2 / 1.96
    if args.uncertainty_type == "quantiles":
        with torch.no_grad():
            for s in range(0, M, 64):
                o = model(xc[s:s + 64])
                z = torch.randn(o[:, 1].shape, device=dev, generator=g)
                up = (o[:, 2] - o[:, 1]).clamp_min(1e-6)
                dn = (o[:, 1] - o[:, 0]).clamp_min(1e-6)
                yc[s:s + 64] = o[:, 1] + s_lab * z * torch.where(z > 0, up, dn)
                del o, z, up, dn
    else:
        yc.copy_(torch.rand(M, 1, hw, hw, device=dev, generator=g))
    ds = TensorDataset(xc, yc)
    ds.im2im_local_shard = True
    ccfg = dict(cfg, batch_size=min(max(conf["batch"], 64), M))   # the reference forwards the calibration set in config batches (calibrate_model.py:118)

    def calib_step():
        with contextlib.redirect_stdout(io.StringIO()):
            calibrate_model(model, ds, ccfg)
    cal_steps = max(1, args.steps // 5)
    dt_cal = timed(calib_step, cal_steps, 1)
    calib_ips = M * world * cal_steps / dt_cal
    lhat = float(model.lhat)
    lam_grid = torch.linspace(conf["lam"][0], conf["lam"][1], conf["num_lambdas"])
    visited = int((lam_grid >= lhat - 1e-9).sum())
    del xc, yc, ds
    torch.cuda.empty_cache()
    # scoring kernel alone on outputs shaped like SURVEY 8(d): lhat lands mid-grid
    pred = torch.rand(M, 1, hw, hw, device=dev, generator=g)
    if two_plane:
        mag = 0.05 * torch.rand_like(pred)
        out3 = torch.stack([pred, mag * mag if args.uncertainty_type == "gaussian" else mag], dim=1).contiguous()
    elif args.uncertainty_type == "softmax":                    # (lower quantile, prediction, upper quantile) bins of 1/50
        q = torch.round(pred * 50) / 50
        out3 = torch.stack([(q - 0.04).clamp(0, 1), q, (q + 0.04).clamp(0, 1)], dim=1).contiguous()
    else:
        out3 = torch.stack([pred - 0.05 * torch.rand_like(pred), pred, pred + 0.05 * torch.rand_like(pred)], dim=1).contiguous()
    lab = pred + 0.05 * torch.randn(pred.shape, device=dev, generator=g)
    L = conf["num_lambdas"]
    lam_eff = lam_grid - (lam_grid[1] - lam_grid[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lam_dev = lam_eff.to(dev)
    from im2im_uq_amd._lib import lib as _abi
    hist = torch.empty((_abi.im2im_rcps_workspace_bytes(M, hw * hw, L) // 4,), dtype=torch.int32, device=dev)
    table = torch.empty((M, L), dtype=torch.float32, device=dev)
    for _ in range(2):
        hip_ops.rcps_loss_table_raw(out3, lab, M, hw * hw, lam_dev, hist, table, None, form)
    reps = 10
    e0.record()
    for _ in range(reps):
        hip_ops.rcps_loss_table_raw(out3, lab, M, hw * hw, lam_dev, hist, table, None, form)
    e1.record()
    torch.cuda.synchronize()
    ms_score = e0.elapsed_time(e1) / reps
    score_gbs = M * calib_bytes_per_img / ms_score / 1e6
    traffic = None                      # HBM bytes per launch from the committed PMC passes (same M and size only)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["rcps_hist_kernel"]
        if rec["images"] == M and rec["hw"] == hw and not two_plane:
            traffic = rec["traffic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    calib = {
        "value": calib_ips, "unit": "calib imgs/s (end-to-end calibrate_model: eval forward + all-lambda scoring + HB scan)",
        "ms_per_step": dt_cal / cal_steps * 1e3, "images_per_gpu": M, "num_lambdas": L,
        "lhat": lhat, "lambdas_visited_by_scan": visited,
        "scoring_only": {"imgs_per_s": M / ms_score * 1e3, "ms": ms_score,
                         "roofline": {"bound": "hbm", "achieved": score_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": score_gbs / PEAK_HBM_GBS, "traffic": traffic,
                                      "kernel": "rcps_hist_kernel (+ memset + suffix)",
                                      "algorithmic_bytes_per_launch": M * calib_bytes_per_img}},
    }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(320)
        if calib and cpu.get("calib_end_to_end"):
            calib["vs_cpu_end_to_end"] = calib_ips / cpu["calib_end_to_end"]["value"]

    if rank == 0:
        what = "quantile regression" if args.uncertainty_type == "quantiles" else args.uncertainty_type
        line = {
            "metric": f"train imgs/sec (+ calib imgs/sec in `calib`), {hw}x{hw} UNet " + what,
            "value": train_ips, "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_train / args.steps * 1e3, "host_enqueue_ms_per_step": host_train, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": conf["dtype"], "data": "synthetic (shaped like the named dataset, random-init weights)",
            "config": {"workload": conf["label"] + (f" [overrides: {', '.join(custom)}]" if custom else "") + f", {args.uncertainty_type}",
                       "per_gpu_batch": B, "global_batch": global_batch, "parallelism": f"dp{world}", "unet_depth": depth,
                       "n_in": n_in, "gflop_per_image_train": train_flop / 1e9,
                       "train_tflops": train_ips * train_flop / 1e12},
            "roofline": roof, "roofline_wgrad": roof_w, "fp32": fp32, "calib": calib, "cpu_baseline": cpu, "per_kernel": per_kernel,
        }
        if roof_dgrad:
            line["roofline_bf16_igemm"] = roof_dgrad
        if cpu:
            line["vs_cpu_train"] = train_ips / cpu["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
