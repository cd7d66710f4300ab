#!/usr/bin/env python3
"""Benchmark of the im2im-uq hot path on MI355X: quantile-regression UNet training + RCPS calibration.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched under torch.distributed.run)

Workload (BASELINE.json configs[1]): fastMRI-shaped synthetic data, 320x320, n_in = 1, the reference's fixed
4-level UNet (17.27 M parameters) + quantile head, bf16 compute mode, random-init weights, inputs resident in HBM.
  * a train "step" = forward + fused quantile loss + backward + (N > 1: one flat RCCL all-reduce of the 17.27 M
    gradients) + fused Adam on a per-GPU batch of --batch images        -> `value` = train imgs/s (whole job)
  * the calibration leg (`calib`) = calibrate_model on --calib-images images per GPU: eval forward, ONE pass of the
    scoring kernel for all 1000 lambdas, (N > 1: all-gather of the loss-table rows), host Hoeffding-Bentkus scan;
    plus the scoring kernel alone.
Per-GPU work is fixed as N grows ("weak" scaling).  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP_PER_IMG = 125.285e9       # SURVEY.md 8(d), measured on the reference model (convs only, 2 FLOP/MAC)
TRAIN_FLOP_PER_IMG = 375.738e9
CALIB_BYTES_PER_IMG = 16 * 320 * 320
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=1000, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              dataset="fastmri-synthetic", lr=1e-4, input_normalization="standard", output_normalization="min-max")


def cpu_baseline(hw, batch, steps):
    """The CPU leg: the oracle (oracle/, the repo's PyTorch-CPU restatement of the reference path, kind "port") timed
    on this box's host cores on a bounded sample of the same workload.  Baseline only -- never the target."""
    from oracle import calibration as oc
    from oracle import model as om
    cores = min(os.cpu_count() or 1, 32)      # torch-CPU conv peaks at ~32 threads on the 256-core host (tools/cpu_threads_probe.py)
    torch.set_num_threads(cores)
    st = {k: (torch.randn(s) * 0.05 if len(s) == 4 else torch.ones(s) if k.endswith(("weight", "running_var")) else torch.zeros(s))
          for k, s in om.state_spec(1, 1)}
    for k in st:
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
    x = torch.randn(batch, 1, hw, hw)
    y = torch.rand(batch, 1, hw, hw)
    om.train_steps(st, [(x, y)], PARAMS, lr=1e-4)                       # warm-up step
    t0 = time.perf_counter()
    om.train_steps(st, [(x, y)] * steps, PARAMS, lr=1e-4)
    dt_train = time.perf_counter() - t0
    # calibration inner loop on pre-materialised outputs: per-lambda batched-64 loop as the reference runs it
    n_cal, n_lam = 256, 200
    out, lab = oc.synth_outputs(n_cal, 1, hw, hw, seed=0)
    lambdas = torch.linspace(0, 6, n_lam)
    t0 = time.perf_counter()
    for lam in lambdas:
        oc.losses_at(out, lab, lam)
    dt_cal = time.perf_counter() - t0
    return {
        "value": batch * steps / dt_train, "unit": "train imgs/s", "cores": cores, "kind": "port",
        "sample": f"{steps} Adam steps, batch {batch}, {hw}x{hw}, fp32, torch-CPU {torch.get_num_threads()} threads "
                  f"({dt_train:.1f} s)",
        "calib_scoring": {"value": n_cal * n_lam / dt_cal, "unit": "image*lambda/s",
                          "imgs_per_s_at_1000_lambdas": n_cal * n_lam / dt_cal / 1000.0,
                          "sample": f"{n_cal} images x {n_lam} lambdas ({dt_cal:.2f} s); the reference re-reads 16 B/px "
                                    f"per lambda, so 1000 lambdas cost 1000x one"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=78, help="per-GPU training batch (78 = the reference's fastMRI batch_size)")
    ap.add_argument("--calib-images", type=int, default=432, help="per-GPU calibration images (3474 / 8 ~ 434)")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--legs", default="train,calib", help="which legs to run (profiling: --legs train / --legs calib)")
    ap.add_argument("--uncertainty-type", default="quantiles",
                    choices=["quantiles", "quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "softmax"],
                    help="final layer (the headline metric is 'quantiles'; the others are the SURVEY 8f rank-1 rows)")
    args = ap.parse_args()

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
    from im2im_uq_amd.core.scripts.train import allreduce_gradients
    from torch.utils.data import TensorDataset

    nn_ops.set_compute_dtype(args.dtype)
    torch.manual_seed(0)                                                  # same init on every rank
    cfg = dict(PARAMS, device=str(dev), batch_size=args.batch, uncertainty_type=args.uncertainty_type,
               num_softmax=50, minimum_lambda_softmax=0, maximum_lambda_softmax=1.2)      # fastmri_test/config.yml
    two_plane = args.uncertainty_type in ("gaussian", "residual_magnitude", "residual_magnitude_l1")
    form = {"gaussian": hip_ops.SETS_SQRT, "residual_magnitude": hip_ops.SETS_SCALE,
            "residual_magnitude_l1": hip_ops.SETS_SCALE, "softmax": hip_ops.SETS_SOFTMAX}.get(args.uncertainty_type, hip_ops.SETS_QUANTILE)
    calib_bytes_per_img = (12 if two_plane else 16) * 320 * 320       # 2 or 3 fp32 output planes + the label, read once
    model = add_uncertainty(UNet(1, 1), cfg).to(dev)
    opt = nn_ops.FusedAdam(model.parameters(), lr=cfg["lr"])
    params = [p for p in model.parameters() if p.requires_grad]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    hw, B = args.size, args.batch
    x = torch.randn(B, 1, hw, hw, device=dev, generator=g)               # input_normalization: standard
    y = torch.rand(B, 1, hw, hw, device=dev, generator=g)                # output_normalization: min-max

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def train_step():
        pred = model(x)
        loss = model.loss_fn(pred, y)
        opt.zero_grad()
        loss.backward()
        allreduce_gradients(params)
        opt.step()
        return loss

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    legs = set(args.legs.split(","))
    # ---------------------------------------------------------------- train leg
    model.train()
    if "train" in legs:
        dt_train = timed(train_step, args.steps, args.warmup)
    else:
        dt_train = float("nan")
        args.no_roofline = True
    train_ips = B * world * args.steps / dt_train

    # ---------------------------------------------------------------- roofline leg (HIP events per conv launch)
    roof = roof_w = per_kernel = None
    if not args.no_roofline:
        nn_ops.TIMER = nn_ops.KernelTimer()
        for _ in range(2):
            train_step()
        rows = nn_ops.TIMER.collect()
        nn_ops.TIMER = None
        per_kernel = {k: {"launches": n, "avg_ms": t / n, "tflops": f / t / 1e9} for k, (n, f, t) in sorted(rows.items())}
        ig = [(n, f, t) for k, (n, f, t) in rows.items() if k.startswith("conv_igemm")]
        wg = [(n, f, t) for k, (n, f, t) in rows.items() if k.startswith("conv_wgrad")]
        def agg(v):
            n, f, t = sum(a for a, _, _ in v), sum(b for _, b, _ in v), sum(c for _, _, c in v)
            return {"bound": "mfma", "achieved": f / t / 1e9, "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
                    "unit": "TFLOP/s", "frac": f / t / 1e9 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3),
                    "traffic": None, "launches_per_step": n // 2, "avg_launch_ms": t / n,
                    "algorithmic_gflop_per_launch": f / n / 1e9}
        roof = dict(agg(ig), kernel="conv_igemm_kernel (forward + data-gradient launches, all tile variants)")
        try:                            # HBM bytes per launch from the committed PMC passes (same batch and size only)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                rec = json.load(f)["conv_igemm_kernel"]
            if rec["per_gpu_batch"] == B and rec["hw"] == hw and args.uncertainty_type == "quantiles":
                roof["traffic"] = rec["traffic_bytes_per_launch"]
        except Exception:  # noqa: BLE001
            pass
        roof_w = dict(agg(wg), kernel="conv_wgrad_kernel (+ its split-K reduce)")

    # ---------------------------------------------------------------- calibration leg
    if "calib" not in legs:
        if rank == 0:
            print(json.dumps({"metric": "train imgs/sec (train leg only)", "value": train_ips, "unit": "imgs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_train / args.steps * 1e3,
                              "dtype": args.dtype, "roofline": roof, "roofline_wgrad": roof_w, "per_kernel": per_kernel}))
        if world > 1:
            dist.destroy_process_group()
        return
    M = args.calib_images
    model.eval()
    xc = torch.randn(M, 1, hw, hw, device=dev, generator=g)
    yc = torch.rand(M, 1, hw, hw, device=dev, generator=g)
    ds = TensorDataset(xc, yc)
    ds.im2im_local_shard = True
    ccfg = dict(cfg, batch_size=min(64, M))
    import contextlib, io
    def calib_step():
        with contextlib.redirect_stdout(io.StringIO()):
            calibrate_model(model, ds, ccfg)
    dt_cal = timed(calib_step, max(1, args.steps // 5), 1)
    cal_steps = max(1, args.steps // 5)
    calib_ips = M * world * cal_steps / dt_cal
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
    lambdas = torch.linspace(0, 6, 1000)
    lam_eff = lambdas - (lambdas[1] - lambdas[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lam_dev = lam_eff.to(dev)
    from im2im_uq_amd._lib import lib as _abi
    hist = torch.empty((_abi.im2im_rcps_workspace_bytes(M, hw * hw, 1000) // 4,), dtype=torch.int32, device=dev)
    table = torch.empty((M, 1000), dtype=torch.float32, device=dev)
    for _ in range(3):
        hip_ops.rcps_loss_table_raw(out3, lab, M, hw * hw, lam_dev, hist, table, None, form)
    reps = 20
    e0.record()
    for _ in range(reps):
        hip_ops.rcps_loss_table_raw(out3, lab, M, hw * hw, lam_dev, hist, table, None, form)
    e1.record()
    torch.cuda.synchronize()
    ms_score = e0.elapsed_time(e1) / reps
    score_gbs = M * calib_bytes_per_img * (hw * hw) / (320 * 320) / ms_score / 1e6
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
        "ms_per_step": dt_cal / cal_steps * 1e3, "images_per_gpu": M, "num_lambdas": 1000,
        "scoring_only": {"imgs_per_s": M / ms_score * 1e3, "ms": ms_score,
                         "roofline": {"bound": "hbm", "achieved": score_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": score_gbs / PEAK_HBM_GBS, "traffic": traffic,
                                      "kernel": "rcps_hist_kernel (+ memset + suffix)",
                                      "algorithmic_bytes_per_launch": M * calib_bytes_per_img}},
    }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(hw, 4, 8)

    if rank == 0:
        line = {
            "metric": "train imgs/sec (+ calib imgs/sec in `calib`), fastMRI 320x320 UNet "
                      + ("quantile regression" if args.uncertainty_type == "quantiles" else args.uncertainty_type),
            "value": train_ips, "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_train / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (fastMRI-shaped, random-init weights)",
            "config": {"workload": f"fastMRI knee singlecoil {hw}x{hw} UNet {args.uncertainty_type} (BASELINE configs[1])",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "train_tflops": train_ips * TRAIN_FLOP_PER_IMG * (hw * hw) / (320 * 320) / 1e12},
            "roofline": roof, "roofline_wgrad": roof_w, "calib": calib, "cpu_baseline": cpu, "per_kernel": per_kernel,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
