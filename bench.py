#!/usr/bin/env python3
"""Benchmark of the im2im-uq hot path on MI355X: quantile-regression UNet training + RCPS calibration.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs no launcher: without a rendezvous environment (RANK / WORLD_SIZE) the script re-executes itself under
`torch.distributed.run` with N ranks, one per GPU over RCCL (im2im_uq_amd/launch.py) -- as the reference uses every GPU
of the node by itself (core/scripts/train.py:112-115).  Launched under torchrun it is a plain rank.  A world that is not
the one asked for, or fewer GPUs than ranks, is a non-zero exit, never a silent single-GPU run.

Default workload = BASELINE.json configs[1] (configs[2] when N > 1): fastMRI-shaped synthetic data, 320x320, n_in = 1, the
reference's 4-level UNet (17.27 M parameters) + quantile head, bf16 compute mode, random-init weights, inputs resident
in HBM.
  * a train "step" = forward + fused quantile loss + backward + (N > 1: bucketed RCCL all-reduce of the 17.27 M
    gradients, launched from backward hooks) + fused Adam on a per-GPU batch         -> `value` = train imgs/s (whole job)
  * the calibration leg (`calib`) = calibrate_model on the calibration split (3,474 images at N = 1, 3474/N per GPU):
    eval forward, ONE pass of the scoring kernel for all 1000 lambdas, (N > 1: all-gather of the loss-table rows),
    host Hoeffding-Bentkus scan that stops mid-grid; plus the scoring kernel alone on that set (5.7 GB >> Infinity Cache).
  * `fp32` = the same train step in the parity mode (exact-fp32 MFMA), so the reference-precision number exists
    beside the bf16 one.
  * `other_configs` (N = 1 default run): short lines for BASELINE configs[0] / [3] / [4] and the config-faithful per-GPU
    batch of 78/8 (what 8-GPU strong scaling runs); `fastmri_pipeline`: the k-space -> image transform.
  * `strong` (N > 1): the reference's GLOBAL batch of 78 split over the ranks (10,10,10,10,10,10,9,9 on 8), beside the
    weak-scaling `value` (78 per GPU).
  * `host_dataset` (N = 1 default run) [r6]: `train_net` and `calibrate_model` themselves on a pageable HOST TensorDataset -- the
    reference's data contract (train.py:147-149, calibrate_model.py:118-123) -- with the pinned prefetcher and with the reference's
    in-line uploads, beside the HBM-resident figures; `value` itself stays HBM-resident as SURVEY 8(d) prescribes.
  * the scalars the north-star targets hang on are repeated at the top level and at the end of the line (`batch10_ms_per_step`,
    `temca1024_imgs_per_s`, `bsbcm512_fp8_over_bf16`, `calib_imgs_per_s`, `calib_forward_frac`, `host_*`).
`--scaling weak` (default) keeps per-GPU work fixed as N grows; `--scaling strong` makes the split batch the headline.
One JSON line is printed by rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as im2im_uq_amd/__init__.py does (this script imports torch first): kernel arguments in device memory

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3           # v_mfma_f32_32x32x2_f32 = the fp32 vector rate
PEAK_FP8_TFLOPS = 5000.0           # dense MX-scaled fp8 MFMA peak (K = 64 / 128 forms), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
def nn_ops_mode():
    from im2im_uq_amd import nn_ops
    return nn_ops.compute_mode()


LIVE_TRAFFIC = {}                  # kernel -> {"traffic_bytes_per_launch", "launches_profiled"}: PMC passes made by THIS run (live_pmc_traffic)


def live_pmc_traffic(timeout=180):
    """HBM bytes per launch of the two dominant kernels from hardware counters, measured by this run: short re-runs of the train /
    calib leg of this same script under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate
    passes, nothing else enabled, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled for gfx950 as its HBM section says).  Any
    failure (no rocprofv3, a timeout) leaves the entry out and the caller falls back to the committed profiles/pmc_traffic.json."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}
    env = dict(os.environ, TMPDIR="/tmp", IM2IM_WGRAD_STREAM="0")
    common = ["--no-cpu-baseline", "--no-fp32", "--no-extras", "--no-roofline", "--steps", "2", "--warmup", "1"]
    legs = {"conv_igemm_kernel": ["--legs", "train"] + common, "rcps_hist_kernel": ["--legs", "calib"] + common}
    out = {}
    for kern, bargs in legs.items():
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="im2im_pmc_", dir="/tmp")
            try:
                subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                sys.executable, os.path.abspath(__file__)] + bargs, env=env, cwd="/tmp", stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=timeout, check=True)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                with open(files[0]) as f:
                    v = [float(r["Counter_Value"]) for r in csv.DictReader(f) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
                if v:
                    vals[counter] = (sum(v) / len(v), len(v))
            except Exception:  # noqa: BLE001
                pass
            finally:
                shutil.rmtree(d, ignore_errors=True)
        if len(vals) == 2:
            out[kern] = {"traffic_bytes_per_launch": (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0,
                         "launches_profiled": vals["FETCH_SIZE"][1]}
    return out


LIVE_SOURCE = ("measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace and --pmc WRITE_SIZE --kernel-trace passes (separate) over "
               "`bench.py --legs {leg} --steps 2`, 2 x FETCH_SIZE + WRITE_SIZE averaged over {n} launches")
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
    "bsbcm512": dict(label="BSBCM-shaped 512x512, 2 input channels, 4-level UNet (BASELINE configs[4]; dtype fp8 = e4m3 / e5m2 operands in the forward and "
                           "data-gradient 3x3 convs of the layers with >= 128 input channels, and in the weight gradients of those with 64 output "
                           "channels -- elsewhere the bf16 weight-gradient kernel is as fast or faster, IM2IM_FP8_WGRAD=1 forces fp8; everything else bf16)",
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
    # [r5] the sample is BOUNDED in time, not only in size: the host of a GPU box is shared (a run next to three busy pods took 4 min
    # here instead of 40 s), so every piece measures its first unit and shrinks to what ~10 s buy; `sample` says what was run
    t_start = time.perf_counter()
    train = []
    for batch, steps in legs:
        if train and (time.perf_counter() - t_start > 15.0 or train[0]["seconds"] / train[0]["steps"] * (batch / train[0]["batch"]) > 5.0):
            break                                                           # slow host: the first leg already is the sample
        x = torch.randn(batch, 1, hw, hw)
        y = torch.rand(batch, 1, hw, hw)
        t0 = time.perf_counter()
        om.train_steps(st, [(x, y)], PARAMS, lr=1e-4)                       # warm-up step
        t_warm = time.perf_counter() - t0
        steps = max(1, min(steps, int(8.0 / max(t_warm, 1e-3))))
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
    done = 0
    for lam in lambdas:
        oc.losses_at(out, lab, lam)
        done += 1
        if done >= 10 and time.perf_counter() - t0 > 8.0:
            break
    n_lam = done
    dt_cal = time.perf_counter() - t0
    per_img_lambda = dt_cal / (n_cal * n_lam)
    # calibrate end to end as the reference does it: eval forward over the set, then the per-lambda loop from the top of
    # the grid down to where the scan stops (here ~550 of 1000 lambdas, the same stop point as the GPU leg's data)
    xe = torch.randn(n_cal_e2e, 1, hw, hw)
    t0 = time.perf_counter()
    fwd_done = 0
    with torch.no_grad():
        for s in range(0, n_cal_e2e, 16):
            om.model_forward(xe[s:s + 16], st, training=False)
            fwd_done += min(16, n_cal_e2e - s)
            if time.perf_counter() - t0 > 8.0:
                break
    n_cal_e2e = fwd_done
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


class Job:
    """what every leg shares: the process group, the device, and the barrier + max-over-ranks timer of the contract."""

    def __init__(self, dist, rank, world, dev, backend):
        self.dist, self.rank, self.world, self.dev, self.backend = dist, rank, world, dev, backend
        self.host_dt = 0.0

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.host_dt = time.perf_counter() - t0      # time the host needed to ENQUEUE the steps (launch-bound when ~ the total)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class Workload:
    """one BASELINE config on this rank: model, optimizer, gradient exchange, HBM-resident synthetic batch."""

    def __init__(self, job, conf, utype="quantiles", strong=False):
        from im2im_uq_amd import nn_ops
        from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
        from im2im_uq_amd.core.models.trunks.unet import UNet
        from im2im_uq_amd.core.scripts.train import GlobalBatchSampler, GradSync, broadcast_module_state
        self.job, self.conf, self.utype, self.strong = job, conf, utype, strong
        self.hw, self.n_in, self.depth = conf["size"], conf["n_in"], conf["depth"]
        rank, world, dev = job.rank, job.world, job.dev
        if strong:
            lo, hi = GlobalBatchSampler.share(conf["batch"], rank, world)
            self.B, self.global_batch = hi - lo, conf["batch"]
        else:
            self.B, self.global_batch = conf["batch"], conf["batch"] * world
        if self.B < 1:
            raise SystemExit("strong scaling: fewer images in the global batch than ranks")
        self.fwd_flop, self.train_flop = conv_flops_per_image(self.hw, self.n_in, self.depth)
        nn_ops.set_compute_dtype(conf["dtype"])
        torch.manual_seed(0)                                              # same init on every rank (and broadcast below)
        self.cfg = dict(PARAMS, device=str(dev), batch_size=self.B, uncertainty_type=utype, num_lambdas=conf["num_lambdas"],
                        minimum_lambda=conf["lam"][0], maximum_lambda=conf["lam"][1],
                        num_softmax=50, minimum_lambda_softmax=0, maximum_lambda_softmax=1.2)      # fastmri_test/config.yml
        self.model = add_uncertainty(UNet(self.n_in, 1, depth=self.depth), self.cfg).to(dev)
        broadcast_module_state(self.model)
        self.opt = nn_ops.FusedAdam(self.model.parameters(), lr=self.cfg["lr"])
        self.sync = GradSync(self.model.parameters()) if world > 1 else None
        self.loss_weight = self.B / self.global_batch if strong else 1.0 / world   # summed over ranks = the global-batch mean loss
        self.gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.x = torch.randn(self.B, self.n_in, self.hw, self.hw, device=dev, generator=self.gen)   # input_normalization: standard
        self.y = torch.rand(self.B, 1, self.hw, self.hw, device=dev, generator=self.gen)          # output_normalization: min-max
        self.allreduce_bytes = int(self.sync.flat.numel() * 4) if self.sync is not None else 0
        self.graphed = None

    def train_step(self):
        if self.graphed is None:          # the train loop's own rule (GraphedStep.wanted): HIP graph for launch-bound batch shapes
            from im2im_uq_amd.core.scripts.train import GraphedStep
            self.graphed = GraphedStep(self.model, self.opt, self.sync) if GraphedStep.wanted(None, self.y.numel(), self.job.world, nn_ops_mode()) else False
        if self.graphed:
            out = self.graphed.step((self.x,), self.y, self.loss_weight if self.sync is not None else 1.0)
            if out is not None:
                return out
        pred = self.model(self.x)
        loss = self.model.loss_fn(pred, self.y)
        if self.sync is None:
            self.opt.zero_grad()
            loss.backward()
        else:
            self.sync.zero_grad()
            (loss * self.loss_weight).backward()
            self.sync.finish()
        self.opt.step()
        return loss

    def train_leg(self, steps, warmup):
        self.model.train()
        dt = self.job.timed(self.train_step, steps, warmup)
        return {"imgs_per_s": self.global_batch * steps / dt, "ms_per_step": dt / steps * 1e3,
                "host_enqueue_ms_per_step": self.job.host_dt / steps * 1e3}

    def kernel_rows(self, overlapped):
        """per-launch durations of the conv kernels over two extra train steps, HIP events on the stream each kernel is
        launched on.  overlapped=False: the weight gradients are put back on the main stream, so every kernel is timed alone
        on the chip (a kernel's roofline is about the kernel); True: the step as `value` times it, weight gradients on
        their own stream sharing the chip with the main stream's kernels."""
        from im2im_uq_amd import nn_ops
        self.model.train()
        was = nn_ops.WGRAD_SIDE_STREAM, nn_ops.BWD_PIPELINE
        if not overlapped:
            nn_ops.WGRAD_SIDE_STREAM = nn_ops.BWD_PIPELINE = False
        nn_ops.TIMER = nn_ops.KernelTimer()
        graphed, self.graphed = self.graphed, False          # per-launch events need eager launches
        for _ in range(2):
            self.train_step()
        self.graphed = graphed
        rows = nn_ops.TIMER.collect()
        nn_ops.TIMER = None
        nn_ops.WGRAD_SIDE_STREAM, nn_ops.BWD_PIPELINE = was
        return rows

    def release(self):
        from im2im_uq_amd import nn_ops
        nn_ops.join_side_streams()
        torch.cuda.synchronize()
        self.model = self.opt = self.sync = self.x = self.y = None
        torch.cuda.empty_cache()


def host_train_record(job, conf, device_resident_ips, n_batches=14, skip=3):
    """train_net itself (core/scripts/train.py, the reference's loop :141-165) fed from a HOST TensorDataset in pageable memory, as a
    reference user calls it: steady-state imgs/s between HIP events recorded at its optimizer steps, with the pinned prefetcher
    (default) and with the reference's in-line `.to(device)` uploads (IM2IM_PREFETCH=0)."""
    from torch.utils.data import TensorDataset
    from im2im_uq_amd import nn_ops, prefetch
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import train_net
    dev, B, hw, n_in = job.dev, conf["batch"], conf["size"], conf["n_in"]
    nn_ops.set_compute_dtype(conf["dtype"])
    cfg = dict(PARAMS, device=str(dev), batch_size=B, num_lambdas=conf["num_lambdas"], minimum_lambda=conf["lam"][0],
               maximum_lambda=conf["lam"][1], num_validation_images=1)
    g = torch.Generator().manual_seed(4321)
    n = B * n_batches
    xs = torch.randn(n, n_in, hw, hw, generator=g)            # pageable host memory
    ys = torch.rand(n, 1, hw, hw, generator=g)
    ds, val = TensorDataset(xs, ys), TensorDataset(xs[:2].clone(), ys[:2].clone())
    rec = {"images_per_epoch": n, "batch": B, "memory": "pageable host TensorDataset, DataLoader(shuffle=True, num_workers=0) as train_net builds it",
           "timed": f"HIP events at the optimizer steps {skip}..{n_batches - 1} of one train_net epoch"}
    orig = nn_ops.FusedAdam.step
    was = prefetch.ENABLED
    try:
        for key, pf in (("inline_upload", False), ("prefetched", True)):
            prefetch.ENABLED = pf
            events = []

            def step(self, closure=None, _events=events):
                out = orig(self, closure)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                _events.append(e)
                return out
            nn_ops.FusedAdam.step = step
            torch.manual_seed(0)
            model = add_uncertainty(UNet(n_in, 1, depth=conf["depth"]), cfg)
            with contextlib.redirect_stdout(io.StringIO()):
                model = train_net(model, ds, val, dev, 1, B, cfg["lr"], False, None, 10 ** 9, 10 ** 9, config=cfg)
            torch.cuda.synchronize()
            nn_ops.FusedAdam.step = orig
            ms = events[skip].elapsed_time(events[-1])
            rec[key + "_imgs_per_s"] = B * (len(events) - 1 - skip) / ms * 1e3
            rec[key + "_ms_per_step"] = ms / (len(events) - 1 - skip)
            del model
    finally:
        nn_ops.FusedAdam.step = orig
        prefetch.ENABLED = was
    rec["train_imgs_per_s"] = rec["prefetched_imgs_per_s"]
    rec["frac_of_device_resident"] = rec["prefetched_imgs_per_s"] / device_resident_ips
    torch.cuda.empty_cache()
    return rec


def distributed_record(job, wl, backend, steps=10):
    """what lets a reader of the line check that N ranks on N devices really exchanged gradients: every rank's
    device (index, name, uuid / PCI bus id, gathered over the job's own process group), the RCCL version, ONE standalone
    all-reduce of the gradient-sized buffer (ms, algorithm and bus bandwidth: bus = 2(N-1)/N * bytes / time, the number to hold
    against the xGMI links) and how much of the exchange a training step does NOT hide: the same step with and without GradSync."""
    import torch.distributed as tdist
    dev, world = job.dev, job.world
    from im2im_uq_amd import launch
    me = launch.device_identity(dev, job.rank)
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        tdist.all_gather_object(ranks, me)
    rec = {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else None, "world_size": world,
           "allreduce_bytes_per_step": wl.allreduce_bytes, "gpus_visible": torch.cuda.device_count(),
           "under_torch_distributed_run": os.environ.get("TORCHELASTIC_RUN_ID") is not None, "ranks": ranks,
           "distinct_devices": launch.distinct_devices(ranks)}
    if world > 1 and backend == "nccl" and rec["distinct_devices"] != world and launch.identifiable(ranks):      # launch.verify_world already refused this at start-up
        raise SystemExit(f"{world} RCCL ranks on {rec['distinct_devices']} distinct devices")
    try:
        rec["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        rec["rccl_version"] = None
    if world == 1 or wl.sync is None:
        return rec
    # one standalone all-reduce of the flat gradient buffer (a copy: the optimizer's gradients stay untouched)
    buf = torch.zeros_like(wl.sync.flat)
    for _ in range(3):
        tdist.all_reduce(buf)
    job.barrier()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        tdist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
    tdist.all_reduce(dt, op=tdist.ReduceOp.MAX)
    ms = float(dt.item()) * 1e3
    nbytes = buf.numel() * 4
    rec["allreduce_ms"] = ms
    rec["allreduce_algbw_gbs"] = nbytes / ms / 1e6
    rec["allreduce_busbw_gbs"] = 2.0 * (world - 1) / world * nbytes / ms / 1e6
    del buf
    # the same training step with and without the gradient exchange (second: every rank on its own, weights restored afterwards)
    state = [p.detach().clone() for p in wl.model.parameters()]
    with_sync = wl.train_leg(steps, 2)["ms_per_step"]
    sync, wl.sync, graphed, wl.graphed = wl.sync, None, wl.graphed, False
    hooks, sync.hook_handles = sync.hook_handles, []
    for h in hooks:
        h.remove()
    for p in wl.model.parameters():
        p.grad = None
    without = wl.train_leg(steps, 2)["ms_per_step"]
    for i, p in enumerate(sync.params):                      # GradSync back in place
        sync.hook_handles.append(p.register_post_accumulate_grad_hook(sync._make_hook(i)))
    wl.sync, wl.graphed = sync, graphed
    with torch.no_grad():
        for p, q in zip(wl.model.parameters(), state):
            p.copy_(q)
    rec["step_ms_with_grad_exchange"] = with_sync
    rec["step_ms_without_grad_exchange"] = without
    rec["grad_exchange_exposed_ms"] = with_sync - without
    return rec


def _agg(v, peak):
    n, f, t = sum(a for a, _, _ in v), sum(b for _, b, _ in v), sum(c for _, _, c in v)
    return {"bound": "mfma", "achieved": f / t / 1e9, "peak": peak, "unit": "TFLOP/s", "frac": f / t / 1e9 / peak,
            "traffic": None, "launches_per_step": n // 2, "avg_launch_ms": t / n, "algorithmic_gflop_per_launch": f / n / 1e9}


def roofline_leg(wl, config_name, full=True):
    """(roofline, roofline_wgrad, per_kernel, roofline of the bf16 igemm in fp8 mode) of one workload."""
    conf = wl.conf
    peak = PEAK_FP32_TFLOPS if conf["dtype"] == "fp32" else PEAK_BF16_TFLOPS
    rows = wl.kernel_rows(overlapped=False)
    per_kernel = {k: {"launches": n, "avg_ms": t / n, "tflops": f / t / 1e9} for k, (n, f, t) in sorted(rows.items())}
    ig = [v for k, v in rows.items() if k.startswith("conv_igemm") or k.startswith("conv_ws")]
    wg = [v for k, v in rows.items() if k.startswith("conv_wgrad") and not k.startswith("conv_wgrad_fp8")]
    wg8 = [v for k, v in rows.items() if k.startswith("conv_wgrad_fp8")]
    f8 = [v for k, v in rows.items() if k.startswith("conv_fp8")]
    roof = dict(_agg(ig, peak), kernel="conv_igemm_kernel + conv_ws_kernel (forward + data-gradient launches, all tile variants)")
    roof_dgrad = None
    if f8:                          # fp8 mode: the fp8 convs run on the block-scaled MFMA -> priced against ITS peak
        roof_fp8 = dict(_agg(f8, PEAK_FP8_TFLOPS), kernel="conv_fp8_kernel (3x3 convs with e4m3 / e5m2 operands)")
        roof_dgrad = dict(roof, kernel="conv_igemm_kernel (bf16: the launches the fp8 kernels do not cover)")
        roof = roof_fp8
    roof_w = dict(_agg(wg, peak), kernel="conv_wgrad_roll_kernel / conv_wgrad_pipe_kernel (3x3) + conv_wgrad_kernel (1x1) + their split-K reduce") if wg else None
    if wg8 and roof_w is not None:      # fp8 mode: the eligible layers' weight gradients run on the block-scaled MFMA -> priced against ITS peak
        roof_w["fp8_launches"] = dict(_agg(wg8, PEAK_FP8_TFLOPS), kernel="conv_wgrad_fp8_kernel (e5m2 dz x e4m3 input, + its split-K reduce)")
    if not full:
        return roof, roof_w, per_kernel, roof_dgrad
    # second reading: the same kernels inside the step that `value` times (weight gradients on their own stream)
    rows_o = wl.kernel_rows(overlapped=True)
    dom = [v for k, v in rows_o.items() if k.startswith("conv_fp8" if f8 else ("conv_igemm", "conv_ws"))]
    if dom:
        roof["frac_in_timed_step"] = _agg(dom, roof["peak"])["frac"]
    wgo = [v for k, v in rows_o.items() if k.startswith("conv_wgrad")]
    if wgo and roof_w:
        roof_w["frac_in_timed_step"] = _agg(wgo, peak)["frac"]
    if conf["dtype"] == "bf16":
        # second stated peak: what the MFMA pipe sustains on random (non-zero) operands on this power-limited part
        roof["peak_random_operands"] = MEASURED_BF16_RANDOM_TFLOPS
        roof["frac_of_random_operand_peak"] = roof["achieved"] / MEASURED_BF16_RANDOM_TFLOPS
    try:                            # HBM bytes per launch from the committed PMC passes (same batch and size only)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["conv_igemm_kernel"]
        if rec["per_gpu_batch"] == wl.B and rec["hw"] == wl.hw and wl.utype == "quantiles" and config_name == "fastmri":
            roof["traffic"] = rec["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/pmc_traffic.json (replayed: PMC passes of the same command, not measured in this run)"
    except Exception:  # noqa: BLE001
        pass
    if "conv_igemm_kernel" in LIVE_TRAFFIC:
        roof["traffic"] = LIVE_TRAFFIC["conv_igemm_kernel"]["traffic_bytes_per_launch"]
        roof["traffic_source"] = LIVE_SOURCE.format(leg="train", n=LIVE_TRAFFIC["conv_igemm_kernel"]["launches_profiled"])
    try:
        # counter-based utilisation from the committed PMC passes (busy MFMA cycles / elapsed SIMD cycles at the clock the chip
        # actually ran): a second reading beside `frac`, which divides FLOP/s by the nominal-clock peak
        name = next(n for n in ("r06_pmc_mfma_busy.json", "r05_pmc_mfma_busy.json", "r04_pmc_mfma_busy.json", "r02_pmc_mfma_busy.json")
                    if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", name)) as f:
            busy = json.load(f)
        if conf["dtype"] == "bf16" and config_name == "fastmri":
            roof["mfma_busy_cycle_frac_pmc"] = {k: round(v["mfma_busy_frac"], 3) for k, v in busy.items()
                                                if isinstance(v, dict) and k.startswith(("conv_igemm", "conv_ws"))}
            roof["mfma_busy_cycle_frac_pmc"]["source"] = (f"profiles/{name} (replayed, not measured in this run: SQ counter passes over "
                                                          f"tools/bench_conv.py with the round-{int(name[1:3])} kernels)")
    except Exception:  # noqa: BLE001
        pass
    return roof, roof_w, per_kernel, roof_dgrad


def calib_leg(wl, steps, calib_images=None, scoring=True, host_images=0):
    """calibrate_model end to end on this rank's share of the config's calibration split, then the scoring kernel alone.
    host_images > 0: also calibrate_model on a HOST TensorDataset (pageable memory, what a reference user hands over,
    calibrate_model.py:118-123) of the first `host_images` images, with and without the pinned prefetcher, beside the same images
    resident in HBM -> calib["host_dataset"]."""
    from im2im_uq_amd import hip_ops
    from im2im_uq_amd._lib import lib as _abi
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.scripts.train import GlobalBatchSampler
    from torch.utils.data import TensorDataset
    job, conf, dev, g = wl.job, wl.conf, wl.job.dev, wl.gen
    hw, n_in, utype = wl.hw, wl.n_in, wl.utype
    lo, hi = GlobalBatchSampler.share(conf["calib_total"], job.rank, job.world)
    M = calib_images if calib_images is not None else (hi - lo)
    two_plane = utype in ("gaussian", "residual_magnitude", "residual_magnitude_l1")
    form = {"gaussian": hip_ops.SETS_SQRT, "residual_magnitude": hip_ops.SETS_SCALE,
            "residual_magnitude_l1": hip_ops.SETS_SCALE, "softmax": hip_ops.SETS_SOFTMAX}.get(utype, hip_ops.SETS_QUANTILE)
    calib_bytes_per_img = (12 if two_plane else 16) * hw * hw            # 2 or 3 fp32 output planes + the label, read once
    # Labels are built from the model's own eval outputs (untimed) so that the scan stops mid-grid like a trained model's
    # does: y = pred + s*z*(half-width on that side), z ~ N(0,1)  =>  the miss rate at lambda is P(|z| > lambda/s); with
    # s = mid-grid / 1.96 (5 % missed at mid-grid) the Hoeffding-Bentkus bound crosses alpha = 0.1 below the middle of the grid (~55-60 % visited).
    model = wl.model
    model.eval()
    xc = torch.randn(M, n_in, hw, hw, device=dev, generator=g)
    yc = torch.empty(M, 1, hw, hw, device=dev)
    s_lab = (conf["lam"][0] + conf["lam"][1]) / 2 / 1.96
    if utype == "quantiles":
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
    ccfg = dict(wl.cfg, batch_size=min(max(conf["batch"], 64), M))   # the reference forwards the calibration set in config batches (calibrate_model.py:118)

    def calib_step():
        with contextlib.redirect_stdout(io.StringIO()):
            calibrate_model(model, ds, ccfg)
    dt_cal = job.timed(calib_step, steps, 1)
    calib_ips = M * job.world * steps / dt_cal

    # the eval forward alone (what the leg is bound by): conv FLOPs of the forward pass against the MFMA peak of the dtype
    def fwd_step():
        with torch.no_grad():
            for s in range(0, M, ccfg["batch_size"]):
                model(xc[s:s + ccfg["batch_size"]])
    dt_fwd = job.timed(fwd_step, max(1, steps), 1)
    fwd_tflops = M * max(1, steps) * wl.fwd_flop / dt_fwd / 1e12
    fwd_peak = {"fp32": PEAK_FP32_TFLOPS, "fp8": PEAK_FP8_TFLOPS}.get(conf["dtype"], PEAK_BF16_TFLOPS)
    lhat = float(model.lhat)
    lam_grid = torch.linspace(conf["lam"][0], conf["lam"][1], conf["num_lambdas"])
    visited = int((lam_grid >= lhat - 1e-9).sum())
    L = conf["num_lambdas"]
    calib = {"value": calib_ips, "unit": "calib imgs/s (end-to-end calibrate_model: eval forward + all-lambda scoring + HB scan)",
             "ms_per_step": dt_cal / steps * 1e3, "images_per_gpu": M, "num_lambdas": L, "lhat": lhat, "lambdas_visited_by_scan": visited,
             "forward_roofline": {"bound": "mfma", "achieved": fwd_tflops, "peak": fwd_peak, "unit": "TFLOP/s", "frac": fwd_tflops / fwd_peak,
                                  "imgs_per_s_forward_only": M * max(1, steps) / dt_fwd, "share_of_leg": (dt_fwd / max(1, steps)) / (dt_cal / steps),
                                  "gflop_per_image_forward": wl.fwd_flop / 1e9,
                                  "note": "eval forward of this rank's calibration shard alone, same batches as calibrate_model; in fp8 mode only the "
                                          "eligible 3x3 convs run on the fp8 MFMA, the peak is the fp8 one"}}
    if host_images:
        from im2im_uq_amd import prefetch
        mh = min(int(host_images), M)
        xh, yh = xc[:mh].cpu(), yc[:mh].cpu()                  # pageable host tensors
        ds_dev = TensorDataset(xc[:mh], yc[:mh])
        ds_dev.im2im_local_shard = True
        ds_host = TensorDataset(xh, yh)
        ds_host.im2im_local_shard = True

        def run(dataset):
            with contextlib.redirect_stdout(io.StringIO()):
                calibrate_model(model, dataset, ccfg)
            return float(model.lhat)
        rec = {"images": mh, "batch": ccfg["batch_size"], "memory": "pageable host TensorDataset, DataLoader(num_workers=0) order"}
        lh = {}
        was = prefetch.ENABLED
        try:
            for key, dataset, pf in (("device_resident", ds_dev, True), ("host_inline_upload", ds_host, False), ("host_prefetched", ds_host, True)):
                prefetch.ENABLED = pf
                run(dataset)                                    # warm-up (pinned staging ring, allocator)
                job.barrier()
                t0 = time.perf_counter()
                lh[key] = run(dataset)
                torch.cuda.synchronize()
                rec[key + "_imgs_per_s"] = mh / (time.perf_counter() - t0)
        finally:
            prefetch.ENABLED = was
        rec["calib_imgs_per_s"] = rec["host_prefetched_imgs_per_s"]
        rec["frac_of_device_resident"] = rec["host_prefetched_imgs_per_s"] / rec["device_resident_imgs_per_s"]
        rec["lhat_identical"] = len(set(lh.values())) == 1
        calib["host_dataset"] = rec
        del xh, yh, ds_dev, ds_host
    del xc, yc, ds
    torch.cuda.empty_cache()
    if not scoring:
        return calib
    # scoring kernel alone on outputs shaped like SURVEY 8(d): lhat lands mid-grid
    pred = torch.rand(M, 1, hw, hw, device=dev, generator=g)
    if two_plane:
        mag = 0.05 * torch.rand_like(pred)
        out3 = torch.stack([pred, mag * mag if utype == "gaussian" else mag], dim=1).contiguous()
    elif utype == "softmax":                    # (lower quantile, prediction, upper quantile) bins of 1/50
        q = torch.round(pred * 50) / 50
        out3 = torch.stack([(q - 0.04).clamp(0, 1), q, (q + 0.04).clamp(0, 1)], dim=1).contiguous()
    else:
        out3 = torch.stack([pred - 0.05 * torch.rand_like(pred), pred, pred + 0.05 * torch.rand_like(pred)], dim=1).contiguous()
    lab = pred + 0.05 * torch.randn(pred.shape, device=dev, generator=g)
    lam_eff = lam_grid - (lam_grid[1] - lam_grid[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lam_dev = lam_eff.to(dev)
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
    traffic = source = None             # HBM bytes per launch from the committed PMC passes (same M and size only)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["rcps_hist_kernel"]
        if rec["images"] == M and rec["hw"] == hw and not two_plane:
            traffic = rec["traffic_bytes_per_launch"]
            source = "profiles/pmc_traffic.json (replayed: PMC passes of the same command, not measured in this run)"
    except Exception:  # noqa: BLE001
        pass
    if "rcps_hist_kernel" in LIVE_TRAFFIC and not two_plane:
        traffic = LIVE_TRAFFIC["rcps_hist_kernel"]["traffic_bytes_per_launch"]
        source = LIVE_SOURCE.format(leg="calib", n=LIVE_TRAFFIC["rcps_hist_kernel"]["launches_profiled"])
    calib["scoring_only"] = {"imgs_per_s": M / ms_score * 1e3, "ms": ms_score,
                             "roofline": {"bound": "hbm", "achieved": score_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": score_gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": source,
                                          "kernel": "rcps_hist_kernel (+ memset + suffix)",
                                          "algorithmic_bytes_per_launch": M * calib_bytes_per_img}}
    del out3, lab, pred, hist, table
    torch.cuda.empty_cache()
    return calib


def sub_record(job, name, steps=4, warmup=2, dtype=None, batch=None, calib=True):
    """a short line for another BASELINE config in the same run: train img/s, ms/step, conv roofline fractions, calib img/s."""
    conf = dict(CONFIGS[name])
    if dtype:
        conf["dtype"] = dtype
    if batch:
        conf["batch"] = batch
    wl = Workload(job, conf)
    tr = wl.train_leg(steps, warmup)
    roof, roof_w, _, roof_dgrad = roofline_leg(wl, name, full=False)
    rec = {"workload": conf["label"], "dtype": conf["dtype"], "per_gpu_batch": wl.B, "steps": steps, "warmup": warmup,
           "value": tr["imgs_per_s"], "unit": "train imgs/s", "ms_per_step": tr["ms_per_step"],
           "host_enqueue_ms_per_step": tr["host_enqueue_ms_per_step"],
           "hip_graph": bool(wl.graphed),      # the train loop's rule: forward + loss + backward replayed as one HIP graph for launch-bound shapes
           "train_tflops": tr["imgs_per_s"] * wl.train_flop / 1e12,
           "roofline": {k: roof[k] for k in ("achieved", "peak", "frac", "unit", "kernel")},
           "roofline_wgrad": {k: roof_w[k] for k in ("achieved", "peak", "frac", "unit", "fp8_launches") if k in roof_w} if roof_w else None}
    if roof_dgrad:
        rec["roofline_bf16_igemm"] = {k: roof_dgrad[k] for k in ("achieved", "peak", "frac", "unit")}
    if calib:
        c = calib_leg(wl, 1, calib_images=min(conf["calib_total"], 128), scoring=False)
        rec["calib"] = {k: c[k] for k in ("value", "images_per_gpu", "num_lambdas", "lhat", "lambdas_visited_by_scan")}
    wl.release()
    return rec


def fastmri_pipeline_record(job, batches=(16, 64), reps=5):
    """the single-coil sample transform (mask x k-space -> centred inverse DFT -> 320x320 crop -> magnitude ->
    normalise) on 640x368 slices: GPU slices/s, the two GEMMs against the fp32-matrix peak, the reference transform
    (oracle.fastmri.unet_data_transform, torch-CPU fft) on the host cores beside it."""
    from im2im_uq_amd.core.datasets.fastmri import masked_ifft2c_abs
    from oracle import fastmri as ofm
    dev = job.dev
    R, C, crop = 640, 368, (320, 320)
    mask1 = torch.from_numpy(ofm.seeded_mask("equispaced", C, [0.08], [4], (1, 2, 3)))
    # pruned separable DFT as two dense complex contractions: T = X * Wc^T (R x C -> R x 320), I = Wr * T (320 x R x 320)
    flop = 8.0 * (R * C * crop[1] + crop[0] * R * crop[1])
    out = {"slice": f"{R}x{C} complex k-space -> {crop[0]}x{crop[1]} magnitude image", "gflop_per_slice": flop / 1e9, "gpu": []}
    for b in batches:
        ks = torch.randn(b, R, C, 2, device=dev) * 1e-4
        masks = mask1.unsqueeze(0).repeat(b, 1)
        for _ in range(2):
            masked_ifft2c_abs(ks, masks, crop, 0.1, 2.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            masked_ifft2c_abs(ks, masks, crop, 0.1, 2.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        io_bytes = b * (R * C * 8 + crop[0] * crop[1] * 4)
        out["gpu"].append({"batch": b, "ms": ms, "slices_per_s": b / ms * 1e3, "tflops_fp32_matrix": b * flop / ms / 1e9,
                           "frac_of_fp32_matrix_peak": b * flop / ms / 1e9 / PEAK_FP32_TFLOPS,
                           "algorithmic_io_gbs": io_bytes / ms / 1e6, "frac_of_hbm_peak_io": io_bytes / ms / 1e6 / PEAK_HBM_GBS})
        del ks
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    n_cpu = 16
    ksc = ofm.det_kspace(n_cpu, R, C, salt=2)
    ofm.unet_data_transform(ksc[0], mask1, crop)
    t0 = time.perf_counter()
    for i in range(n_cpu):                               # the reference transforms one slice per __getitem__ (num_workers = 0)
        ofm.normalize(ofm.unet_data_transform(ksc[i], mask1, crop), 0.1, 2.0)
    dt = time.perf_counter() - t0
    out["cpu_reference_transform"] = {"slices_per_s": n_cpu / dt, "cores": cores, "kind": "port",
                                      "sample": f"{n_cpu} slices one at a time, torch-CPU pocketfft, {cores} threads"}
    out["vs_cpu"] = max(r["slices_per_s"] for r in out["gpu"]) / (n_cpu / dt)
    return out


_T0 = time.perf_counter()


def _phase(name):
    """wall-clock marks on stderr (the JSON line on stdout stays the only stdout output): where a run's time goes"""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write(f"[bench {time.perf_counter() - _T0:7.1f} s] {name}\n")
        sys.stderr.flush()


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
                    help="bf16 (default) | fp32 (parity mode) | fp8 (bf16 storage, e4m3 / e5m2 operands in the 3x3 convs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 (parity mode) companion number")
    ap.add_argument("--no-extras", action="store_true", help="skip the short sub-records (other BASELINE configs, batch 10, "
                    "fastMRI pipeline; N > 1: the strong-scaling companion)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-run the legs under rocprofv3 --pmc for roofline.traffic "
                    "(the committed profiles/pmc_traffic.json is reported instead)")
    ap.add_argument("--legs", default="train,calib", help="which legs to run (profiling: --legs train / --legs calib)")
    ap.add_argument("--uncertainty-type", default="quantiles",
                    choices=["quantiles", "quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "softmax"],
                    help="final layer (the headline metric is 'quantiles'; the others are the SURVEY 8f rank-1 rows)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    from im2im_uq_amd import launch
    if args.gpus > 1 and not launch.in_rendezvous_env():
        # no launcher around us: become one.  The ranks print the JSON line; this process only relays their exit code.
        sys.exit(launch.spawn_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    dist, rank, world, dev, backend = launch.init_distributed(expected_world=args.gpus)
    job = Job(dist, rank, world, dev, backend)

    conf = dict(CONFIGS[args.config])
    custom = []
    for key, val in (("size", args.size), ("depth", args.depth), ("dtype", args.dtype)):
        if val is not None and val != conf[key]:
            conf[key] = val
            custom.append(f"{key}={val}")
    if args.batch is not None:
        conf["batch"] = args.batch
    default_run = (args.config == "fastmri" and not custom and args.batch is None and args.calib_images is None
                   and args.uncertainty_type == "quantiles" and args.legs == "train,calib" and not args.no_extras)

    if default_run and world == 1 and not args.no_live_pmc:
        _phase("live PMC passes start")
        LIVE_TRAFFIC.update(live_pmc_traffic())             # before this process allocates anything: the passes have the GPU to themselves
        _phase("live PMC passes done")

    from im2im_uq_amd import nn_ops
    wl = Workload(job, conf, args.uncertainty_type, strong=args.scaling == "strong")
    hw, B, global_batch = wl.hw, wl.B, wl.global_batch
    legs = set(args.legs.split(","))

    # ---------------------------------------------------------------- train leg
    tr = {"imgs_per_s": float("nan"), "ms_per_step": float("nan"), "host_enqueue_ms_per_step": None}
    _phase("workload built")
    if "train" in legs:
        tr = wl.train_leg(args.steps, args.warmup)
        _phase("train leg done")
    else:
        args.no_roofline = True
    train_ips = tr["imgs_per_s"]

    # ---------------------------------------------------------------- roofline leg (HIP events per conv launch)
    roof = roof_w = per_kernel = roof_dgrad = None
    if not args.no_roofline:
        roof, roof_w, per_kernel, roof_dgrad = roofline_leg(wl, args.config)

    # ---------------------------------------------------------------- fp32 companion (parity mode, exact-fp32 MFMA)
    fp32 = None
    if "train" in legs and conf["dtype"] == "bf16" and not args.no_fp32 and world == 1 and args.config == "fastmri":
        nn_ops.set_compute_dtype("fp32")
        s32 = max(2, args.steps // 5)
        dt32 = job.timed(wl.train_step, s32, 1)
        nn_ops.set_compute_dtype(conf["dtype"])
        ips32 = global_batch * s32 / dt32
        fp32 = {"value": ips32, "unit": "imgs/s", "steps": s32, "warmup": 1, "ms_per_step": dt32 / s32 * 1e3, "dtype": "fp32",
                "train_tflops": ips32 * wl.train_flop / 1e12, "peak": PEAK_FP32_TFLOPS,
                "frac_of_fp32_peak_whole_step": ips32 * wl.train_flop / 1e12 / PEAK_FP32_TFLOPS,
                "note": "same step in the parity mode (v_mfma_f32_32x32x2_f32, fp32 storage): the reference's own precision"}

    _phase("roofline + fp32 legs done")
    dist_info = distributed_record(job, wl, backend) if "train" in legs else {"world_size": world}

    if "calib" not in legs:
        if rank == 0:
            print(json.dumps({"metric": "train imgs/sec (train leg only)", "value": train_ips, "unit": "imgs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": tr["ms_per_step"],
                              "host_enqueue_ms_per_step": tr["host_enqueue_ms_per_step"], "dtype": conf["dtype"],
                              "distributed": dist_info, "roofline": roof, "roofline_wgrad": roof_w, "fp32": fp32, "per_kernel": per_kernel}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- calibration leg
    want_host = default_run and world == 1
    calib = calib_leg(wl, max(1, args.steps // 5), calib_images=args.calib_images, host_images=(args.calib_images or conf["calib_total"]) if want_host else 0)
    wl.release()
    _phase("calibration leg done")
    host_ds = None
    if want_host:
        # [r6] what a user of train_net / calibrate_model gets from a HOST dataset (the reference's data contract), beside the HBM-resident legs
        try:
            ht = host_train_record(job, conf, train_ips)
            hc = calib.get("host_dataset", {})
            host_ds = {"train_imgs_per_s": ht["train_imgs_per_s"], "calib_imgs_per_s": hc.get("calib_imgs_per_s"),
                       "frac_of_device_resident": {"train": ht["frac_of_device_resident"], "calib": hc.get("frac_of_device_resident")},
                       "train_inline_upload_imgs_per_s": ht["inline_upload_imgs_per_s"],
                       "calib_inline_upload_imgs_per_s": hc.get("host_inline_upload_imgs_per_s"), "train": ht, "calib": hc}
        except Exception as e:  # noqa: BLE001  -- a companion record must never cost the headline line
            host_ds = {"error": f"{type(e).__name__}: {e}"}
        nn_ops.set_compute_dtype(conf["dtype"])
        _phase("host-dataset legs done")

    # ---------------------------------------------------------------- companions in the same run
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_extras and "train" in legs:
        # the config-faithful reading beside the weak-scaling one: the reference's GLOBAL batch of 78 split over the ranks
        ws = Workload(job, conf, args.uncertainty_type, strong=True)
        ts = ws.train_leg(args.steps, args.warmup)
        strong = {"value": ts["imgs_per_s"], "unit": "train imgs/s", "scaling": "strong", "global_batch": ws.global_batch,
                  "per_gpu_batch_rank0": ws.B, "ms_per_step": ts["ms_per_step"],
                  "host_enqueue_ms_per_step": ts["host_enqueue_ms_per_step"], "steps": args.steps, "warmup": args.warmup,
                  "note": "calibration is always split over the ranks (3,474 images in total), so `calib` is already the strong-scaling reading"}
        ws.release()
    others = pipeline = None
    if default_run and world == 1:
        others = {}
        for name, kw in (("batch10", dict(config="fastmri", batch=10, steps=40, warmup=8, calib=False)),
                         ("denoise32", dict(config="denoise32", steps=40, warmup=6)), ("temca1024", dict(config="temca1024", steps=3)),
                         ("bsbcm512_fp8", dict(config="bsbcm512", steps=4)), ("bsbcm512_bf16", dict(config="bsbcm512", steps=4, dtype="bf16", calib=False))):
            try:
                cname = kw.pop("config")
                others[name] = sub_record(job, cname, **kw)
            except Exception as e:  # noqa: BLE001  -- a sub-record must never cost the headline line
                others[name] = {"error": f"{type(e).__name__}: {e}"}
        nn_ops.set_compute_dtype(conf["dtype"])
        try:
            pipeline = fastmri_pipeline_record(job)
        except Exception as e:  # noqa: BLE001
            pipeline = {"error": f"{type(e).__name__}: {e}"}

    _phase("other configs + pipeline done")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(320)
        if calib and cpu.get("calib_end_to_end"):
            calib["vs_cpu_end_to_end"] = calib["value"] / cpu["calib_end_to_end"]["value"]

    _phase("cpu baseline done")
    if rank == 0:
        what = "quantile regression" if args.uncertainty_type == "quantiles" else args.uncertainty_type
        line = {
            "metric": f"train imgs/sec (+ calib imgs/sec in `calib`), {hw}x{hw} UNet " + what,
            "value": train_ips, "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tr["ms_per_step"], "host_enqueue_ms_per_step": tr["host_enqueue_ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": conf["dtype"],
            "data": "synthetic (shaped like the named dataset, random-init weights)",
            "config": {"workload": conf["label"] + (f" [overrides: {', '.join(custom)}]" if custom else "") + f", {args.uncertainty_type}",
                       "per_gpu_batch": B, "global_batch": global_batch, "parallelism": f"dp{world}", "unet_depth": wl.depth,
                       "n_in": wl.n_in, "gflop_per_image_train": wl.train_flop / 1e9,
                       "train_tflops": train_ips * wl.train_flop / 1e12},
            "distributed": dist_info,
            "roofline": roof, "roofline_wgrad": roof_w, "fp32": fp32, "calib": calib, "cpu_baseline": cpu, "per_kernel": per_kernel,
        }
        if roof_dgrad:
            line["roofline_bf16_igemm"] = roof_dgrad
        if strong:
            line["strong"] = strong
        if world > 1:
            # [r5] the two numbers a reader of a scaling run wants first, at the top level: the config-faithful strong-scaling step
            # (the reference's global batch of 78 split over the ranks) and the part of the gradient exchange a step does not hide
            line["strong_ms_per_step"] = strong["ms_per_step"] if strong else (tr["ms_per_step"] if args.scaling == "strong" else None)
            line["strong_imgs_per_s"] = strong["value"] if strong else (train_ips if args.scaling == "strong" else None)
            line["grad_exchange_exposed_ms"] = dist_info.get("grad_exchange_exposed_ms")
            line["allreduce_ms"] = dist_info.get("allreduce_ms")
            line["distinct_devices"] = dist_info.get("distinct_devices")
        if others:
            line["other_configs"] = others
        if pipeline:
            line["fastmri_pipeline"] = pipeline
        if cpu:
            line["vs_cpu_train"] = train_ips / cpu["value"]
        if host_ds:
            line["host_dataset"] = host_ds
        # [r6] the scalars two north-star targets hang on, at the TOP level and at the END of the line (a parser that keeps only scalars,
        # or only the tail of the line, still carries them)
        def _get(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        fp8, b16 = _get(others, "bsbcm512_fp8", "value"), _get(others, "bsbcm512_bf16", "value")
        line["batch10_ms_per_step"] = _get(others, "batch10", "ms_per_step")
        line["batch10_imgs_per_s"] = _get(others, "batch10", "value")
        line["temca1024_imgs_per_s"] = _get(others, "temca1024", "value")
        line["bsbcm512_fp8_over_bf16"] = (fp8 / b16) if fp8 and b16 else None
        line["denoise32_imgs_per_s"] = _get(others, "denoise32", "value")
        line["calib_imgs_per_s"] = _get(calib, "value")
        line["calib_forward_frac"] = _get(calib, "forward_roofline", "frac")
        line["calib_scoring_hbm_frac"] = _get(calib, "scoring_only", "roofline", "frac")
        line["conv_roofline_frac"] = _get(roof, "frac")
        line["conv_roofline_frac_in_timed_step"] = _get(roof, "frac_in_timed_step")
        line["wgrad_roofline_frac"] = _get(roof_w, "frac")
        line["host_train_imgs_per_s"] = _get(host_ds, "train_imgs_per_s")
        line["host_calib_imgs_per_s"] = _get(host_ds, "calib_imgs_per_s")
        line["host_train_frac_of_device_resident"] = _get(host_ds, "frac_of_device_resident", "train")
        line["host_calib_frac_of_device_resident"] = _get(host_ds, "frac_of_device_resident", "calib")
        line["cpu_baseline_imgs_per_s"] = _get(cpu, "value")
        line["value_repeat"] = train_ips
        line["ms_per_step_repeat"] = tr["ms_per_step"]
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
