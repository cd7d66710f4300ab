#!/usr/bin/env python3
"""calibrate_model from a pageable host TensorDataset (3,474 images, batch 78): producer thread vs consumer-thread fetch vs GIL switch interval."""
import contextlib, io, os, sys, time
import torch
from torch.utils.data import TensorDataset
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from im2im_uq_amd import launch, prefetch
from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
dist, rank, world, dev, backend = launch.init_distributed(expected_world=1)
job = bench.Job(dist, rank, world, dev, backend)
wl = bench.Workload(job, dict(bench.CONFIGS["fastmri"]))
model = wl.model.eval()
M = 3474
g = torch.Generator().manual_seed(0)
xh, yh = torch.randn(M, 1, 320, 320, generator=g), torch.rand(M, 1, 320, 320, generator=g)
ds_host = TensorDataset(xh, yh); ds_host.im2im_local_shard = True
ds_dev = TensorDataset(xh.to(dev), yh.to(dev)); ds_dev.im2im_local_shard = True
cfg = dict(wl.cfg, batch_size=78)
def run(ds):
    with contextlib.redirect_stdout(io.StringIO()):
        calibrate_model(model, ds, cfg)
    torch.cuda.synchronize()
def timed(ds):
    run(ds)
    t0 = time.perf_counter(); run(ds); return M / (time.perf_counter() - t0)
for rep in range(2):
    base = timed(ds_dev)
    out = [f"resident {base:7.0f}"]
    for name, th, sw, depth in (("thread", True, 0.005, 2), ("no-thread", False, 0.005, 2), ("thread sw=0.5ms", True, 0.0005, 2), ("thread depth 4", True, 0.005, 4), ("no-thread depth 4", False, 0.005, 4)):
        prefetch.THREAD, prefetch.DEPTH = th, depth
        sys.setswitchinterval(sw)
        v = timed(ds_host)
        out.append(f"{name} {v:7.0f} ({v / base:.3f})")
    sys.setswitchinterval(0.005); prefetch.THREAD, prefetch.DEPTH = True, 2
    print(" | ".join(out))
