"""where the time of one end-to-end calibrate_model goes at the bench configuration (N = 3,474, 320x320, 1000 lambdas)."""
import sys, os, time, io, contextlib, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.data import TensorDataset
import bench
from im2im_uq_amd import nn_ops, hip_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
from im2im_uq_amd.core.calibration import calibrate_model as cm

dev = torch.device("cuda:0")
nn_ops.set_compute_dtype("bf16")
bs = int(os.environ.get("BS", 78))
cfg = dict(bench.PARAMS, device=str(dev), batch_size=bs, num_lambdas=1000, minimum_lambda=0.0, maximum_lambda=6.0)
torch.manual_seed(0)
model = add_uncertainty(UNet(1, 1), cfg).to(dev)
M = 3474
g = torch.Generator(device=dev).manual_seed(5)
xc = torch.randn(M, 1, 320, 320, device=dev, generator=g)
model.eval()
with torch.no_grad():
    outs = torch.cat([model(xc[i:i + 78]) for i in range(0, M, 78)])
    mid = outs[:, 1]
    yc = (mid + (outs[:, 2] - mid).abs() / 1.96 * torch.randn(mid.shape, device=dev, generator=g)).contiguous()
del outs
ds = TensorDataset(xc, yc)
def sync(): torch.cuda.synchronize()
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        sync(); t0 = time.perf_counter()
        o, l = cm.collect_outputs(model, ds, cfg, dev); t_enq = time.perf_counter(); sync(); t1 = time.perf_counter()
        lambdas = cm.lambda_grid(cfg); dl = lambdas[1] - lambdas[0]
        model.set_lhat(lambdas[-1] + dl - 1e-9)
        table = hip_ops.rcps_loss_table(o, l, lambdas - dl, form=cm.sets_form(model)); sync(); t2 = time.perf_counter()
        lhat, tab, trace = cm.scan_loss_table(table, lambdas, 0.1, 0.1); t3 = time.perf_counter()
    print(f"collect_outputs {1e3*(t1-t0):.1f} ms (host enqueue {1e3*(t_enq-t0):.1f}) | table {1e3*(t2-t1):.1f} | scan {1e3*(t3-t2):.1f} | visited {len(trace)} | total {1e3*(t3-t0):.1f}")
with contextlib.redirect_stdout(io.StringIO()):
    sync(); t0 = time.perf_counter(); cm.calibrate_model(model, ds, cfg); sync(); t1 = time.perf_counter()
print(f"calibrate_model end to end {1e3*(t1-t0):.1f} ms = {M/(t1-t0):.0f} img/s at batch {bs}")
