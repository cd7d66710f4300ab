import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
dev = torch.device("cuda:0")
x = torch.randn(10, 64, 320, 320, device=dev).to(memory_format=torch.channels_last)
def t(fn, n=2000):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("permute plain              %.2f us" % t(lambda: x.permute(0, 2, 3, 1)))
xr = x.clone().requires_grad_(True)
print("permute requires_grad      %.2f us" % t(lambda: xr.permute(0, 2, 3, 1)))
xa = x.clone(); xa._im2im_lazy_ss = torch.zeros(2, 64, device=dev)
print("permute with py attribute  %.2f us" % t(lambda: xa.permute(0, 2, 3, 1)))
print("nhwc(x.detach())           %.2f us" % t(lambda: nn_ops.nhwc(x.detach())))
# whole step host time, split forward / backward / optimizer, no profiler
B = int(os.environ.get("B", 10))
nn_ops.set_compute_dtype("bf16")
cfg = dict(bench.PARAMS, device=str(dev), batch_size=B, num_lambdas=100, minimum_lambda=0.0, maximum_lambda=6.0)
torch.manual_seed(0)
m = add_uncertainty(UNet(1, 1), cfg).to(dev)
opt = nn_ops.FusedAdam(m.parameters(), lr=1e-3)
xi = torch.randn(B, 1, 320, 320, device=dev); y = torch.rand(B, 1, 320, 320, device=dev)
tf = tb = to = 0.0
for i in range(25):
    if i == 5: tf = tb = to = 0.0; torch.cuda.synchronize()
    t0 = time.perf_counter(); loss = m.loss_fn(m(xi), y); t1 = time.perf_counter()
    opt.zero_grad(); loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
    if i % 5 == 4: torch.cuda.synchronize()      # keep the queue short: host times are enqueue times, not back-pressure
print("host per step: forward %.2f ms  backward %.2f ms  optimizer %.2f ms" % (tf / 20 * 1e3, tb / 20 * 1e3, to / 20 * 1e3))
