"""why is a second bench process on the same box sometimes 2x slower?  per-step host time and allocator counters."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 78)); hw = 320
nn_ops.set_compute_dtype("bf16")
cfg = dict(bench.PARAMS, device=str(dev), batch_size=B, num_lambdas=100, minimum_lambda=0.0, maximum_lambda=6.0)
torch.manual_seed(0)
m = add_uncertainty(UNet(1, 1), cfg).to(dev)
opt = nn_ops.FusedAdam(m.parameters(), lr=1e-3)
x = torch.randn(B, 1, hw, hw, device=dev); y = torch.rand(B, 1, hw, hw, device=dev)
def step():
    loss = m.loss_fn(m(x), y); opt.zero_grad(); loss.backward(); opt.step()
def stats():
    s = torch.cuda.memory_stats()
    return dict(mallocs=s["num_device_alloc"], frees=s["num_device_free"], retries=s["num_alloc_retries"], reserved_gb=round(s["reserved_bytes.all.current"] / 2**30, 1),
                peak_alloc_gb=round(s["allocated_bytes.all.peak"] / 2**30, 1))
t00 = time.perf_counter()
for i in range(5): step()
print("after warmup (no sync)", round(time.perf_counter() - t00, 3), stats(), flush=True)
torch.cuda.synchronize()
print("warmup synced", round(time.perf_counter() - t00, 3), flush=True)
t0 = time.perf_counter(); hs = []
for i in range(20):
    t = time.perf_counter(); step(); hs.append(round((time.perf_counter() - t) * 1e3, 1))
h = time.perf_counter() - t0; torch.cuda.synchronize(); d = time.perf_counter() - t0
print("ms/step", round(d / 20 * 1e3, 2), "host", round(h / 20 * 1e3, 2), "per-step host ms", hs, stats(), flush=True)
