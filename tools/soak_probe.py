"""300 training steps at the bench shape without a host sync: loss trajectory, allocator counters at start / end (the
record_stream growth of profiles/r02_allocator_stall.txt must not come back), step time of the first and last 50 steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet

dev = torch.device("cuda:0")
B, hw = 78, 320
nn_ops.set_compute_dtype("bf16")
cfg = dict(bench.PARAMS, device=str(dev), batch_size=B, num_lambdas=100, minimum_lambda=0.0, maximum_lambda=6.0)
torch.manual_seed(0)
m = add_uncertainty(UNet(1, 1), cfg).to(dev)
opt = nn_ops.FusedAdam(m.parameters(), lr=1e-3)
g = torch.Generator(device=dev).manual_seed(3)
y = torch.rand(B, 1, hw, hw, device=dev, generator=g)
x = y + 0.1 * torch.randn(B, 1, hw, hw, device=dev, generator=g)
losses = []
def step():
    loss = m.loss_fn(m(x), y); opt.zero_grad(); loss.backward(); opt.step(); losses.append(loss.detach())
def stats():
    s = torch.cuda.memory_stats()
    return dict(mallocs=s["num_device_alloc"], reserved_gb=round(s["reserved_bytes.all.current"] / 2**30, 1), retries=s["num_alloc_retries"])
for _ in range(5): step()
torch.cuda.synchronize(); print("after warm-up", stats(), flush=True)
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); t3 = time.perf_counter()
l = torch.stack(losses).float().cpu()
print("after 305 steps", stats())
print(f"ms/step first 50: {(t1-t0)/50*1e3:.2f}   middle 200: {(t2-t1)/200*1e3:.2f}   last 50: {(t3-t2)/50*1e3:.2f}")
print("loss at steps 0, 5, 20, 50, 100, 200, 304:", [round(float(l[i]), 5) for i in (0, 5, 20, 50, 100, 200, 304)], "all finite:", bool(torch.isfinite(l).all()))
