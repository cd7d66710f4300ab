"""does one HIP graph hold the whole training step?  eager vs captured (forward + loss + backward; Adam eager) at the bench
shape: ms per step, host enqueue ms per step, and loss trajectories side by side."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 78)); hw = int(os.environ.get("HW", 320)); depth = int(os.environ.get("DEPTH", 4))
nn_ops.set_compute_dtype("bf16")
cfg = dict(bench.PARAMS, device=str(dev), batch_size=B, uncertainty_type="quantiles", num_lambdas=100, minimum_lambda=0.0, maximum_lambda=6.0)

def make():
    torch.manual_seed(0)
    m = add_uncertainty(UNet(1, 1, depth=depth), cfg).to(dev)
    return m, nn_ops.FusedAdam(m.parameters(), lr=1e-3)

g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 1, hw, hw, device=dev, generator=g); y = torch.rand(B, 1, hw, hw, device=dev, generator=g)

def run(fn, steps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    h = time.perf_counter() - t0; torch.cuda.synchronize(); d = time.perf_counter() - t0
    return d / steps * 1e3, h / steps * 1e3

m, opt = make()
losses_e = []
def eager():
    loss = m.loss_fn(m(x), y); opt.zero_grad(); loss.backward(); opt.step(); losses_e.append(loss.detach())
print("eager ms/step, host ms/step:", run(eager))

m2, opt2 = make()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt2.zero_grad(set_to_none=True); loss = m2.loss_fn(m2(x), y); loss.backward(); opt2.step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
opt2.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    sloss = m2.loss_fn(m2(x), y)
    sloss.backward()
    nn_ops.join_side_streams()
losses_g = []
def graphed():
    graph.replay(); opt2.step(); losses_g.append(sloss.detach().clone())
print("graph ms/step, host ms/step:", run(graphed))
try:
    m3, opt3 = make()
    s3 = torch.cuda.Stream(); s3.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s3):
        for _ in range(3):
            opt3.zero_grad(set_to_none=True); l3 = m3.loss_fn(m3(x), y); l3.backward(); opt3.step()
    torch.cuda.current_stream().wait_stream(s3)
    g3 = torch.cuda.CUDAGraph(); opt3.zero_grad(set_to_none=True)
    with torch.cuda.graph(g3):
        sl3 = m3.loss_fn(m3(x), y); sl3.backward(); nn_ops.join_side_streams(); opt3.step()
    print("graph incl. Adam ms/step, host ms/step:", run(lambda: g3.replay()))
except Exception as e:  # noqa: BLE001
    print("graph incl. Adam: failed:", repr(e)[:300])
le = torch.stack(losses_e).cpu(); lg = torch.stack(losses_g).cpu()
print("eager losses", le[:6].tolist()); print("graph losses (3 steps later start)", lg[:6].tolist())
