"""where the HOST time of one training step goes (cProfile over 10 eager steps at the bench shape)."""
import sys, os, cProfile, pstats, io
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
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
