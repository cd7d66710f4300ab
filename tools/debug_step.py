import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from test_model_gpu import build, PARAMS, rel_l2, DEV
from oracle import model as om
from im2im_uq_amd import nn_ops
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model = build(1, dt); model.train()
opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
st = om.det_state(1, 1)
leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if om.is_param(k)}
work = dict(st); work.update(leaves)
ropt = torch.optim.Adam(list(leaves.values()), lr=1e-3)
for step in range(2):
    x, y = om.det_images(4, 1, 32, 32, salt=step)
    loss = model.loss_fn(model(x.to(DEV)), y.to(DEV)); opt.zero_grad(); loss.backward()
    rl = om.quantile_loss(om.model_forward(x, work, training=True), y, PARAMS); ropt.zero_grad(); rl.backward()
    print(f"step {step}: loss {loss.item():.7f} ref {rl.item():.7f}")
    rows = []
    for name, p in model.named_parameters():
        g, rg = p.grad.cpu(), leaves[name].grad
        rows.append((rel_l2(g, rg), name, float(rg.abs().max()), float((g-rg).abs().max())))
    rows.sort(reverse=True)
    for r in rows[:12]: print("  grad rel %.2e  %-55s max|ref| %.2e max|diff| %.2e" % r)
    opt.step(); ropt.step()
    rows = []
    for name, p in model.named_parameters():
        d = (p.detach().cpu() - leaves[name].detach()).abs()
        rows.append((float(d.max()), float((d > 1e-4).float().mean()), name))
    rows.sort(reverse=True)
    for r in rows[:12]: print("  param maxdiff %.2e frac>1e-4 %.4f %s" % r)
