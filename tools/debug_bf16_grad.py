import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np, copy
from test_model_gpu import PARAMS, rel_l2, DEV
from oracle import model as om
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
for init in ("det", "default"):
    torch.manual_seed(0)
    m = add_uncertainty(UNet(1, 1), dict(PARAMS))
    if init == "det": m.load_state_dict(om.det_state(1, 1))
    m = m.to(DEV)
    size, b = 96, 4
    x, y = om.det_images(b, 1, size, size, salt=7)
    grads = {}
    for dt in ("fp32", "bf16"):
        nn_ops.set_compute_dtype(dt)
        mm = copy.deepcopy(m); mm.train()
        loss = mm.loss_fn(mm(x.to(DEV)), y.to(DEV)); loss.backward()
        grads[dt] = {n: p.grad.float().cpu() for n, p in mm.named_parameters()}
        print(init, dt, "loss", loss.item())
    for n in grads["fp32"]:
        if n.endswith("0.weight") or n.endswith("3.weight") or "last_layer" in n or "out.conv" in n:
            print(f"  {init} {n:60s} {rel_l2(grads['bf16'][n], grads['fp32'][n]):.4f}")
