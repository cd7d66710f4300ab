import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np, copy
from test_model_gpu import build, PARAMS, rel_l2, DEV
from oracle import model as om
from im2im_uq_amd import nn_ops
from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
from im2im_uq_amd.core.models.trunks.unet import UNet
for init in ("det", "default"):
    for size, b in ((32, 2), (96, 4), (160, 2)):
        torch.manual_seed(0)
        m = add_uncertainty(UNet(1, 1), dict(PARAMS))
        if init == "det": m.load_state_dict(om.det_state(1, 1))
        m = m.to(DEV)
        x, y = om.det_images(b, 1, size, size, salt=7)
        if init == "default":
            x = torch.randn(b, 1, size, size); y = torch.rand(b, 1, size, size)
        outs = {}
        for mode in ("train", "eval"):
            for dt in ("fp32", "bf16"):
                nn_ops.set_compute_dtype(dt)
                mm = copy.deepcopy(m)
                mm.train(mode == "train")
                with torch.no_grad():
                    outs[(mode, dt)] = mm(x.to(DEV)).float().cpu()
            print(f"init={init} size={size} B={b} {mode}: bf16 vs fp32 rel_l2 = {rel_l2(outs[(mode,'bf16')], outs[(mode,'fp32')]):.4f}")
