"""print the fp32 / bf16 loss curves of tests/test_train_parity_gpu.py's run at a few checkpoints (tolerance calibration)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_train_parity_gpu as t
from im2im_uq_amd.core.datasets.synthetic import SyntheticDenoiseDataset

for lr in (1e-3, 3e-4):
    t.PARAMS["lr"] = lr
    hw, steps = 64, 600
    ds = SyntheticDenoiseDataset(num_images=96 * 3, num_inputs=1, side=hw, noise=0.1, seed=5)
    x, y = ds.x.to(t.DEV), ds.y.to(t.DEV)
    data = ((x[:96], y[:96]), (x[96:192], y[96:192]), (x[192:], y[192:]))
    res = {dt: t._run(dt, data, steps, hw) for dt in ("fp32", "bf16", "fp32")}
    r32, r16 = res["fp32"], res["bf16"]
    for a in range(40, steps + 1, 40):
        m32, m16 = r32["losses"][a - 40:a].mean(), r16["losses"][a - 40:a].mean()
        print(f"lr {lr} steps {a - 40}-{a}: fp32 {m32:.5f} bf16 {m16:.5f} ratio {m16 / m32:.4f}")
    print("lhat", r32["lhat"], r16["lhat"], "risk", r32["risk"], r16["risk"],
          "rel", t.rel_l2(r16["mid"], r32["mid"]), t.rel_l2(r16["lo"], r32["lo"]), t.rel_l2(r16["hi"], r32["hi"]))
