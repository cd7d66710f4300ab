"""spread of the end-to-end training outcome (tests/test_train_parity_gpu.py's run) over small perturbations of the initial
weights, per compute mode: how far apart do two fp32 runs land, and does bf16 land inside that spread?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import test_train_parity_gpu as t
from im2im_uq_amd.core.datasets.synthetic import SyntheticDenoiseDataset

hw, steps = 64, int(os.environ.get("STEPS", "600"))
t.PARAMS["lr"] = float(os.environ.get("LR", "1e-3"))
ds = SyntheticDenoiseDataset(num_images=96 * 3, num_inputs=1, side=hw, noise=0.1, seed=5)
x, y = ds.x.to(t.DEV), ds.y.to(t.DEV)
data = ((x[:96], y[:96]), (x[96:192], y[96:192]), (x[192:], y[192:]))
ref = t._run("fp32", data, steps, hw)
for dt in ("fp32", "bf16"):
    for seed, pert in ((1, 1e-4), (2, 1e-4), (0, 0.0)):
        if dt == "fp32" and pert == 0.0:
            continue
        r = t._run(dt, data, steps, hw, perturb=pert, seed=seed)
        size, rsize = float((r["hi"] - r["lo"]).mean()), float((ref["hi"] - ref["lo"]).mean())
        d = t._distance(r, ref)
        print(f"{dt} seed {seed} pert {pert}: tail {r['losses'][-200:].mean():.5f} lhat {r['lhat']:.3f} risk {r['risk']:.4f} size {size:.4f} (ref {rsize:.4f}) "
              + " ".join(f"{k} {v:.3f}" for k, v in d.items()), flush=True)
