#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the reference never
travels to the GPU box -- only the small .npz files written here do.  Each
fixture holds inputs and the reference's outputs for one reference call site.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixture ids follow SURVEY.md section 8(c): G1..G11; G12 covers the final layers of section 8(f) rank 1.
(`make_golden.py g12` regenerates only the named groups.)
"""
import os
import sys
import types
import io
import contextlib

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("IM2IM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# wandb is absent in the image: in-memory no-op stub so core.scripts.{train,eval} import.
wandb = types.ModuleType("wandb")
wandb.init = lambda *a, **k: None
wandb.log = lambda *a, **k: None
wandb.watch = lambda *a, **k: None
wandb.Image = lambda x: x
wandb.config = {}
sys.modules["wandb"] = wandb
sys.modules.setdefault("h5py", types.ModuleType("h5py"))    # only FastMRIDataset's file reading needs it; the transforms do not

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import TensorDataset

from core.models.losses.pinball import PinballLoss
from core.models.finallayers.quantile_layer import quantile_regression_loss_fn
from core.models.trunks.unet_parts import DoubleConv, Down, Up, OutConv
from core.models.trunks.unet import UNet
from core.models.add_uncertainty import add_uncertainty, ModelWithUncertainty
from core.models.finallayers.quantile_layer import quantile_regression_nested_sets_from_output
from core.calibration.calibrate_model import (calibrate_model, fraction_missed_loss,
                                              get_rcps_metrics_from_outputs, get_rcps_losses_from_outputs)
from core.calibration.bounds import HB_mu_plus
from core.utils import fix_randomness
import core.scripts.train as ref_train
import core.scripts.eval as ref_eval

from oracle import model as om
from oracle import calibration as oc

PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=50, rcps_loss="fraction_missed", minimum_lambda=0,
              maximum_lambda=6, device="cpu", dataset="synthetic", batch_size=8, lr=1e-3,
              input_normalization="standard", output_normalization="min-max", num_validation_images=2)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} B)")


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


# ---------------------------------------------------------------- G1 / G2
def g1_g2():
    fix_randomness(0)
    out = torch.randn(4, 16, 16)
    tgt = torch.rand(4, 16, 16)
    out[0, 0, :8] = tgt[0, 0, :8]           # exact ties e == 0
    res = {"output": out, "target": tgt}
    for q in (0.05, 0.95):
        o = out.clone().requires_grad_(True)
        l = PinballLoss(quantile=q)(o, tgt)
        l.backward()
        tag = str(q).replace(".", "")
        res[f"loss_{tag}"] = l.detach()
        res[f"grad_{tag}"] = o.grad
    save("g1_pinball", **res)

    pred = torch.randn(4, 3, 1, 16, 16) * 0.3 + 0.5
    y = torch.rand(4, 1, 16, 16)
    pred[1, 0, 0, 3, :5] = y[1, 0, 3, :5]
    p = pred.clone().requires_grad_(True)
    l = quantile_regression_loss_fn(p, y, PARAMS)
    l.backward()
    w = dict(PARAMS, q_lo_weight=0.5, q_hi_weight=2.0, mse_weight=3.0, q_lo=0.1, q_hi=0.8)
    p2 = pred.clone().requires_grad_(True)
    l2 = quantile_regression_loss_fn(p2, y, w)
    l2.backward()
    save("g2_quantile_loss", pred=pred, target=y, loss=l.detach(), grad=p.grad,
         loss_w=l2.detach(), grad_w=p2.grad, w=np.array([0.5, 2.0, 3.0, 0.1, 0.8]))


# ---------------------------------------------------------------- G3 parts
def run_part(mod, inputs, train):
    mod.train(train)
    ins = [t.clone().requires_grad_(True) for t in inputs]
    y = mod(*ins)
    g = torch.cos(torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape) * 0.37)  # fixed upstream grad
    y.backward(g)
    rec = {"y": y.detach(), "gy": g}
    for i, t in enumerate(ins):
        rec[f"x{i}"] = inputs[i]
        rec[f"gx{i}"] = t.grad
    for k, v in mod.state_dict().items():
        rec["state_after." + k] = v.clone()
    for k, v in mod.named_parameters():
        rec["grad." + k] = v.grad.clone()
    return rec


def g3():
    fix_randomness(0)
    parts = {
        "doubleconv": (DoubleConv(3, 8, 6), [torch.randn(2, 3, 16, 16)]),
        "down": (Down(8, 16), [torch.randn(2, 8, 16, 16)]),
        "up_bilinear": (Up(16, 8, True), [torch.randn(2, 8, 8, 8), torch.randn(2, 8, 16, 16)]),
        "up_bilinear_pad": (Up(16, 8, True), [torch.randn(2, 8, 5, 6), torch.randn(2, 8, 11, 13)]),
        "up_convT": (Up(16, 8, False), [torch.randn(2, 16, 8, 8), torch.randn(2, 8, 16, 16)]),
        "outconv": (OutConv(8, 4), [torch.randn(2, 8, 16, 16)]),
    }
    for name, (mod, ins) in parts.items():
        # non-trivial BN affine + running stats
        with torch.no_grad():
            for k, v in mod.state_dict().items():
                if "running_var" in k:
                    v.copy_(1 + 0.3 * torch.rand_like(v))
                elif "running_mean" in k:
                    v.copy_(0.2 * torch.randn_like(v))
            for k, p in mod.named_parameters():
                if p.dim() == 1:
                    p.add_(0.2 * torch.randn_like(p))
        init = {"state_before." + k: v.clone() for k, v in mod.state_dict().items()}
        rec = dict(init)
        for train in (True, False):
            mod.load_state_dict({k[len("state_before."):]: v for k, v in init.items()})
            mod.zero_grad()
            r = run_part(mod, ins, train)
            rec.update({("train." if train else "eval.") + k: v for k, v in r.items()})
        save("g3_" + name, **rec)


# ---------------------------------------------------------------- G4 / G5
def build_ref_model(n_in=1):
    model = add_uncertainty(UNet(n_in, 1), dict(PARAMS))
    st = om.det_state(n_in, 1)
    missing = model.load_state_dict(st, strict=True)
    return model, st


def g4():
    for n_in, hw in ((1, 32), (2, 48)):
        model, st = build_ref_model(n_in)
        x, y = om.det_images(2, n_in, hw, hw, salt=1)
        model.eval()
        with torch.no_grad():
            out_eval = model(x)
        model.train()
        out_train = model(x)
        sd = model.state_dict()
        save(f"g4_model_fwd_nin{n_in}", x=x, out_eval=out_eval, out_train=out_train.detach(),
             rm_inc1=sd["baseModel.inc.double_conv.1.running_mean"],
             rv_inc1=sd["baseModel.inc.double_conv.1.running_var"],
             rm_up4=sd["baseModel.up4.conv.double_conv.4.running_mean"],
             rv_up4=sd["baseModel.up4.conv.double_conv.4.running_var"],
             nbt=sd["baseModel.inc.double_conv.1.num_batches_tracked"])


def g5():
    model, st = build_ref_model(1)
    lr = 1e-3
    opt = torch.optim.Adam(model.parameters(), lr=lr)     # train.py:120
    model.train()
    losses = []
    xs, ys = [], []
    for step in range(5):
        x, y = om.det_images(4, 1, 32, 32, salt=step)
        y = y[:, :1]
        xs.append(x); ys.append(y)
        pred = model(x)                                    # train.py:152
        loss = model.loss_fn(pred, y)                      # train.py:153
        losses.append(loss.item())
        opt.zero_grad(); loss.backward(); opt.step()       # train.py:158-162
    model.eval()
    xp, _ = om.det_images(2, 1, 32, 32, salt=9)
    with torch.no_grad():
        probe = model(xp)
    sd = model.state_dict()
    keys = ["baseModel.inc.double_conv.0.weight", "baseModel.down4.maxpool_conv.1.double_conv.3.weight",
            "baseModel.up1.conv.double_conv.0.weight", "baseModel.up4.conv.double_conv.4.weight",
            "baseModel.up4.conv.double_conv.4.bias", "baseModel.out.conv.weight", "baseModel.out.conv.bias",
            "last_layer.lower.weight", "last_layer.prediction.bias", "last_layer.upper.weight",
            "baseModel.inc.double_conv.1.running_mean", "baseModel.up4.conv.double_conv.4.running_var"]
    rec = {"losses": np.array(losses), "probe_x": xp, "probe_out": probe, "lr": np.array(lr)}
    for k in keys:   # big tensors: float64 checksums + a strided sample, not the full 9 MB
        flat = sd[k].flatten().double()
        stride = max(1, flat.numel() // 512)
        rec["sum." + k] = flat.sum()
        rec["l2." + k] = flat.pow(2).sum().sqrt()
        rec["sample." + k] = sd[k].flatten()[::stride][:512]
    save("g5_adam_trajectory", **rec)


# ---------------------------------------------------------------- G6
def identity_model():
    return ModelWithUncertainty(nn.Identity(), nn.Identity(), quantile_regression_loss_fn,
                                quantile_regression_nested_sets_from_output, dict(PARAMS))


def g6():
    out, y = oc.synth_outputs(3, 1, 16, 16, seed=3)
    out[0, 0, 0, 0, :4] = out[0, 1, 0, 0, :4] + 0.01      # lower above pred -> clamp path
    out[0, 2, 0, 1, :4] = out[0, 1, 0, 1, :4] - 0.01      # upper below pred -> clamp path
    m = identity_model()
    lams = torch.tensor([-0.1224, 0.0, 1e-3, 0.5, 1.0, 2.5, 6.0, 40.0])
    lows, ups = [], []
    for lam in lams:
        lo, mid, hi = m.nested_sets_from_output(out.clone(), lam)
        lows.append(lo); ups.append(hi)
    save("g6_nested_sets", output=out, lams=lams, lower=torch.stack(lows), upper=torch.stack(ups))


# ---------------------------------------------------------------- G7
def run_calib(out, y, cfg):
    m = identity_model()
    ds = TensorDataset(out.clone(), y.clone())
    with quiet():
        m, table = calibrate_model(m, ds, cfg)
    # per-lambda trace (Rhat, RhatPlus) recomputed with the reference's own functions
    lambdas = torch.linspace(cfg["minimum_lambda"], cfg["maximum_lambda"], cfg["num_lambdas"])
    dl = lambdas[1] - lambdas[0]
    trace = []
    for j in range(len(lambdas) - 1, -1, -1):
        with quiet():
            losses = get_rcps_losses_from_outputs(m, TensorDataset(out.clone(), y.clone()), fraction_missed_loss,
                                                  lambdas[j] - dl, "cpu")
            rp = HB_mu_plus(losses.mean().item(), losses.shape[0], cfg["delta"])
        trace.append((j, losses.mean().item(), rp))
        if losses.mean() >= cfg["alpha"] or rp > cfg["alpha"]:
            break
    return m.lhat, table, np.array(trace, dtype=np.float64)


def g7():
    cases = {}
    out, y = oc.synth_outputs(128, 1, 16, 16, seed=1)
    cases["mid"] = (out, y, dict(PARAMS, batch_size=32, num_lambdas=100, maximum_lambda=20))
    out, y = oc.synth_outputs(130, 1, 16, 16, seed=2)
    cases["n130"] = (out, y, dict(PARAMS, batch_size=64, num_lambdas=50, maximum_lambda=6))
    out, y = oc.synth_outputs(32, 1, 16, 16, seed=4, width=0.05)
    y2 = out[:, 1].clone()                       # labels == pred: zero risk at every lambda (Q3)
    cases["zero_risk"] = (out, y2, dict(PARAMS, batch_size=16, num_lambdas=20, maximum_lambda=6))
    out, y = oc.synth_outputs(48, 1, 16, 16, seed=5)
    cases["no_stop"] = (out, y, dict(PARAMS, batch_size=16, num_lambdas=10, minimum_lambda=50.0,
                                     maximum_lambda=60.0, alpha=0.5, delta=0.5))
    out, y = oc.synth_outputs(40, 2, 12, 20, seed=6)  # multi-channel, non-square
    cases["c2"] = (out, y, dict(PARAMS, batch_size=16, num_lambdas=64, maximum_lambda=8))
    for name, (out, y, cfg) in cases.items():
        lhat, table, trace = run_calib(out, y, cfg)
        save("g7_calibrate_" + name, output=out, label=y, lhat=lhat, table=table, trace=trace,
             cfg=np.array([cfg["alpha"], cfg["delta"], cfg["num_lambdas"], cfg["minimum_lambda"],
                           cfg["maximum_lambda"], cfg["batch_size"]], dtype=np.float64))


# ---------------------------------------------------------------- G8
def g8():
    rows = []
    with quiet():
        for muhat, n, delta in [(0.1, 10000, 0.1), (0.05, 100, 0.1), (0.08, 347, 0.1), (0.5, 10, 0.1),
                                (0.099, 3474, 0.1), (0.0, 50, 0.1), (1.0, 50, 0.1), (0.999, 200, 0.1)]:
            rows.append((muhat, n, delta, HB_mu_plus(muhat, n, delta)))
        rng = np.random.RandomState(0)
        for n in (16, 128, 434, 3474, 27794):
            for delta in (0.1, 0.01, 0.5):
                for muhat in np.concatenate([rng.rand(6), rng.rand(4) * 0.12, [1.0 / n, 0.1, 0.0999]]):
                    # muhat as the fp32-rounded value Rhat.item() would be
                    mh = float(np.float32(muhat))
                    rows.append((mh, n, delta, HB_mu_plus(mh, n, delta)))
    save("g8_hb_bound", rows=np.array(rows, dtype=np.float64))


# ---------------------------------------------------------------- G9 / G10
class _Wrap(torch.utils.data.Dataset):
    """map-style dataset of (model_output, label): with nn.Identity as the model this drives
    get_loss_table / eval_set_metrics through their forward loops unchanged."""
    def __init__(self, out, y):
        self.out, self.y = out, y
    def __len__(self):
        return self.out.shape[0]
    def __getitem__(self, i):
        return self.out[i], self.y[i]


def g9_g10():
    out, y = oc.synth_outputs(22, 1, 16, 16, seed=7)
    cfg = dict(PARAMS, num_lambdas=40, maximum_lambda=6)
    m = identity_model()
    with quiet():
        table = ref_eval.get_loss_table(m, _Wrap(out.clone(), y), cfg)
    save("g9_loss_table", output=out, label=y, table=table,
         cfg=np.array([cfg["num_lambdas"], cfg["minimum_lambda"], cfg["maximum_lambda"]], dtype=np.float64))

    out, y = oc.synth_outputs(70, 1, 16, 16, seed=8)
    m = identity_model()
    lhat = torch.tensor(1.25)
    m.set_lhat(lhat)
    fix_randomness(0)
    with quiet():
        losses, sizes, spearman, strat, mse, spatial = get_rcps_metrics_from_outputs(
            m, TensorDataset(out.clone(), y), fraction_missed_loss, "cpu")
    save("g10_metrics", output=out, label=y, lhat=lhat, losses=losses, spatial=spatial,
         rng_sizes=sizes, rng_spearman=np.array(spearman), rng_strat=strat, rng_mse=np.array(mse))


# ---------------------------------------------------------------- G11 end-to-end (config 1)
def g11():
    fix_randomness(0)
    n_train, n_cal, n_val, hw = 8, 16, 4, 32
    x, y = om.det_images(n_train + n_cal + n_val, 1, hw, hw, salt=3)
    cfg = dict(PARAMS, batch_size=8, lr=1e-3, num_lambdas=50, maximum_lambda=6)
    model, st = build_ref_model(1)
    tr = TensorDataset(x[:n_train], y[:n_train])
    ca = TensorDataset(x[n_train:n_train + n_cal], y[n_train:n_train + n_cal])
    va = TensorDataset(x[n_train + n_cal:], y[n_train + n_cal:])
    with quiet(), contextlib.redirect_stderr(io.StringIO()):
        model = ref_train.train_net(model, tr, va, "cpu", 2, 8, 1e-3, False, None, 100, 100, cfg)
        model.eval()
        with torch.no_grad():
            val_table = ref_eval.get_loss_table(model, va, cfg)
            model, cal_table = calibrate_model(model, ca, cfg)
            lo, mid, hi = model.nested_sets((x[n_train + n_cal:],))
    save("g11_end_to_end", x=x, y=y, lhat=model.lhat, cal_table=cal_table, val_table=val_table,
         lower=lo, pred=mid, upper=hi, split=np.array([n_train, n_cal, n_val]))


# ---------------------------------------------------------------- G12 the other final layers
def g12(only_types=()):
    """per uncertainty type: (a) final layer forward + train loss + gradients on a fixed feature map with the
    closed-form weights, (b) ModelWithUncertainty.nested_sets_from_output at several lambdas (with the floor) and the
    layer's own raw edges, (c) calibrate_model + get_rcps_metrics_from_outputs on synthetic outputs."""
    from core.models.add_uncertainty import add_uncertainty as ref_add

    class Trunk(nn.Module):                       # the factory only needs the two channel counts (add_uncertainty.py:57)
        n_channels_middle, n_channels_out = 32, 1

        def forward(self, x):
            return x

    for utype in ("quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "inn"):
        if only_types and utype not in only_types:
            continue
        params = dict(PARAMS, uncertainty_type=utype, beta=0.1)
        model = ref_add(Trunk(), params)
        st = om.det_state(1, 1, utype=utype)
        model.last_layer.load_state_dict({k[len("last_layer."):]: v for k, v in st.items() if k.startswith("last_layer.")})
        idx = torch.arange(2 * 32 * 16 * 16, dtype=torch.float64).reshape(2, 32, 16, 16)
        feat = (0.8 * torch.sin(0.37 * idx) * torch.cos(0.011 * idx + 0.3)).to(torch.float32).requires_grad_(True)
        y = (0.5 + 0.4 * torch.sin(0.05 * torch.arange(2 * 16 * 16, dtype=torch.float64))).to(torch.float32).reshape(2, 1, 16, 16)
        pred = model(feat)
        loss = model.loss_fn(pred, y)
        loss.backward()
        grads = {"g_" + n.replace(".", "_"): p.grad for n, p in model.last_layer.named_parameters()}
        # (b) nested sets
        lams = torch.tensor([-0.1224, 0.0, 1e-3, 0.5, 1.0, 2.5, 6.0])
        if utype in ("quantiles_l1", "inn"):
            out, lab = oc.synth_outputs(3, 1, 16, 16, seed=3)
        else:
            out, lab = oc.synth_outputs_two_plane(3, 1, 16, 16, seed=3, utype=utype)
        lows, ups, raw_lows, raw_ups = [], [], [], []
        for lam in lams:
            lo, mid, hi = model.nested_sets_from_output(out.clone(), lam)
            lows.append(lo); ups.append(hi)
            rlo, _, rhi = model.in_nested_sets_from_output_fn(model, out.clone(), lam)
            raw_lows.append(rlo); raw_ups.append(rhi)
        # (c) calibration on synthetic outputs through the Identity trick (G7)
        cfg = dict(params, batch_size=32, num_lambdas=80, maximum_lambda=8)
        if utype in ("quantiles_l1", "inn"):
            cout, cy = oc.synth_outputs(96, 1, 16, 16, seed=7)
        else:
            cout, cy = oc.synth_outputs_two_plane(96, 1, 16, 16, seed=7, utype=utype)
        ident = ModelWithUncertainty(nn.Identity(), nn.Identity(), model.in_train_loss_fn, model.in_nested_sets_from_output_fn, cfg)
        with quiet():
            ident, table = calibrate_model(ident, TensorDataset(cout.clone(), cy.clone()), cfg)
        fix_randomness(0)
        with quiet():
            losses, sizes, spearman, strat, mse, spatial = get_rcps_metrics_from_outputs(
                ident, TensorDataset(cout.clone(), cy.clone()), fraction_missed_loss, "cpu")
        save("g12_" + utype, feat=feat, target=y, pred=pred, loss=loss, g_feat=feat.grad, **grads,
             sets_output=out, lams=lams, lower=torch.stack(lows), upper=torch.stack(ups),
             raw_lower=torch.stack(raw_lows), raw_upper=torch.stack(raw_ups),
             cal_output=cout, cal_label=cy, lhat=ident.lhat, table=table, risk=losses, spatial=spatial,
             sizes=sizes, spearman=np.float64(spearman), mse=np.float64(mse), strat=strat,
             cfg=np.array([cfg["alpha"], cfg["delta"], cfg["num_lambdas"], cfg["minimum_lambda"], cfg["maximum_lambda"],
                           cfg["batch_size"]], dtype=np.float64))


# ---------------------------------------------------------------- G13 softmax final layer
def g13():
    """softmax layer (num_softmax = 50): forward + cross-entropy loss + gradients on a fixed feature map; nested sets of
    synthetic class logits at several lambdas; calibrate_model + metrics through the Identity trick."""
    from core.models.add_uncertainty import add_uncertainty as ref_add
    from core.models.finallayers.softmax_layer import softmax_loss_fn, softmax_nested_sets_from_output

    class Trunk(nn.Module):
        n_channels_middle, n_channels_out = 32, 1

        def forward(self, x):
            return x

    params = dict(PARAMS, uncertainty_type="softmax", num_softmax=50, minimum_lambda_softmax=0, maximum_lambda_softmax=2.0)
    model = ref_add(Trunk(), params)
    st = om.det_state(1, 1, utype="softmax")
    model.last_layer.load_state_dict({k[len("last_layer."):]: v for k, v in st.items() if k.startswith("last_layer.")})
    idx = torch.arange(2 * 32 * 16 * 16, dtype=torch.float64).reshape(2, 32, 16, 16)
    feat = (0.8 * torch.sin(0.37 * idx) * torch.cos(0.011 * idx + 0.3)).to(torch.float32).requires_grad_(True)
    y = (0.5 + 0.45 * torch.sin(0.05 * torch.arange(2 * 16 * 16, dtype=torch.float64))).to(torch.float32).reshape(2, 1, 16, 16)
    y.view(-1)[:3] = torch.tensor([0.0, 1.0, 1.0 / 49])            # exact class edges (bucketize right=False)
    pred = model(feat)
    loss = model.loss_fn(pred, y)
    loss.backward()
    grads = {"g_" + n.replace(".", "_"): p.grad for n, p in model.last_layer.named_parameters()}
    lams = torch.tensor([-0.0253, 0.0, 0.25, 0.5, 0.99, 2.0])
    out, lab = oc.synth_logits(3, 50, 16, 16, seed=3)
    lows, ups, raw_lows, raw_ups, preds = [], [], [], [], []
    for lam in lams:
        lo, mid, hi = model.nested_sets_from_output(out.clone(), lam)
        lows.append(lo); ups.append(hi); preds.append(mid)
        rlo, _, rhi = softmax_nested_sets_from_output(model, out.clone(), lam)
        raw_lows.append(rlo); raw_ups.append(rhi)
    cfg = dict(params, batch_size=32, num_lambdas=80)
    cout, cy = oc.synth_logits(96, 50, 16, 16, seed=7)
    cout = cout.to(torch.float16).to(torch.float32)                # fp16-representable logits: the fixture stores them in 2 bytes
    ident = ModelWithUncertainty(nn.Identity(), nn.Identity(), softmax_loss_fn, softmax_nested_sets_from_output, cfg)
    with quiet():
        ident, table = calibrate_model(ident, TensorDataset(cout.clone(), cy.clone()), cfg)
    fix_randomness(0)
    with quiet():
        losses, sizes, spearman, strat, mse, spatial = get_rcps_metrics_from_outputs(
            ident, TensorDataset(cout.clone(), cy.clone()), fraction_missed_loss, "cpu")
    save("g13_softmax", feat=feat, target=y, pred=pred, loss=loss, g_feat=feat.grad, **grads,
         sets_output=out, lams=lams, lower=torch.stack(lows), upper=torch.stack(ups), prediction=torch.stack(preds),
         raw_lower=torch.stack(raw_lows), raw_upper=torch.stack(raw_ups),
         cal_output=cout.to(torch.float16), cal_label=cy, lhat=ident.lhat, table=table, risk=losses, spatial=spatial,
         cfg=np.array([cfg["alpha"], cfg["delta"], cfg["num_lambdas"], cfg["minimum_lambda_softmax"],
                       cfg["maximum_lambda_softmax"], cfg["batch_size"]], dtype=np.float64))


# ---------------------------------------------------------------- G14 UNet blocks at kernel-supported channel counts
def g14(only_parts=()):
    """G3's blocks again with channel counts the MFMA kernels take (multiples of 32; <= 8 for a first conv), closed-form
    weights (oracle.model.det_fill keyed by the parameter name, so nothing but inputs and results is stored): forward,
    input/parameter gradients and running statistics in train mode, forward in eval mode."""
    parts = {
        "doubleconv": (DoubleConv(2, 64, 32), [(2, 2, 12, 16)]),
        "down": (Down(32, 64), [(2, 32, 12, 16)]),
        "up_bilinear": (Up(128, 64, True), [(2, 64, 6, 8), (2, 64, 12, 16)]),
        "up_bilinear_pad": (Up(128, 64, True), [(2, 64, 5, 7), (2, 64, 11, 15)]),
        "outconv": (OutConv(64, 32), [(2, 64, 12, 16)]),
        "up_convT": (Up(128, 64, False), [(2, 128, 6, 8), (2, 64, 12, 16)]),
        "up_convT_pad": (Up(128, 64, False), [(2, 128, 5, 7), (2, 64, 11, 15)]),
    }
    for name, (mod, shapes) in parts.items():
        if only_parts and name not in only_parts:
            continue
        mod.load_state_dict({k: om.det_fill("g14." + name + "." + k, tuple(v.shape)) for k, v in mod.state_dict().items()})
        ins = []
        for i, shp in enumerate(shapes):
            n = int(np.prod(shp))
            idx = torch.arange(n, dtype=torch.float64)
            ins.append((torch.sin(0.731 * idx * (1 + (idx % 5)) + i) + 0.3 * torch.cos(0.0137 * idx)).to(torch.float32).reshape(shp))
        init = {k: v.clone() for k, v in mod.state_dict().items()}
        rec = {}
        for train in (True, False):
            mod.load_state_dict(init)
            mod.zero_grad()
            r = run_part(mod, ins, train)
            keep = {k: v for k, v in r.items() if not k.startswith("state_after.") or "running" in k}
            for k in list(keep):                              # big conv weight gradients: every 5th element + the L2 norm
                if k.startswith("grad.") and keep[k].dim() == 4 and keep[k].numel() > 20000:
                    gfull = keep.pop(k)
                    keep[k + ".every5"] = gfull.reshape(-1)[::5].clone()
                    keep[k + ".norm"] = gfull.double().norm()
            rec.update({("train." if train else "eval.") + k: v for k, v in keep.items() if train or k in ("y",) or k.startswith("x")})
        save("g14_" + name, **rec)


# ---------------------------------------------------------------- G19 Adam trajectory on WELL-CONDITIONED weights
def g19():
    """[r5] the whole inner loop of train_net (core/scripts/train.py:141-165: forward, loss, zero_grad, backward, Adam step) for TEN
    steps on a network with the reference's own DEFAULT initialisation (nn.Conv2d / nn.BatchNorm2d defaults under a fixed
    torch.manual_seed -- what router.py trains from), not the closed-form det_state weights of G5 whose dead channels make Adam's
    sign(g) updates a coin flip.  A depth-2, base-32 assembly of the reference's DoubleConv / Down / Up / OutConv (RefUNetDepth
    below) keeps the stored initial + final state_dicts small; noise images (no max-pool / ReLU near-ties); lr 1e-4."""
    torch.manual_seed(1234)
    model = add_uncertainty(RefUNetDepth(1, 1, 2, base=32), dict(PARAMS))
    init = {k: v.clone() for k, v in model.state_dict().items()}
    lr, steps, nb, hw = 1e-4, 10, 4, 32
    fix_randomness(19)
    ys = torch.rand(steps, nb, 1, hw, hw)
    xs = ys + 0.1 * torch.randn(steps, nb, 1, hw, hw)
    opt = torch.optim.Adam(model.parameters(), lr=lr)     # train.py:120
    model.train()
    losses = []
    for s in range(steps):
        pred = model(xs[s])                                # train.py:152
        loss = model.loss_fn(pred, ys[s])                  # train.py:153
        losses.append(loss.item())
        opt.zero_grad(); loss.backward(); opt.step()       # train.py:158-162
    model.eval()
    with torch.no_grad():
        probe = model(xs[0])
    rec = {"losses": np.array(losses), "x": xs, "y": ys, "lr": np.array(lr), "probe_out": probe, "depth": np.array(2), "base": np.array(32)}
    for k, v in init.items():
        rec["init." + k] = v
    for k, v in model.state_dict().items():
        rec["final." + k] = v.clone()
    save("g19_adam_trajectory_default_init", **rec)


# ---------------------------------------------------------------- G17 UNets of other depths, from the reference's own parts
class RefUNetDepth(nn.Module):
    """the reference's DoubleConv / Down / Up / OutConv assembled by the recipe of core/models/trunks/unet.py:20-46 for
    a given depth (the reference hard-codes depth 4; SURVEY D3: this assembly is the oracle for BASELINE configs[0]
    "2-level" and configs[3] "deeper UNet").  depth=4 reproduces the reference's UNet state_dict exactly (asserted)."""

    def __init__(self, n_channels_in, n_channels_out, depth, base=64, bilinear=True):
        super().__init__()
        self.n_channels_in, self.n_channels_middle, self.n_channels_out = n_channels_in, 32, n_channels_out
        self.depth = depth
        factor = 2 if bilinear else 1
        width = [base * 2 ** i for i in range(depth + 1)]
        self.inc = DoubleConv(n_channels_in, base)
        for i in range(1, depth + 1):
            setattr(self, f"down{i}", Down(width[i - 1], width[i] // factor if i == depth else width[i]))
        for k in range(1, depth + 1):
            setattr(self, f"up{k}", Up(width[depth - k + 1], width[depth - k] // factor if k < depth else base, bilinear))
        self.out = OutConv(base, self.n_channels_middle)

    def forward(self, x):
        feats = [self.inc(x)]
        for i in range(1, self.depth + 1):
            feats.append(getattr(self, f"down{i}")(feats[-1]))
        h = feats[-1]
        for k in range(1, self.depth + 1):
            h = getattr(self, f"up{k}")(h, feats[self.depth - k])
        return self.out(h)


def g17():
    ref4 = UNet(1, 1).state_dict()
    mine4 = RefUNetDepth(1, 1, 4).state_dict()
    assert list(ref4.keys()) == list(mine4.keys()) and all(ref4[k].shape == mine4[k].shape for k in ref4)
    for depth, hw, nb in ((2, 32, 4), (5, 96, 2)):      # depth 5 at 96x96: 3x3x2 = 18 samples per channel in the deepest BatchNorm
        model = add_uncertainty(RefUNetDepth(1, 1, depth), dict(PARAMS))
        st = om.det_state(1, 1, depth=depth)
        model.load_state_dict(st, strict=True)
        # configs[0]: synthetic Gaussian-denoise, y ~ U[0,1], x = y + 0.1 N(0,1).  Noise images for both depths: the smooth
        # closed-form det_images put many max-pool windows / ReLU inputs on near-ties, where any two fp32 summation orders
        # take different branches and the gradients differ by percents (tests/test_model_gpu.py says the same)
        fix_randomness(depth)
        y = torch.rand(nb, 1, hw, hw)
        x = y + 0.1 * torch.randn(nb, 1, hw, hw)
        model.eval()
        with torch.no_grad():
            out_eval = model(x)
        model.train()
        out_train = model(x)
        loss = model.loss_fn(out_train, y)
        loss.backward()
        sd = model.state_dict()
        rec = dict(x=x, y=y, out_eval=out_eval, out_train=out_train.detach(), loss=loss.detach(), depth=np.array(depth))
        for k, prm in model.named_parameters():           # float64 norm + a strided sample of every parameter gradient
            g = prm.grad.flatten()
            rec["gnorm." + k] = g.double().norm()
            rec["gsample." + k] = g[::max(1, g.numel() // 256)][:256].clone()
        for k, v in sd.items():
            if "running" in k and (".inc." in k or f".up{depth}." in k or f".down{depth}." in k):
                rec["state." + k] = v.clone()
        if depth == 2:                     # a short Adam run (train.py:141-165) + calibration on held-out images
            model.load_state_dict(st, strict=True)
            model.zero_grad()
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            losses = []
            for step in range(5):
                pred = model(x)
                l = model.loss_fn(pred, y)
                losses.append(l.item())
                opt.zero_grad(); l.backward(); opt.step()
            rec["adam_losses"] = np.array(losses)
            fix_randomness(1)
            yc = torch.rand(16, 1, hw, hw)
            xc = yc + 0.1 * torch.randn(16, 1, hw, hw)
            cfg = dict(PARAMS, batch_size=8, num_lambdas=50)
            with quiet(), contextlib.redirect_stderr(io.StringIO()):
                model, table = calibrate_model(model, TensorDataset(xc, yc), cfg)
            rec.update(cal_x=xc, cal_y=yc, cal_table=table, lhat=model.lhat)
        save(f"g17_unet_depth{depth}", **rec)


# ---------------------------------------------------------------- G15 evaluate_from_loss_table
def g15():
    from core.calibration.calibrate_model import evaluate_from_loss_table
    n_img, L = 96, 60
    out, lab = oc.synth_outputs(n_img, 1, 12, 12, seed=8)
    lambdas = torch.linspace(0, 8, L)
    table = torch.stack([oc.losses_at(out, lab, lam) for lam in lambdas], dim=1)       # [N, L], risk falls with lambda
    vals, hb_rows = [], []
    cases = [(40, 0.1, 0.1), (64, 0.1, 0.2), (30, 0.1, 0.05), (48, 0.1, 1e-4)]         # last: no lambda qualifies
    for n, alpha, delta in cases:
        fix_randomness(n)
        with quiet():
            vals.append(float(evaluate_from_loss_table(table, n, alpha, delta)))
        fix_randomness(n)
        perm = torch.randperm(table.shape[0])
        rh = table[perm][:n].mean(dim=0)
        with quiet():
            hb_rows.append(np.array([HB_mu_plus(r, n, delta) for r in rh], dtype=np.float64))   # 0-dim fp32 tensors in, as the reference passes them
    save("g15_evaluate_from_loss_table", table=table, cases=np.array(cases, dtype=np.float64), values=np.array(vals),
         hb=np.stack(hb_rows))


# ---------------------------------------------------------------- G16 WNet
def g16():
    from core.models.trunks.wnet import WNet
    model = add_uncertainty(WNet(1, 1), dict(PARAMS))          # each path takes ONE channel of a 2-channel input (wnet.py:41)
    model.load_state_dict({k: om.det_fill(k, tuple(v.shape)) for k, v in model.state_dict().items() if k != "lhat"}, strict=True)
    fix_randomness(16)
    y = torch.rand(2, 1, 64, 64)
    x = torch.cat([y + 0.1 * torch.randn(2, 1, 64, 64), y + 0.2 * torch.randn(2, 1, 64, 64)], dim=1)
    model.eval()
    with torch.no_grad():
        out_eval = model(x)
    model.train()
    out_train = model(x)
    loss = model.loss_fn(out_train, y)
    loss.backward()
    rec = dict(x=x, y=y, out_eval=out_eval, out_train=out_train.detach(), loss=loss.detach())
    rec["keys"] = np.array(list(model.state_dict().keys()))
    for k, prm in model.named_parameters():
        g = prm.grad.flatten()
        rec["gnorm." + k] = g.double().norm()
        rec["gsample." + k] = g[::max(1, g.numel() // 256)][:256].clone()
    for k, v in model.state_dict().items():
        if "running" in k and ("p1inc" in k or "p2down4" in k or "up4" in k):
            rec["state." + k] = v.clone()
    save("g16_wnet", **rec)


# ---------------------------------------------------------------- G18 fastMRI input pipeline
def g18():
    """the reference's own mask functions, apply_mask, ifft2c, complex_center_crop, complex_abs and UnetDataTransform
    (core/datasets/fastmri/{subsample,transforms,fftc,math_util}.py) on synthetic k-space: masks for several widths and
    seeds; a small slice stored in full; two full-size singlecoil-knee shapes (640x368 and 640x372 -> 320x320) whose
    closed-form k-space (oracle.fastmri.det_kspace) needs no storing, results kept as a 4x-strided sample + checksums."""
    from core.datasets.fastmri import subsample as ref_sub
    from core.datasets.fastmri import transforms as ref_tr
    from oracle import fastmri as ofm
    rec = {}
    # masks: called with a seed (tuple of ords of a file name, transforms.py:289) and un-seeded after rng.seed(k)
    cases = []
    for kind, cls in (("equispaced", ref_sub.EquispacedMaskFunc), ("random", ref_sub.RandomMaskFunc)):
        for cols in (368, 372, 320, 96):
            for fname in ("file1000001.h5", "file1000277.h5", "x"):
                seed = tuple(map(ord, fname))
                m = cls([0.08], [4])((1, cols, 2), seed).reshape(-1).numpy()
                cases.append((kind, cols, fname))
                rec[f"mask.{kind}.{cols}.{fname}"] = m.astype(np.uint8)
    m2 = ref_sub.EquispacedMaskFunc([0.08, 0.04], [4, 8])
    rec["mask.two_rates"] = np.stack([m2((1, 368, 2), (s,)).reshape(-1).numpy() for s in range(6)]).astype(np.uint8)
    # transform, small slice in full
    tr = ref_tr.UnetDataTransform("singlecoil", mask_func=ref_sub.EquispacedMaskFunc([0.08], [4]), use_seed=True)
    ks = ofm.det_kspace(1, 96, 72, salt=1)[0]
    kc = (ks[..., 0] + 1j * ks[..., 1]).numpy()
    target = torch.rand(48, 40, generator=torch.Generator().manual_seed(18)).numpy()      # seeded: the fixture regenerates bit for bit
    image, tgt, _, _, _, _, _ = tr(kc, None, target, {"max": 1.0}, "file1000001.h5", 3)
    rec.update(small_kspace=ks, small_target=target, small_image=image, small_target_out=tgt)
    # FLAIR-203 rule (image narrower than the target's width)
    image_f, _, _, _, _, _, _ = tr(kc, None, None, {"recon_size": (64, 80, 1)}, "file1000001.h5", 3)
    rec["small_image_flair"] = image_f
    # full-size shapes
    for cols in (368, 372):
        ks = ofm.det_kspace(1, 640, cols, salt=cols)[0]
        kc = (ks[..., 0] + 1j * ks[..., 1]).numpy()
        image, _, _, _, _, _, _ = tr(kc, None, np.zeros((320, 320), np.float32), {}, "file1000277.h5", 0)
        rec[f"full{cols}.sample"] = image[::4, ::4].clone()
        rec[f"full{cols}.sum"] = image.double().sum()
        rec[f"full{cols}.sumsq"] = (image.double() ** 2).sum()
        rec[f"full{cols}.max"] = image.max()
    save("g18_fastmri_pipeline", **rec)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, fn in (("g1_g2", g1_g2), ("g3", g3), ("g4", g4), ("g5", g5), ("g19", g19), ("g6", g6), ("g7", g7), ("g8", g8),
                     ("g9_g10", g9_g10), ("g11", g11), ("g12", g12), ("g13", g13), ("g14", g14), ("g15", g15), ("g16", g16), ("g17", g17), ("g18", g18)):
        if not only or name in only:
            fn()
    if "g12_inn" in only:
        g12(("inn",))
    if "g14_convT" in only:
        g14(("up_convT", "up_convT_pad"))
