"""Functional (state_dict-driven) CPU restatement of the reference network.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference builds the network from nn.Module classes; this restatement is a
set of pure functions over a flat ``state`` dict whose keys are exactly the
reference's ``state_dict()`` keys (``baseModel.inc.double_conv.0.weight`` ...
``last_layer.upper.bias``), so the same weights can be fed to the reference,
to this oracle and to the HIP path.  All arithmetic is PyTorch-CPU fp32, the
same substrate the reference's CPU path runs on.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # torch.nn.BatchNorm2d default, core/models/trunks/unet_parts.py:17
BN_MOMENTUM = 0.1   # idem
GN_GROUPS = 32      # DoubleConv(norm="group"): nn.GroupNorm(min(32, C), C), torch's eps default 1e-5

# (prefix, Cin, Cmid, Cout) for the DoubleConv blocks of the UNet, core/models/trunks/unet.py:20-30 with
# bilinear=True.  The reference fixes depth=4, base=64 (nine blocks: inc, down1-4, up1-4); the same recipe
# written for any depth (level i has base*2**i channels, the deepest Down and every Up but the last halve
# theirs) is what BASELINE configs[0] ("2-level") and configs[3] ("deeper UNet") are assembled from.
def unet_blocks(n_in: int, depth: int = 4, base: int = 64) -> List[Tuple[str, int, int, int]]:
    width = [base * 2 ** i for i in range(depth + 1)]
    blocks = [("inc", n_in, base, base)]
    for i in range(1, depth + 1):
        co = width[i] // 2 if i == depth else width[i]
        blocks.append((f"down{i}.maxpool_conv.1", width[i - 1], co, co))
    for k in range(1, depth + 1):
        cin = width[depth - k + 1]
        blocks.append((f"up{k}.conv", cin, cin // 2, width[depth - k] // 2 if k < depth else base))
    return blocks


NUM_SOFTMAX = 50     # experiments/fastmri_test/config.yml num_softmax

# head names per uncertainty type, in the reference's registration (= state_dict) order
HEADS = {
    "quantiles": ("lower", "prediction", "upper"),                    # quantile_layer.py:15-17
    "quantiles_l1": ("lower", "prediction", "upper"),                 # quantile_l1_layer.py:15-17
    "gaussian": ("mean", "variance"),                                 # gaussian_layer.py:12-13
    "residual_magnitude": ("prediction", "residual_magnitude"),       # residual_magnitude_layer.py:12-13
    "residual_magnitude_l1": ("prediction", "residual_magnitude"),    # residual_magnitude_l1_layer.py:12-13
    "inn": ("lower", "prediction", "upper"),                          # inn_layer.py:14-16
}


def state_spec(n_in: int = 1, n_out: int = 1, n_mid: int = 32, utype: str = "quantiles", depth: int = 4,
               base: int = 64, norm: str = "batch") -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) list in the reference's state_dict order
    (unet.py:20-31, unet_parts.py:15-22,90, quantile_layer.py:15-17)."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    for prefix, cin, cmid, cout in unet_blocks(n_in, depth, base):
        p = f"baseModel.{prefix}.double_conv"
        for idx, (ci, co) in ((0, (cin, cmid)), (3, (cmid, cout))):
            spec.append((f"{p}.{idx}.weight", (co, ci, 3, 3)))
            spec.append((f"{p}.{idx}.bias", (co,)))
            spec.append((f"{p}.{idx + 1}.weight", (co,)))
            spec.append((f"{p}.{idx + 1}.bias", (co,)))
            if norm == "group":                 # nn.GroupNorm has no buffers
                continue
            spec.append((f"{p}.{idx + 1}.running_mean", (co,)))
            spec.append((f"{p}.{idx + 1}.running_var", (co,)))
            spec.append((f"{p}.{idx + 1}.num_batches_tracked", ()))
    spec.append(("baseModel.out.conv.weight", (n_mid, base, 1, 1)))
    spec.append(("baseModel.out.conv.bias", (n_mid,)))
    if utype == "softmax":                                             # softmax_layer.py:11 (one conv to num_softmax classes)
        spec.append(("last_layer.output_layers.0.weight", (NUM_SOFTMAX, n_mid, 3, 3)))
        spec.append(("last_layer.output_layers.0.bias", (NUM_SOFTMAX,)))
        return spec
    for head in HEADS[utype]:
        spec.append((f"last_layer.{head}.weight", (n_out, n_mid, 3, 3)))
        spec.append((f"last_layer.{head}.bias", (n_out,)))
    return spec


def is_param(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


# ---- bf16 storage emulation ---------------------------------------------------------------------
# The HIP path's throughput mode keeps activations (and the gradients flowing between layers) in bf16 and
# feeds bf16 operands to the MFMA units with fp32 accumulation -- the numerics of torch.autocast(bfloat16)
# applied to the reference.  `emulate_bf16=True` restates the reference with a bf16 round-trip at exactly
# the points where the kernels store a bf16 tensor, so the bf16 mode can be checked against the reference
# arithmetic *at that precision* (tests/test_model_gpu.py), separately from the bf16-vs-fp32 distance.
class _RoundFwd(torch.autograd.Function):          # operand rounding (weights): value rounded, gradient untouched
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _store(x, emulate):
    """a tensor the kernels write in the compute dtype: value and incoming gradient both pass through bf16."""
    return x.to(torch.bfloat16).to(torch.float32) if emulate else x


def _operand(w, emulate):
    return _RoundFwd.apply(w) if emulate else w


# ---- fp8 operand emulation (the 'fp8' compute mode, csrc/conv_fp8.hip) -------------------------------------------------
# Forward 3x3 convolutions with Ci % 64 == 0 and Co % 64 == 0 take OCP e4m3 operands: the input activation is scaled by
# 2^4, clamped to +-448 and rounded to e4m3; each output channel's weights are divided by the power of two that maps the
# channel's max |w| into [128, 256) and rounded to e4m3; products accumulate in fp32.  Values pass through rounded,
# gradients pass through untouched (the backward kernels run on the bf16 tensors).
FP8_MAX = 448.0


class _Fp8Fwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        q = (x * scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32)
        return q / scale

    @staticmethod
    def backward(ctx, g):
        return g, None


def fp8_activation(x):
    return _Fp8Fwd.apply(x, 16.0)


def fp8_weight(w):
    amax = w.detach().abs().amax(dim=(1, 2, 3), keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) + 1          # amax = f * 2^e, f in [0.5, 1)  (frexp)
    scale = torch.where(amax > 0, torch.exp2(e - 8), torch.ones_like(amax))
    return _Fp8Fwd.apply(w, 1.0 / scale)


def fp8_eligible(w):
    return w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0 and w.shape[2] == 3


def double_conv(x, state, prefix, training, emulate_bf16=False):
    """(conv3x3 pad1 + bias -> BatchNorm2d -> ReLU) x 2, unet_parts.py:15-25.  emulate_bf16 = "fp8": bf16 storage as
    for True, plus e4m3 operands in the eligible forward convolutions."""
    p = f"baseModel.{prefix}.double_conv"
    for idx in (0, 3):
        w = state[f"{p}.{idx}.weight"]
        if emulate_bf16 == "fp8" and fp8_eligible(w):
            w = fp8_weight(w)
            x = fp8_activation(x)
        elif w.shape[1] > 8:                    # the <=8-channel first conv runs on fp32 weights in the kernels
            w = _operand(w, emulate_bf16)
        x = F.conv2d(x, w, state[f"{p}.{idx}.bias"], padding=1)
        if f"{p}.{idx + 1}.running_mean" not in state:
            # the GroupNorm variant of the block (north-star extra, not in the reference: SURVEY D1): nn.GroupNorm(min(32, C), C)
            x = _store(x, emulate_bf16)
            x = F.group_norm(x, min(GN_GROUPS, x.shape[1]), state[f"{p}.{idx + 1}.weight"], state[f"{p}.{idx + 1}.bias"], eps=BN_EPS)
            x = _store(F.relu(x), emulate_bf16)
            continue
        if training:
            x = _store(x, emulate_bf16)         # train mode stores the pre-BN conv output; eval folds BN into the conv
        x = F.batch_norm(x, state[f"{p}.{idx + 1}.running_mean"], state[f"{p}.{idx + 1}.running_var"],
                         state[f"{p}.{idx + 1}.weight"], state[f"{p}.{idx + 1}.bias"],
                         training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
        if training:
            state[f"{p}.{idx + 1}.num_batches_tracked"] += 1
        x = _store(F.relu(x), emulate_bf16)
    return x


def up_block(x_deep, x_skip, state, prefix, training, emulate_bf16=False):
    """bilinear x2 (align_corners=True) [or ConvTranspose2d(k=2, s=2) when the state holds `<block>.up.weight`] -> zero-pad
    to skip -> cat([skip, up]) -> DoubleConv, unet_parts.py:50-69."""
    if f"baseModel.{prefix[:-len('.conv')] if prefix.endswith('.conv') else prefix}.up.weight" in state:      # bilinear=False, unet_parts.py:53
        stem = prefix[:-len(".conv")] if prefix.endswith(".conv") else prefix
        u = F.conv_transpose2d(x_deep, state[f"baseModel.{stem}.up.weight"], state[f"baseModel.{stem}.up.bias"], stride=2)
    else:
        u = F.interpolate(x_deep, scale_factor=2, mode="bilinear", align_corners=True)
    dy = x_skip.shape[2] - u.shape[2]
    dx = x_skip.shape[3] - u.shape[3]
    u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    cat = _store(torch.cat([x_skip, u], dim=1), emulate_bf16)
    return double_conv(cat, state, prefix, training, emulate_bf16)


def unet_depth(state) -> int:
    """number of Down blocks the state describes (4 for the reference's UNet)."""
    d = 0
    while f"baseModel.down{d + 1}.maxpool_conv.1.double_conv.0.weight" in state:
        d += 1
    return d


def unet_forward(x, state, training: bool, emulate_bf16=False):
    """core/models/trunks/unet.py:33-46 (for the depth the state describes)."""
    depth = unet_depth(state)
    x1 = double_conv(x, state, "inc", training, emulate_bf16)
    skips = [x1]
    h = x1
    for i in range(1, depth + 1):
        h = double_conv(F.max_pool2d(h, 2), state, f"down{i}.maxpool_conv.1", training, emulate_bf16)  # unet_parts.py:33-40
        skips.append(h)
    for i in range(1, depth + 1):
        h = up_block(h, skips[depth - i], state, f"up{i}.conv", training, emulate_bf16)
    out = F.conv2d(h, _operand(state["baseModel.out.conv.weight"], emulate_bf16), state["baseModel.out.conv.bias"])  # unet_parts.py:90-94
    return _store(out, emulate_bf16)


def wnet_forward(x, state, training: bool, emulate_bf16=False):
    """core/models/trunks/wnet.py:40-66: two half-width encoders, one per input channel, concatenated level by level and
    decoded by the UNet's Up blocks."""
    feats = {}
    for path, ch in (("p1", 0), ("p2", 1)):
        h = double_conv(x[:, ch:ch + 1], state, path + "inc", training, emulate_bf16)
        feats[path] = [h]
        for i in range(1, 5):
            h = double_conv(F.max_pool2d(h, 2), state, f"{path}down{i}.maxpool_conv.1", training, emulate_bf16)
            feats[path].append(h)
    join = [torch.cat((a, b), dim=1) for a, b in zip(feats["p1"], feats["p2"])]
    h = join[4]
    for i in range(1, 5):
        h = up_block(h, join[4 - i], state, f"up{i}.conv", training, emulate_bf16)
    out = F.conv2d(h, _operand(state["baseModel.out.conv.weight"], emulate_bf16), state["baseModel.out.conv.bias"])
    return _store(out, emulate_bf16)


def quantile_heads(feat, state):
    """three 3x3 heads stacked on a new dim 1 -> [B,3,C,H,W], quantile_layer.py:19-21."""
    outs = [F.conv2d(feat, state[f"last_layer.{h}.weight"], state[f"last_layer.{h}.bias"], padding=1)
            for h in ("lower", "prediction", "upper")]
    return torch.stack(outs, dim=1)


def final_layer(feat, state, utype="quantiles"):
    """the final layers' forward: 3x3 heads stacked on a new dim 1; ReLU on the gaussian variance
    (gaussian_layer.py:15-17), abs on the residual magnitude (residual_magnitude_layer.py:15-17)."""
    if utype == "softmax":                                             # softmax_layer.py:13-14, n_channels_out = 1
        return F.conv2d(feat, state["last_layer.output_layers.0.weight"], state["last_layer.output_layers.0.bias"],
                        padding=1).unsqueeze(2)
    outs = [F.conv2d(feat, state[f"last_layer.{h}.weight"], state[f"last_layer.{h}.bias"], padding=1) for h in HEADS[utype]]
    if utype == "gaussian":
        outs[1] = torch.relu(outs[1])
    elif utype in ("residual_magnitude", "residual_magnitude_l1"):
        outs[1] = outs[1].abs()
    return torch.stack(outs, dim=1)


def model_forward(x, state, training: bool = False, emulate_bf16: bool = False, utype: str = "quantiles"):
    """ModelWithUncertainty.forward, core/models/add_uncertainty.py:25-27."""
    return final_layer(unet_forward(x, state, training, emulate_bf16), state, utype)


def pinball(output, target, q: float):
    """core/models/losses/pinball.py:12-26 (mean reduction).
    loss_i = q*|e| if e<0 ; (1-q)*|e| if e>0 ; 0 if e==0, e = output-target."""
    err = output - target
    a = err.abs()
    loss = torch.where(err < 0, q * a, torch.where(err > 0, (1 - q) * a, torch.zeros_like(a)))
    return loss.mean()


def quantile_loss(pred, target, params):
    """quantile_regression_loss_fn, quantile_layer.py:23-32.  pred [B,3,C,H,W], target [B,C,H,W]."""
    t = target.squeeze()
    return (params["q_lo_weight"] * pinball(pred[:, 0].squeeze(), t, params["q_lo"])
            + params["q_hi_weight"] * pinball(pred[:, 2].squeeze(), t, params["q_hi"])
            + params["mse_weight"] * F.mse_loss(pred[:, 1].squeeze(), t))


def uq_loss(pred, target, params, utype="quantiles"):
    """the train loss of each final layer (all mean-reduced):
    quantiles quantile_layer.py:23-32; quantiles_l1 quantile_l1_layer.py:23-32 (L1 point loss); gaussian
    gaussian_layer.py:19-23 (nn.GaussianNLLLoss, eps 1e-6, clamp under no_grad); residual_magnitude(_l1)
    residual_magnitude(_l1)_layer.py:19-25 (MSE|L1 point loss + MSE of the magnitude against |y - pred|)."""
    t = target.squeeze()
    if utype == "quantiles":
        return quantile_loss(pred, target, params)
    if utype == "quantiles_l1":
        return (params["q_lo_weight"] * pinball(pred[:, 0].squeeze(), t, params["q_lo"])
                + params["q_hi_weight"] * pinball(pred[:, 2].squeeze(), t, params["q_hi"])
                + params["mse_weight"] * F.l1_loss(pred[:, 1].squeeze(), t))
    if utype == "gaussian":
        mean, var = pred[:, 0].squeeze(), pred[:, 1].squeeze()
        v = var.clone()
        with torch.no_grad():
            v.clamp_(min=1e-6)
        return (0.5 * (torch.log(v) + (mean - t) ** 2 / v)).mean()
    if utype in ("residual_magnitude", "residual_magnitude_l1"):
        p0, m = pred[:, 0].squeeze(), pred[:, 1].squeeze()
        first = F.mse_loss(p0, t) if utype == "residual_magnitude" else F.l1_loss(p0, t)
        return first + F.mse_loss(m, (t - p0).abs())
    if utype == "inn":                                                 # inn_layer.py:22-28, losses/inn.py:12-21
        lo, mid, hi = pred[:, 0].squeeze(), pred[:, 1].squeeze(), pred[:, 2].squeeze()
        interval = torch.relu(t - hi).square() + torch.relu(lo - t).square() + params["beta"] * torch.abs(hi - lo)
        return F.mse_loss(mid, t) + interval.mean()
    if utype == "softmax":                                             # softmax_layer.py:15-25
        k = pred.shape[1]
        classes = torch.linspace(0, 1, k)
        idx = torch.bucketize(target, classes, right=False)
        idx[idx >= k] = k - 1
        return F.cross_entropy(pred, idx)
    raise NotImplementedError(utype)


# ----------------------------------------------------------------------------
# deterministic closed-form initialisation shared by the golden generator, the
# oracle tests and the HIP tests (so 69 MB of weights never need storing).
def det_fill(key: str, shape: Tuple[int, ...]) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= s
    seed = sum((i + 1) * ord(c) for i, c in enumerate(key)) % 9973
    idx = torch.arange(n, dtype=torch.float64)
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if key.endswith("running_mean"):
        v = 0.05 * torch.sin(0.91 * idx + seed)
    elif key.endswith("running_var"):
        v = 1.0 + 0.25 * torch.cos(0.53 * idx + seed)
    elif len(shape) == 4:  # conv weight: ~kaiming-uniform scale 1/sqrt(fan_in)
        fan_in = shape[1] * shape[2] * shape[3]
        v = math.sqrt(3.0 / fan_in) * torch.sin(0.618 * idx * (1 + (idx % 7)) + seed)
    elif ".double_conv.1." in key or ".double_conv.4." in key:
        if key.endswith("weight"):   # BN gamma
            v = 1.0 + 0.2 * torch.cos(1.3 * idx + seed)
        else:                        # BN beta
            v = 0.1 * torch.sin(0.7 * idx + seed)
    else:  # conv bias
        v = 0.05 * torch.cos(0.37 * idx + seed)
    return v.to(torch.float32).reshape(shape)


def det_state(n_in: int = 1, n_out: int = 1, utype: str = "quantiles", depth: int = 4, base: int = 64,
              norm: str = "batch") -> Dict[str, torch.Tensor]:
    return {k: det_fill(k, shp) for k, shp in state_spec(n_in, n_out, utype=utype, depth=depth, base=base, norm=norm)}


def default_init_state(n_in: int = 1, n_out: int = 1, seed: int = 0, utype: str = "quantiles", depth: int = 4,
                       base: int = 64) -> Dict[str, torch.Tensor]:
    """The initialisation the reference's modules get from torch when router.py builds them (core/models/trunks/unet_parts.py:
    nn.Conv2d / nn.BatchNorm2d defaults; finallayers/quantile_layer.py:11-13): conv weights kaiming_uniform_(a=sqrt(5)), i.e.
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); conv biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm gamma 1, beta 0, running
    mean 0, running variance 1, no batches tracked.  Drawn from a seeded generator key by key: the same distribution as a
    freshly constructed reference model (not its bits -- those depend on torch's construction order)."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    spec = state_spec(n_in, n_out, utype=utype, depth=depth, base=base)
    fan = {}
    for k, shp in spec:
        if len(shp) == 4:
            fan[k[:-len("weight")]] = shp[1] * shp[2] * shp[3]
    for k, shp in spec:
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            st[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            st[k] = torch.ones(shp)
        elif len(shp) == 4:
            b = 1.0 / math.sqrt(shp[1] * shp[2] * shp[3])
            st[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        elif k[:-len("bias")] in fan and k.endswith("bias"):             # a convolution's bias
            b = 1.0 / math.sqrt(fan[k[:-len("bias")]])
            st[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        elif k.endswith("weight"):                                         # BatchNorm gamma
            st[k] = torch.ones(shp)
        else:                                                              # BatchNorm beta
            st[k] = torch.zeros(shp)
    return st


def det_images(n: int, c: int, h: int, w: int, salt: int = 0):
    """deterministic (input, target) pair: target in [0,1], input = target + structured noise."""
    idx = torch.arange(n * c * h * w, dtype=torch.float64).reshape(n, c, h, w)
    y = 0.5 + 0.5 * torch.sin(0.013 * idx + 0.7 * salt) * torch.cos(0.0071 * idx * (1 + salt))
    x = y + 0.1 * torch.sin(1.7 * idx + salt)
    return x.to(torch.float32), y.to(torch.float32)


# ----------------------------------------------------------------------------
def train_steps(state, batches, params, lr: float):
    """The inner loop of train_net, core/scripts/train.py:141-165: Adam with torch
    defaults (train.py:120), loss -> zero_grad -> backward -> step.  ``state`` is
    updated in place; returns the list of per-step losses."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in state.items() if is_param(k)}
    work = dict(state)
    work.update(leaves)
    opt = torch.optim.Adam(list(leaves.values()), lr=lr)
    losses = []
    for x, y in batches:
        pred = model_forward(x, work, training=True)
        loss = quantile_loss(pred, y, params)
        losses.append(float(loss.item()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    for k, v in leaves.items():
        state[k] = v.detach()
    return losses
