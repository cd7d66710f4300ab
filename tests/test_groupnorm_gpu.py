"""GroupNorm(+ReLU) kernels and the DoubleConv / UNet norm="group" option (named by BASELINE.json:north_star; the reference
itself uses BatchNorm2d, SURVEY D1 -- so the oracle is torch.nn.GroupNorm / F.group_norm in fp32 on the CPU, directly and
through oracle.model's GroupNorm variant of the block).  fp32 mode: tight; bf16: stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32, BF16 = torch.float32, torch.bfloat16
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _restore_dtype():
    from im2im_uq_amd import nn_ops
    yield
    nn_ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(3, 64, 32, 20, 24), (2, 32, 32, 9, 7), (1, 128, 8, 40, 40), (5, 64, 64, 16, 16), (2, 256, 32, 6, 5)])
def test_group_norm_relu_fwd_bwd_vs_torch(case, dt):
    """standalone relu(GroupNorm(x)) of an NHWC tensor: statistics (shifted sums, fp64 merge), per-image coefficients,
    apply; backward dx / dgamma / dbeta.  (B, C, groups, H, W); a channel mean far from zero checks the cancellation-free
    statistics.  Tolerance: fp32 2e-5 relative L2; bf16 1e-2 (storage rounding of x and of the result)."""
    from im2im_uq_amd import nn_ops
    b, c, groups, h, w = case
    x = rnd(b, c, h, w, seed=1) + 3.0 * rnd(1, c, 1, 1, seed=2)
    gamma, beta = 1.0 + 0.3 * rnd(c, seed=3), 0.2 * rnd(c, seed=4)
    gy = rnd(b, c, h, w, seed=5)
    xq = x.to(dt).to(F32).clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.relu(F.group_norm(xq, groups, gr, br, eps=1e-5))
    ref.backward(gy.to(dt).to(F32))
    xd = x.to(DEV).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    out = nn_ops.group_norm_relu(xd, gd, bd, groups, 1e-5)
    out.backward(gy.to(DEV).to(dt))
    t = 2e-5 if dt == F32 else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < t
    assert rel_l2(xd.grad.float().cpu(), xq.grad) < (1e-4 if dt == F32 else 2e-2)
    assert rel_l2(gd.grad.cpu(), gr.grad) < (2e-5 if dt == F32 else 1e-2)
    assert rel_l2(bd.grad.cpu(), br.grad) < (2e-5 if dt == F32 else 1e-2)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 1, 64, 64, 40, 36), (3, 64, 64, 128, 24, 20), (1, 128, 64, 64, 70, 66)])
def test_double_conv_group_norm_vs_torch_block(case, dt):
    """DoubleConv(norm="group") -- conv epilogue statistics per image, GroupNorm applied lazily by the second conv (one
    image per 16x16 tile, per-image coefficients), the block's result materialised -- against the same block built from
    nn.Conv2d / nn.GroupNorm / nn.ReLU on the CPU in fp32: output, input gradient, every parameter gradient."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.trunks.unet_parts import DoubleConv
    b, cin, cmid, cout, h, w = case
    nn_ops.set_compute_dtype(dt)
    torch.manual_seed(0)
    mod = DoubleConv(cin, cout, cmid, norm="group")
    ref = torch.nn.Sequential(torch.nn.Conv2d(cin, cmid, 3, padding=1), torch.nn.GroupNorm(min(32, cmid), cmid), torch.nn.ReLU(),
                              torch.nn.Conv2d(cmid, cout, 3, padding=1), torch.nn.GroupNorm(min(32, cout), cout), torch.nn.ReLU())
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn_like(p))
    ref.load_state_dict({k: v.clone() for k, v in mod.double_conv.state_dict().items()})
    x = rnd(b, cin, h, w, seed=3)
    gy = rnd(b, cout, h, w, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy)
    mod = mod.to(DEV)
    xd = x.to(DEV).requires_grad_(cin > 8)
    yd = mod(xd)
    yd.backward(gy.to(DEV).to(yd.dtype))
    t = 5e-5 if dt == "fp32" else 3e-2
    assert rel_l2(yd.detach().float().cpu(), yr.detach()) < t
    if cin > 8:
        # bf16: z is stored in bf16 before the normalisation, so some ReLU decisions of two stacked layers flip on noise inputs
        assert rel_l2(xd.grad.float().cpu(), xr.grad) < (2e-4 if dt == "fp32" else 1e-1)
    for (name, p), (_, pr) in zip(mod.double_conv.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad.cpu(), pr.grad) < (3e-4 if dt == "fp32" else 8e-2), name


@pytest.mark.parametrize("depth,hw", [(2, 32), (4, 64)])
def test_unet_group_norm_forward_loss_gradients_vs_oracle_fp32(depth, hw):
    """UNet(norm="group") + quantile head end to end (train and eval are the same computation for GroupNorm) against the
    oracle's GroupNorm variant of the network: outputs atol 5e-4, loss 5e-5, parameter gradients 2 % worst / 0.5 % median."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from oracle import model as om
    nn_ops.set_compute_dtype("fp32")
    st = om.det_state(1, 1, depth=depth, norm="group")
    model = add_uncertainty(UNet(1, 1, depth=depth, norm="group"), dict(PARAMS))
    model.load_state_dict(st)
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(31)
    y = torch.rand(3, 1, hw, hw, generator=g)
    x = y + 0.1 * torch.randn(3, 1, hw, hw, generator=g)
    pred = model(x.to(DEV))
    loss = model.loss_fn(pred, y.to(DEV))
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if om.is_param(k)}
    work = dict(st); work.update(leaves)
    ref_pred = om.model_forward(x, work, training=True)
    ref_loss = om.quantile_loss(ref_pred, y, PARAMS)
    ref_loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), ref_pred.detach().numpy(), rtol=0, atol=5e-4)
    assert loss.item() == pytest.approx(ref_loss.item(), rel=5e-5)
    errs = {n: rel_l2(p.grad.cpu(), leaves[n].grad) for n, p in model.named_parameters()}
    assert max(errs.values()) < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    assert float(np.median(list(errs.values()))) < 5e-3
    model.eval()
    with torch.no_grad():
        out = model(x.to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref_pred.detach().numpy(), rtol=0, atol=5e-4)


def test_unet_group_norm_trains_bf16():
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = add_uncertainty(UNet(1, 1, norm="group"), dict(PARAMS)).to(DEV).train()
    opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(5)
    y = torch.rand(4, 1, 96, 96, generator=g).to(DEV)
    x = y + 0.1 * torch.randn(4, 1, 96, 96, generator=g).to(DEV)
    losses = []
    for _ in range(12):
        loss = model.loss_fn(model(x), y)
        losses.append(loss.item())
        opt.zero_grad(); loss.backward(); opt.step()
    assert all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0]
