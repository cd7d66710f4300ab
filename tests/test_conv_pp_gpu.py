"""The 8-wave ping-pong conv kernel (csrc/conv_pp.hip) against the 4-wave kernel it replaces for the shapes it covers: the
two accumulate in the same order (chunk -> tap -> k-step -> MFMA), so outputs AND BatchNorm partial statistics must be
bit-identical; against torch's conv2d the usual bf16 bound.  Covers both tile shapes (16x16; 4 images x 8x8), ragged
image / batch edges, ghost groups (odd item counts), lazy BatchNorm inputs, split inputs (the Up block's concat), split
outputs (its data-gradient), and the three epilogues (store / statistics / folded affine + ReLU)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


def _both(fn):
    from im2im_uq_amd import hip_ops
    try:
        hip_ops.set_option("conv_pp", 0)
        ref = fn()
        hip_ops.set_option("conv_pp", 3)
        got = fn()
    finally:
        hip_ops.set_option("conv_pp", 0)
    torch.cuda.synchronize()
    return ref, got


SHAPES = [  # b, h, w, ci, co
    (3, 40, 40, 64, 128),      # 4 x 8x8 tiles, batch not a multiple of 4
    (2, 96, 80, 128, 128),     # 16x16 tiles
    (1, 72, 88, 128, 256),     # overhanging tiles, two channel blocks
    (5, 20, 20, 256, 128),     # small extent, odd tile count -> a ghost group
    (2, 64, 64, 64, 64),       # 64-wide tile
    (3, 24, 40, 128, 64),      # 64-wide, 8x8 tiles
    (1, 16, 16, 64, 128),      # a single tile: the second group is a ghost
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("epi", ["store", "stats", "affine"])
def test_pp_kernel_is_bit_identical_to_the_4_wave_kernel(shape, epi):
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(hash(shape) % 1000)
    x = torch.randn(b, h, w, ci, device=DEV, generator=g).to(BF16)
    wt = torch.randn(co, ci, 3, 3, device=DEV, generator=g) * 0.05
    bias = torch.randn(co, device=DEV, generator=g)
    wf, _ = nn_ops.pack_weight(wt, BF16)
    ss = torch.stack([torch.rand(co, device=DEV, generator=g) + 0.5, torch.randn(co, device=DEV, generator=g)])

    def run():
        if epi == "store":
            return (nn_ops.conv_fwd(x, wf, bias),)
        if epi == "stats":
            return nn_ops.conv_fwd(x, wf, bias, want_stats=True)
        return (nn_ops.conv_fwd(x, wf, None, scale_shift=ss, relu=True),)
    ref, got = _both(run)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    want = F.conv2d(x.float().permute(0, 3, 1, 2), wt.to(BF16).float(), bias if epi != "affine" else None, padding=1).permute(0, 2, 3, 1)
    if epi == "affine":
        want = torch.relu(want * ss[0] + ss[1])
    err = float((got[0].float() - want).norm() / want.norm())
    assert err < 6e-3, err


@pytest.mark.parametrize("shape", [(2, 48, 64, 128, 128), (3, 40, 40, 256, 128), (2, 32, 48, 128, 64)])
def test_pp_kernel_lazy_and_split_operands(shape):
    """lazy BatchNorm+ReLU inputs (in_ss), the channel-split input of the Up blocks and the split output of their
    data-gradient: bit-identical to the 4-wave kernel."""
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co = shape
    g = torch.Generator(device=DEV).manual_seed(7)
    lo = torch.randn(b, h, w, ci // 2, device=DEV, generator=g).to(BF16)
    hi = torch.randn(b, h, w, ci // 2, device=DEV, generator=g).to(BF16)
    full = torch.randn(b, h, w, ci, device=DEV, generator=g).to(BF16)
    wt = torch.randn(co, ci, 3, 3, device=DEV, generator=g) * 0.05
    wf, wd = nn_ops.pack_weight(wt, BF16)
    ss_full = torch.stack([torch.rand(ci, device=DEV, generator=g) + 0.5, torch.randn(ci, device=DEV, generator=g)]).contiguous()
    ss_lo = torch.stack([torch.rand(ci // 2, device=DEV, generator=g) + 0.5, torch.randn(ci // 2, device=DEV, generator=g)]).contiguous()
    dz = torch.randn(b, h, w, co, device=DEV, generator=g).to(BF16)

    def run():
        out = [nn_ops.conv_fwd(full, wf, None, in_ss=ss_full, want_stats=True)]
        out.append(nn_ops.conv_fwd(lo, wf, None, in_ss=ss_lo, x_hi=hi, want_stats=True))
        out.append(nn_ops.conv_fwd(lo, wf, None, x_hi=hi, in_ss_hi=ss_lo, want_stats=True))
        flat = [t for pair in out for t in pair]
        if ci // 2 % 64 == 0:
            flat += list(nn_ops.conv_fwd(dz, wd, split_out=ci // 2))        # data-gradient into d(skip), d(up)
        return flat
    ref, got = _both(run)
    assert len(ref) == len(got)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    # and the lazy split form equals the materialised concatenation
    a_lo = torch.relu(lo.float() * ss_lo[0] + ss_lo[1]).to(BF16)
    cat = torch.cat([a_lo, hi], dim=-1).contiguous()
    mat = nn_ops.conv_fwd(cat, wf, None, want_stats=True)
    assert torch.equal(mat[0], got[2]) and torch.equal(mat[1], got[3])


def test_pp_kernel_runs_the_unet_step_bit_identically():
    """one bf16 train step of the reference UNet with and without the ping-pong kernel: same loss, same gradients."""
    from im2im_uq_amd import hip_ops, nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    nn_ops.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = add_uncertainty(UNet(1, 1), dict(params)).to(DEV).train()
    x = torch.randn(3, 1, 80, 96, device=DEV)
    y = torch.rand(3, 1, 80, 96, device=DEV)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    outs = []
    try:
        for mode in (0, 3):
            hip_ops.set_option("conv_pp", mode)
            model.load_state_dict(state)
            for p in model.parameters():
                p.grad = None
            loss = model.loss_fn(model(x), y)
            loss.backward()
            nn_ops.join_side_streams()
            torch.cuda.synchronize()
            outs.append((loss.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    finally:
        hip_ops.set_option("conv_pp", 0)
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
