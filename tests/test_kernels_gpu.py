"""GPU parity of each training-path HIP kernel against a plain PyTorch-CPU fp32 restatement of the same op
(the substrate the reference runs on).  fp32 mode: tight tolerances; bf16 mode: tolerance stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32, BF16 = torch.float32, torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def tol(dt):
    return 2e-5 if dt == F32 else 1.5e-2


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dt):
    """quantise a CPU reference operand the way the kernel will see it."""
    return t.to(dt).to(F32)


def merged_moments(stats):
    """merge per-tile (mean, M2, count) rows [R,3,C] -> total (count, mean, M2) per channel in float64 (Chan et al.)."""
    st = stats.double().cpu()
    n = st[:, 2].sum(0)
    mean = (st[:, 2] * st[:, 0]).sum(0) / n
    m2 = st[:, 1].sum(0) + (st[:, 2] * (st[:, 0] - mean) ** 2).sum(0)
    return n, mean, m2


CONV_CASES = [
    # B, H, W, Ci, Co, taps
    (2, 20, 24, 64, 64, 9),       # 8x8 tiles, overhang in both dims
    (1, 40, 40, 128, 128, 9),     # 8x8 tiles, BN=128
    (2, 70, 66, 32, 32, 9),       # 16x16 tiles, BN=32, overhang
    (1, 80, 72, 64, 128, 9),      # 16x16 tiles, BN=128
    (1, 64, 64, 96, 64, 9),       # 3 K-chunks
    (2, 33, 17, 64, 32, 1),       # 1x1, small
    (1, 96, 80, 64, 32, 1),       # 1x1, 16x16 tiles
    (1, 9, 7, 256, 256, 9),       # deep level style
    (16, 40, 40, 128, 512, 9),    # [r3] 4-image 8x8 tiles (>= 384 workgroups: the shape class of the batch-78 deep levels; smaller
                                  #      launches now take 2-image tiles, e.g. the first two cases)
    (3, 64, 72, 64, 64, 9),       # [r3] 32x16 tiles of the 64-output-channel layers (H % 32 == 0), overhang in W, odd batch
    (2, 96, 80, 128, 64, 9),      # [r3] 32x16 tiles, 4 K-chunks; data-gradient = 64 -> 128 on the direct-weight 128-wide tile
]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dt):
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co, taps = case
    k = 3 if taps == 9 else 1
    x = rnd(b, ci, h, w, seed=1)
    wt = rnd(co, ci, k, k, seed=2, scale=(ci * taps) ** -0.5)
    bias = rnd(co, seed=3, scale=0.1)
    xq, wq = q(x, dt).requires_grad_(True), q(wt, dt).requires_grad_(True)
    ref = F.conv2d(xq, wq, bias, padding=k // 2)
    x_d = x.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    wf, wd = nn_ops.pack_weight(wt.to(DEV), dt)
    y, stats = nn_ops.conv_fwd(x_d, wf, bias.to(DEV), want_stats=True)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(got, ref.detach()) < tol(dt)
    # BatchNorm partial statistics of the STORED values
    n, mean, m2 = merged_moments(stats)
    yst = y.double().cpu().reshape(-1, co)
    assert float((n - yst.shape[0]).abs().max()) == 0.0
    np.testing.assert_allclose(mean.numpy(), yst.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m2.numpy(), ((yst - yst.mean(0)) ** 2).sum(0).numpy(), rtol=2e-5)
    # folded affine + relu epilogue
    ss = torch.stack([rnd(co, seed=4).abs() + 0.5, rnd(co, seed=5)]).to(DEV)
    y2 = nn_ops.conv_fwd(x_d, wf, None, ss, relu=True).float().cpu().permute(0, 3, 1, 2)
    ref2 = F.relu(F.conv2d(xq, wq, None, padding=k // 2) * ss[0].cpu()[None, :, None, None] + ss[1].cpu()[None, :, None, None])
    assert rel_l2(y2, ref2.detach()) < tol(dt)
    # backward: dgrad = forward kernel with the flipped/transposed weights, wgrad kernel
    gy = rnd(b, co, h, w, seed=6)
    gyq = q(gy, dt)
    ref.backward(gyq)
    gy_d = gy.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    dx = nn_ops.conv_fwd(gy_d, wd).float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(dx, xq.grad) < tol(dt)
    if ci % 64 == 0:
        dw = nn_ops.conv_wgrad(x_d, gy_d, taps).cpu().view(co, ci, k, k)
        assert rel_l2(dw, wq.grad) < tol(dt)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(2, 20, 24, 64, 64), (1, 70, 66, 64, 128), (3, 16, 16, 128, 32), (1, 80, 40, 256, 64)])
def test_split_operand_conv_is_bit_identical_to_concatenated(case, dt):
    """The Up block's cat([skip, up], 1) (unet_parts.py:68) is not materialised: forward and weight-gradient read the
    two halves from separate tensors (each with its own lazy BatchNorm coefficients), the data-gradient writes
    d(skip) and d(up) separately.  Same arithmetic in the same order as on the concatenated tensor => identical bits."""
    from im2im_uq_amd import nn_ops
    b, h, w, c, co = case
    lo = rnd(b, h, w, c, seed=1).to(dt).to(DEV)                    # pre-BatchNorm z of the skip (lazy)
    hi = rnd(b, h, w, c, seed=2).abs().to(dt).to(DEV)              # upsampled half: a plain, non-negative activation
    ss_lo = torch.stack([1.0 + 0.1 * rnd(c, seed=3), 0.2 * rnd(c, seed=4)]).to(DEV)
    wt = rnd(co, 2 * c, 3, 3, seed=5, scale=0.05).to(DEV)
    bias = rnd(co, seed=6).to(DEV)
    wf, wd = nn_ops.pack_weight(wt, dt)
    # concatenated reference operand: materialise the lazy half exactly as bn_relu_apply does, then cat
    cat = torch.cat([nn_ops.bn_relu_apply(lo, ss_lo), hi], dim=3).contiguous()
    y_ref, st_ref = nn_ops.conv_fwd(cat, wf, bias, want_stats=True)
    y, st = nn_ops.conv_fwd(lo, wf, bias, want_stats=True, in_ss=ss_lo, x_hi=hi)
    assert torch.equal(y, y_ref) and torch.equal(st, st_ref)
    # both halves lazy
    ss_hi = torch.stack([1.0 + 0.1 * rnd(c, seed=7), 0.2 * rnd(c, seed=8)]).to(DEV)
    cat2 = torch.cat([nn_ops.bn_relu_apply(lo, ss_lo), nn_ops.bn_relu_apply(hi, ss_hi)], dim=3).contiguous()
    assert torch.equal(nn_ops.conv_fwd(lo, wf, bias, in_ss=ss_lo, x_hi=hi, in_ss_hi=ss_hi), nn_ops.conv_fwd(cat2, wf, bias))
    # eval epilogue (folded BatchNorm + ReLU) on plain halves
    fold = torch.stack([1.0 + 0.1 * rnd(co, seed=9), 0.1 * rnd(co, seed=10)]).to(DEV)
    cat3 = torch.cat([hi, hi.flip(0)], dim=3).contiguous()
    assert torch.equal(nn_ops.conv_fwd(hi, wf, None, fold, relu=True, x_hi=hi.flip(0).contiguous()),
                       nn_ops.conv_fwd(cat3, wf, None, fold, relu=True))
    # weight gradient
    dz = rnd(b, h, w, co, seed=11).to(dt).to(DEV)
    assert torch.equal(nn_ops.conv_wgrad(lo, dz, 9, x_ss=ss_lo, x_hi=hi), nn_ops.conv_wgrad(cat, dz, 9))
    assert torch.equal(nn_ops.conv_wgrad(lo, dz, 9, x_ss=ss_lo, x_hi=hi, x_ss_hi=ss_hi), nn_ops.conv_wgrad(cat2, dz, 9))
    # data gradient split over two destination tensors
    dx_ref = nn_ops.conv_fwd(dz, wd)
    dx_lo, dx_hi = nn_ops.conv_fwd(dz, wd, split_out=c)
    assert torch.equal(dx_lo, dx_ref[..., :c]) and torch.equal(dx_hi, dx_ref[..., c:])


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(2, 9, 7, 64, 19, 15), (1, 20, 20, 128, 40, 40), (2, 33, 17, 64, 67, 36), (1, 1, 1, 64, 2, 2),
                                  (1, 2, 3, 64, 7, 9), (3, 5, 40, 192, 10, 80), (1, 6, 6, 32, 13, 12),
                                  (4, 160, 160, 64, 320, 320), (2, 20, 20, 512, 40, 40)])     # the up4 / up1 shapes of the BASELINE network
def test_upsample2x_alone_matches_concat_kernel_and_torch(dt, case):
    """im2im_upsample2x_concat_* with Cs = 0: bilinear x2 (align_corners) + zero pad without the skip copy.  Channel counts
    that are multiples of 64 take the tiled kernels (up2x_*_tiled_kernel): forward bit-identical to the row-walking kernel,
    tiles overhanging the image, padding on both sides, single-pixel sources."""
    from im2im_uq_amd import nn_ops
    b, h, w, c, hh, ww = case
    deep = rnd(b, c, h, w, seed=1).abs()
    d_dev = deep.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV).permute(0, 3, 1, 2).requires_grad_(True)
    skip = torch.zeros(b, c, hh, ww).permute(0, 2, 3, 1).contiguous().to(dt).to(DEV).permute(0, 3, 1, 2)
    up = nn_ops.Upsample2x.apply(d_dev, hh, ww)
    cat = nn_ops.UpsampleConcat.apply(d_dev.detach(), skip)
    assert torch.equal(up, cat[:, c:])
    ref = F.interpolate(q(deep, dt), scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.pad(ref, [(ww - 2 * w) // 2, ww - 2 * w - (ww - 2 * w) // 2, (hh - 2 * h) // 2, hh - 2 * h - (hh - 2 * h) // 2])
    assert rel_l2(up.detach().float().cpu(), ref) < tol(dt)
    g = rnd(b, c, hh, ww, seed=3)
    g_dev = g.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV).permute(0, 3, 1, 2)
    up.backward(g_dev)
    dref = deep.clone().requires_grad_(True)
    r = F.pad(F.interpolate(dref, scale_factor=2, mode="bilinear", align_corners=True),
              [(ww - 2 * w) // 2, ww - 2 * w - (ww - 2 * w) // 2, (hh - 2 * h) // 2, hh - 2 * h - (hh - 2 * h) // 2])
    r.backward(q(g, dt))
    assert rel_l2(d_dev.grad.float().cpu(), dref.grad) < tol(dt)


def test_upsample2x_tiled_kernels_random_geometries():
    """40 seeded random geometries (source 1..37 px per side, 0..3 px of padding on either side, 64 / 128 channels, fp32 so
    that the comparison is tight) through the tiled upsampling kernels against torch: forward and gradient.  Tiles
    overhanging every edge, sources narrower than a tile, odd pads."""
    from im2im_uq_amd import nn_ops
    rng = np.random.default_rng(20260929)
    for it in range(40):
        b, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 38)), int(rng.integers(1, 38))
        c = int(rng.choice([64, 128]))
        hh, ww = 2 * h + int(rng.integers(0, 4)), 2 * w + int(rng.integers(0, 4))
        deep = rnd(b, c, h, w, seed=100 + it)
        d_dev = deep.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2).requires_grad_(True)
        up = nn_ops.Upsample2x.apply(d_dev, hh, ww)
        dref = deep.clone().requires_grad_(True)
        pad = [(ww - 2 * w) // 2, ww - 2 * w - (ww - 2 * w) // 2, (hh - 2 * h) // 2, hh - 2 * h - (hh - 2 * h) // 2]
        ref = F.pad(F.interpolate(dref, scale_factor=2, mode="bilinear", align_corners=True), pad)
        assert rel_l2(up.detach().cpu(), ref.detach()) < 2e-6, (it, b, h, w, c, hh, ww)
        g = rnd(b, c, hh, ww, seed=500 + it)
        up.backward(g.permute(0, 2, 3, 1).contiguous().to(DEV).permute(0, 3, 1, 2))
        ref.backward(g)
        assert rel_l2(d_dev.grad.cpu(), dref.grad) < 2e-6, (it, b, h, w, c, hh, ww)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(2, 20, 24, 64, 64), (1, 70, 66, 32, 128), (3, 16, 16, 128, 32), (1, 80, 40, 64, 256)])
def test_dgrad_with_batchnorm_backward_sums(case, dt):
    """im2im_conv_dgrad_bn: dx identical to the plain data-gradient, partial sums == sum(g), sum(g*xhat) of that dx."""
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co = case                                        # dz has ci channels, dx / z have co
    dz = rnd(b, h, w, ci, seed=1).to(dt).to(DEV)
    wt = rnd(ci, co, 3, 3, seed=2, scale=0.05).to(DEV)          # the forward conv's weight [Co_fwd = ci][Ci_fwd = co]
    _, wd = nn_ops.pack_weight(wt, dt)
    z = rnd(b, h, w, co, seed=3).to(dt).to(DEV)
    ss = torch.stack([1.0 + 0.2 * rnd(co, seed=4), 0.3 * rnd(co, seed=5)]).to(DEV)
    mi = torch.stack([0.1 * rnd(co, seed=6), 1.0 + 0.2 * rnd(co, seed=7).abs()]).to(DEV)
    dx, partial = nn_ops.conv_dgrad_bn(dz, wd, z, ss, mi)
    assert torch.equal(dx, nn_ops.conv_fwd(dz, wd))
    g = dx.double() * ((z.double() * ss[0].double() + ss[1].double()) > 0)
    mask32 = (z.float() * ss[0] + ss[1]) > 0                    # the kernel's own fp32 test decides the mask
    g = dx.double() * mask32
    xhat = (z.double() - mi[0].double()) * mi[1].double()
    s1, s2 = g.sum(dim=(0, 1, 2)), (g * xhat).sum(dim=(0, 1, 2))
    got = partial.double().sum(0)
    t = 2e-5 if dt == F32 else 2e-5
    assert float((got[0] - s1).abs().max() / (s1.abs().max() + 1e-30)) < t
    assert float((got[1] - s2).abs().max() / (s2.abs().max() + 1e-30)) < t


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 1, 37, 29, 64), (1, 2, 48, 48, 64), (2, 3, 16, 20, 32), (1, 6, 33, 18, 32)])
def test_smallconv_family(shape, dt):
    from im2im_uq_amd import nn_ops
    b, cs, h, w, cl = shape
    # s2l forward (first conv): x NCHW fp32 -> NHWC
    x = rnd(b, cs, h, w, seed=1)
    wt = rnd(cl, cs, 3, 3, seed=2, scale=0.3)
    bias = rnd(cl, seed=3, scale=0.1)
    x_r, w_r = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = F.conv2d(x_r, w_r, bias, padding=1)
    _, wd = nn_ops.pack_weight(wt.to(DEV), F32)
    z, stats = nn_ops.smallconv_s2l(x.to(DEV), wd, bias.to(DEV), None, cl, dt, flip=True, want_stats=True)
    got = z.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(got, ref.detach()) < (1e-5 if dt == F32 else 5e-3)
    zs = z.double().cpu().reshape(-1, cl)
    n, mean, m2 = merged_moments(stats)
    assert float((n - zs.shape[0]).abs().max()) == 0.0
    np.testing.assert_allclose(mean.numpy(), zs.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m2.numpy(), ((zs - zs.mean(0)) ** 2).sum(0).numpy(), rtol=2e-5)
    # its weight gradient (l_major)
    gz = rnd(b, cl, h, w, seed=4)
    ref.backward(q(gz, dt))
    gz_d = gz.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    dw, _ = nn_ops.smallconv_wgrad(x.to(DEV), gz_d, l_major=True, want_bias=False)
    assert rel_l2(dw.cpu().view(cl, cs, 3, 3), w_r.grad) < 1e-4
    # l2s forward (heads): NHWC feat -> NCHW planes, and its backward pair
    feat = rnd(b, cl, h, w, seed=5)
    wh = rnd(cs, cl, 3, 3, seed=6, scale=0.2)
    bh = rnd(cs, seed=7, scale=0.1)
    f_r, wh_r, bh_r = q(feat, dt).requires_grad_(True), wh.clone().requires_grad_(True), bh.clone().requires_grad_(True)
    refh = F.conv2d(f_r, wh_r, bh_r, padding=1)
    wf, _ = nn_ops.pack_weight(wh.to(DEV), F32, want_wd=False)
    f_d = feat.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    out = nn_ops.smallconv_l2s(f_d, wf, bh.to(DEV), cs)
    assert rel_l2(out.cpu(), refh.detach()) < 1e-5
    go = rnd(b, cs, h, w, seed=8)
    refh.backward(go)
    dfeat = nn_ops.smallconv_s2l(go.to(DEV), wf, None, None, cl, dt, flip=True).float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(dfeat, f_r.grad) < (1e-5 if dt == F32 else 5e-3)
    dwh, dbh = nn_ops.smallconv_wgrad(go.to(DEV), f_d, l_major=False, want_bias=True)
    assert rel_l2(dwh.cpu().view(cs, cl, 3, 3), wh_r.grad) < 1e-4
    assert rel_l2(dbh.cpu(), bh_r.grad) < 1e-5


def test_batchnorm_statistics_survive_a_large_channel_offset():
    """|mean| >> sigma: E[z^2] - E[z]^2 in fp32 would lose mean^2/var ~ 1e6 of its digits; the epilogue's per-tile
    (mean, M2, count) partials merged in fp64 keep the variance to fp32 accuracy (torch's CPU BatchNorm accumulates in
    double, so the reference does too)."""
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co = 2, 40, 48, 32, 64
    x = rnd(b, h, w, ci, seed=1).to(DEV)
    wt = rnd(co, ci, 3, 3, seed=2, scale=0.02).to(DEV)
    bias = torch.full((co,), 300.0).to(DEV)                       # conv output ~ 300 +- 0.3
    wf, _ = nn_ops.pack_weight(wt, F32, want_wd=False)
    z, stats = nn_ops.conv_fwd(x, wf, bias, want_stats=True)
    gamma, beta = torch.ones(co, device=DEV), torch.zeros(co, device=DEV)
    mean_invstd, _ = nn_ops.bn_finalize(stats, b * h * w, gamma, beta, None, None, 0.1, 1e-5)
    zz = z.double().reshape(-1, co)
    ref_var = zz.var(0, unbiased=False)
    got_var = 1.0 / mean_invstd[1].double() ** 2 - 1e-5
    assert float(((got_var - ref_var).abs() / ref_var).max()) < 1e-5
    assert float((mean_invstd[0].double() - zz.mean(0)).abs().max()) < 1e-4


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 20, 24, 64), (1, 33, 31, 128), (3, 8, 8, 512), (2, 16, 16, 32)])
def test_batchnorm_relu_fwd_bwd(shape, dt):
    from im2im_uq_amd import nn_ops
    b, h, w, c = shape
    z = rnd(b, c, h, w, seed=1) * 1.5 + 0.3
    gamma, beta = rnd(c, seed=2).abs() + 0.5, rnd(c, seed=3) * 0.2
    rm, rv = rnd(c, seed=4) * 0.1, rnd(c, seed=5).abs() + 0.5
    zq = q(z, dt).requires_grad_(True)
    g_r, b_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(zq, rm_r, rv_r, g_r, b_r, training=True, momentum=0.1, eps=1e-5))
    z_d = z.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    zf = z_d.float().reshape(-1, c)
    # three partial rows of unequal size in the epilogue's (mean, M2, count) form
    parts = torch.split(zf, [zf.shape[0] // 2, zf.shape[0] // 3, zf.shape[0] - zf.shape[0] // 2 - zf.shape[0] // 3])
    stats = torch.stack([torch.stack([pz.mean(0), ((pz - pz.mean(0)) ** 2).sum(0), torch.full((c,), float(pz.shape[0]), device=DEV)])
                         for pz in parts]).contiguous()
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    mean_invstd, scale_shift = nn_ops.bn_finalize(stats, b * h * w, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, 0.1, 1e-5)
    a = nn_ops.bn_relu_apply(z_d, scale_shift)
    assert rel_l2(a.float().cpu().permute(0, 3, 1, 2), ref.detach()) < (1e-5 if dt == F32 else 6e-3)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm_r.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv_r.numpy(), rtol=1e-4, atol=1e-5)
    ga = rnd(b, c, h, w, seed=6)
    ref.backward(q(ga, dt))
    ga_d = ga.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    dz, dgamma, dbeta = nn_ops.bn_relu_bwd(ga_d, z_d, scale_shift, mean_invstd)
    t = 2e-5 if dt == F32 else 1.2e-2
    assert rel_l2(dz.float().cpu().permute(0, 3, 1, 2), zq.grad) < t
    assert rel_l2(dgamma.cpu(), g_r.grad) < t and rel_l2(dbeta.cpu(), b_r.grad) < t
    # eval fold
    fold = nn_ops.bn_fold_eval(gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), None, 1e-5).cpu()
    ref_sc = gamma / torch.sqrt(rv + 1e-5)
    np.testing.assert_allclose(fold[0].numpy(), ref_sc.numpy(), rtol=1e-5)
    np.testing.assert_allclose(fold[1].numpy(), (beta - rm * ref_sc).numpy(), rtol=1e-4, atol=1e-6)
    cs = nn_ops.colsum(z_d)
    np.testing.assert_allclose(cs.detach().cpu().numpy(), zf.detach().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 20, 24, 64), (1, 33, 31, 128), (3, 9, 8, 512)])
@pytest.mark.parametrize("with_da", [True, False])
def test_fused_pool_batchnorm_backward_vs_torch(shape, dt, with_da):
    """im2im_bn_relu_pool_bwd == autograd of [a = relu(bn(z)); loss = <a, da> + <maxpool2(a), dpool>] (odd extents too)."""
    from im2im_uq_amd import nn_ops
    b, h, w, c = shape
    z = q(rnd(b, h, w, c, seed=1), dt)
    gamma, beta = 1.0 + 0.2 * rnd(c, seed=2), 0.1 * rnd(c, seed=3)
    da = q(rnd(b, h, w, c, seed=4), dt)
    dpool = q(rnd(b, h // 2, w // 2, c, seed=5), dt)
    mean = z.mean(dim=(0, 1, 2))
    var = z.var(dim=(0, 1, 2), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    ss = torch.stack([gamma * invstd, beta - mean * gamma * invstd])
    mi = torch.stack([mean, invstd])
    dz, dgamma, dbeta = nn_ops.bn_relu_pool_bwd(da.to(dt).to(DEV) if with_da else None, dpool.to(dt).to(DEV), z.to(dt).to(DEV),
                                                ss.to(DEV), mi.to(DEV))
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.relu(F.batch_norm(zr.permute(0, 3, 1, 2), None, None, gr, br, training=True, eps=1e-5))
    obj = (F.max_pool2d(a, 2) * dpool.permute(0, 3, 1, 2)).sum()
    if with_da:
        obj = obj + (a * da.permute(0, 3, 1, 2)).sum()
    obj.backward()
    t = 5e-5 if dt == F32 else 2e-2
    assert rel_l2(dz.float().cpu(), zr.grad) < t
    assert rel_l2(dgamma.cpu(), gr.grad) < t and rel_l2(dbeta.cpu(), br.grad) < t


@pytest.mark.parametrize("dt", [F32, BF16])
def test_maxpool_and_upsample_concat(dt):
    from im2im_uq_amd import nn_ops
    for (b, c, h, w) in [(2, 64, 16, 20), (1, 32, 7, 9), (1, 128, 40, 40)]:
        x = q(rnd(b, c, h, w, seed=1), dt)
        x[:, :, :2, :4] = 0.0                                     # ties (post-ReLU zeros): first max wins
        xr = x.clone().requires_grad_(True)
        ref = F.max_pool2d(xr, 2)
        xd = x.to(DEV).to(dt).to(memory_format=torch.channels_last).requires_grad_(True)
        y = nn_ops.MaxPool2.apply(xd)
        assert torch.equal(y.float().cpu(), ref.detach())
        g = q(rnd(*ref.shape, seed=2), dt)
        ref.backward(g)
        y.backward(g.to(DEV).to(dt))
        assert torch.equal(xd.grad.float().cpu(), xr.grad)
    for (b, cd, h, w, cs, hh, ww) in [(2, 32, 8, 8, 32, 16, 16), (1, 64, 5, 6, 32, 11, 13), (1, 512, 20, 20, 512, 40, 40)]:
        deep, skip = q(rnd(b, cd, h, w, seed=3), dt), q(rnd(b, cs, hh, ww, seed=4), dt)
        dr, sr = deep.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        u = F.interpolate(dr, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = hh - u.shape[2], ww - u.shape[3]
        ref = torch.cat([sr, F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])], dim=1)
        dd = deep.to(DEV).to(dt).to(memory_format=torch.channels_last).requires_grad_(True)
        sd = skip.to(DEV).to(dt).to(memory_format=torch.channels_last).requires_grad_(True)
        out = nn_ops.UpsampleConcat.apply(dd, sd)
        assert rel_l2(out.float().cpu(), ref.detach()) < (1e-6 if dt == F32 else 4e-3)
        g = q(rnd(*ref.shape, seed=5), dt)
        ref.backward(g)
        out.backward(g.to(DEV).to(dt))
        assert rel_l2(dd.grad.float().cpu(), dr.grad) < (1e-6 if dt == F32 else 6e-3)
        assert torch.equal(sd.grad.float().cpu(), sr.grad)


def test_quantile_loss_golden_and_oracle():
    from conftest import load_golden
    from im2im_uq_amd.core.models.finallayers.quantile_layer import quantile_regression_loss_fn
    from im2im_uq_amd.core.models.losses.pinball import PinballLoss
    g = load_golden("g2_quantile_loss")
    for params, lk, gk in ((dict(q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1), "loss", "grad"),
                           (dict(q_lo=0.1, q_hi=0.8, q_lo_weight=0.5, q_hi_weight=2.0, mse_weight=3.0), "loss_w", "grad_w")):
        p = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
        loss = quantile_regression_loss_fn(p, torch.from_numpy(g["target"]).to(DEV), params)
        (loss * 1.0).backward()
        assert loss.item() == pytest.approx(float(g[lk]), rel=2e-6)
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[gk], rtol=1e-5, atol=1e-9)
    g = load_golden("g1_pinball")
    for qq, tag in ((0.05, "005"), (0.95, "095")):
        o = torch.from_numpy(g["output"]).to(DEV).requires_grad_(True)
        loss = PinballLoss(quantile=qq)(o, torch.from_numpy(g["target"]).to(DEV))
        loss.backward()
        assert loss.item() == pytest.approx(float(g[f"loss_{tag}"]), rel=2e-6)
        np.testing.assert_allclose(o.grad.cpu().numpy(), g[f"grad_{tag}"], rtol=1e-5, atol=0)
    # full-size property: loss of (pred == target everywhere) is exactly 0 and its gradient is 0
    y = torch.rand(2, 1, 320, 320, device=DEV)
    p = torch.stack([y, y, y], dim=1).contiguous().requires_grad_(True)
    loss = quantile_regression_loss_fn(p, y, dict(q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1))
    loss.backward()
    assert loss.item() == 0.0 and float(p.grad.abs().max()) == 0.0


def test_fused_adam_matches_torch_adam():
    from im2im_uq_amd import nn_ops
    shapes = [(64, 1, 3, 3), (64,), (512, 1024, 3, 3), (3,), (32, 64, 1, 1)] + [(7,)] * 30    # > 24 tensors: two launches
    ps_ref = [rnd(*s, seed=i).requires_grad_(True) for i, s in enumerate(shapes)]
    ps = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    o_ref = torch.optim.Adam(ps_ref, lr=1e-3)
    o = nn_ops.FusedAdam(ps, lr=1e-3)
    for step in range(4):
        for i, (pr, p) in enumerate(zip(ps_ref, ps)):
            g = rnd(*pr.shape, seed=100 * step + i) * (10.0 ** (i % 5 - 3))
            pr.grad = g.clone()
            p.grad = g.to(DEV)
        o_ref.step(); o.step()
    for pr, p in zip(ps_ref, ps):
        np.testing.assert_allclose(p.detach().cpu().numpy(), pr.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_fused_adam_bias_correction_is_per_parameter():
    """a parameter that starts receiving gradients later (unfrozen, or unused at first) has its own step count in
    torch.optim.Adam; FusedAdam launches once per distinct step count so its bias correction matches."""
    from im2im_uq_amd import nn_ops
    ps_ref = [rnd(40, 3, seed=1).requires_grad_(True), rnd(17, seed=2).requires_grad_(True)]
    ps = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    o_ref, o = torch.optim.Adam(ps_ref, lr=1e-2), nn_ops.FusedAdam(ps, lr=1e-2)
    for step in range(5):
        for i, (pr, p) in enumerate(zip(ps_ref, ps)):
            if i == 1 and step < 3:
                pr.grad, p.grad = None, None             # no gradient during the first three steps
                continue
            g = rnd(*pr.shape, seed=10 * step + i)
            pr.grad, p.grad = g.clone(), g.to(DEV)
        o_ref.step(); o.step()
    assert int(o.state[ps[0]]["step"]) == 5 and int(o.state[ps[1]]["step"]) == 2
    for pr, p in zip(ps_ref, ps):
        np.testing.assert_allclose(p.detach().cpu().numpy(), pr.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_inn_loss_class_matches_formula():
    """losses/inn.py mirror: INNLoss(beta)(lower, upper, target), mean and sum reductions, value and gradients."""
    from im2im_uq_amd.core.models.losses.inn import INNLoss
    lo, up, y = rnd(3, 20, 24, seed=1), rnd(3, 20, 24, seed=2) + 0.5, rnd(3, 20, 24, seed=3)
    for red in ("mean", "sum"):
        l_d, u_d = lo.to(DEV).requires_grad_(True), up.to(DEV).requires_grad_(True)
        got = INNLoss(beta=0.2, reduction=red)(l_d, u_d, y.to(DEV))
        got.backward()
        l_r, u_r = lo.clone().requires_grad_(True), up.clone().requires_grad_(True)
        ref = torch.relu(y - u_r).square() + torch.relu(l_r - y).square() + 0.2 * torch.abs(u_r - l_r)
        ref = ref.sum() if red == "sum" else ref.mean()
        ref.backward()
        assert got.item() == pytest.approx(ref.item(), rel=2e-5)
        assert rel_l2(l_d.grad.cpu(), l_r.grad) < 2e-5 and rel_l2(u_d.grad.cpu(), u_r.grad) < 2e-5


# BASELINE configs[1] layer shapes: B >= 8 so the 1x16x16 tile path sees several images, the split
# cases are the Up blocks' first conv reading [skip, upsampled] from two tensors.  (B, H, W, Ci, Co, split)
BASELINE_LAYER_CASES = [
    (8, 320, 320, 64, 64, False),      # inc / up4 second conv: the 1x16x16, BN=64 variant (40 % of the FLOPs)
    (8, 320, 320, 128, 64, True),      # up4 first conv: 64 skip + 64 upsampled channels
    (8, 160, 160, 64, 128, False),     # down1 first conv: 1x16x16, BN=128
    (8, 40, 40, 1024, 512, True),      # up1 first conv: 4x8x8 tiles, 512 + 512 channels, K = 9216
]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", BASELINE_LAYER_CASES)
def test_conv_fwd_dgrad_wgrad_at_baseline_layer_shapes(case, dt):
    """forward (+ BatchNorm partial statistics), data-gradient and weight-gradient at the benchmarked layer shapes
    against F.conv2d / its autograd in fp32 on the CPU, operands quantised the way the kernel sees them.
    Tolerance (relative L2): fp32 2e-5, bf16 1.5e-2 (bf16 rounding of the stored result; K up to 9216)."""
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co, split = case
    x = rnd(b, ci, h, w, seed=11)
    wt = rnd(co, ci, 3, 3, seed=12, scale=(ci * 9) ** -0.5)
    bias = rnd(co, seed=13, scale=0.1)
    gy = rnd(b, co, h, w, seed=16)
    xq, wq = q(x, dt).requires_grad_(True), q(wt, dt).requires_grad_(True)
    torch.set_num_threads(min(32, torch.get_num_threads() or 1))
    ref = F.conv2d(xq, wq, bias, padding=1)
    ref.backward(q(gy, dt))
    x_d = x.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    gy_d = gy.to(DEV).permute(0, 2, 3, 1).contiguous().to(dt)
    wf, wd = nn_ops.pack_weight(wt.to(DEV), dt)
    if split:
        lo, hi = x_d[..., : ci // 2].contiguous(), x_d[..., ci // 2:].contiguous()
        y, stats = nn_ops.conv_fwd(lo, wf, bias.to(DEV), want_stats=True, x_hi=hi)
        dw = nn_ops.conv_wgrad(lo, gy_d, 9, x_hi=hi)
        dx_lo, dx_hi = nn_ops.conv_fwd(gy_d, wd, split_out=ci // 2)
        dx = torch.cat([dx_lo, dx_hi], dim=3)
    else:
        y, stats = nn_ops.conv_fwd(x_d, wf, bias.to(DEV), want_stats=True)
        dw = nn_ops.conv_wgrad(x_d, gy_d, 9)
        dx = nn_ops.conv_fwd(gy_d, wd)
    torch.cuda.synchronize()
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref.detach()) < tol(dt)
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2), xq.grad) < tol(dt)
    assert rel_l2(dw.cpu().view(co, ci, 3, 3), wq.grad) < tol(dt)
    # worst single element, against the tensor's RMS (a dropped tile or a wrong halo shows here, not in the L2)
    for got, want in ((y.float().cpu().permute(0, 3, 1, 2), ref.detach()), (dx.float().cpu().permute(0, 3, 1, 2), xq.grad),
                      (dw.cpu().view(co, ci, 3, 3), wq.grad)):
        rms = float(want.double().pow(2).mean().sqrt())
        assert float((got - want).abs().max()) < (2e-4 if dt == F32 else 6e-2) * rms
    n, mean, m2 = merged_moments(stats)
    yst = y.double().cpu().reshape(-1, co)
    assert float((n - yst.shape[0]).abs().max()) == 0.0
    np.testing.assert_allclose(mean.numpy(), yst.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m2.numpy(), ((yst - yst.mean(0)) ** 2).sum(0).numpy(), rtol=2e-5)
