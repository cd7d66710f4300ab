"""Round-4 kernels against CPU restatements and against the one-kernel paths they replace:
  * split-K convolution for under-filled launches (im2im_conv_fwd_split_ws, csrc/conv_mfma.hip EPI 4 + conv_splitk_reduce_kernel)
    vs F.conv2d on the CPU and vs the unsplit kernel (reference op: nn.Conv2d of unet_parts.py:16,19 at the per-GPU batch of a
    data-parallel job, train.py:112-115);
  * the block-per-32x32 weight packer vs the element-per-thread one (bit-equal);
  * the two-stage BatchNorm reductions: many launches of changing shapes in a row against float64 sums."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32, BF16 = torch.float32, torch.bfloat16


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def merged_moments(stats):
    st = stats.double().cpu()
    n = st[:, 2].sum(0)
    mean = (st[:, 2] * st[:, 0]).sum(0) / n
    m2 = st[:, 1].sum(0) + (st[:, 2] * (st[:, 0] - mean) ** 2).sum(0)
    return n, mean, m2


SPLITK_CASES = [
    # B, H, W, Ci, Co, split_in, split_out      (all: 8x8 tiles of two images, < 384 workgroups, K = 9*Ci >= 4608)
    (10, 20, 20, 512, 512, False, 0),       # the 20x20 level at the per-GPU batch of 78/8: 180 workgroups -> 4 splits
    (10, 40, 40, 512, 256, False, 0),       # 40x40, 250 workgroups -> 3 splits of 6/6/4 chunks
    (3, 24, 20, 1024, 128, True, 0),        # split INPUT (the Up block's [skip, upsampled]), overhanging tiles, odd batch
    (4, 16, 24, 512, 256, False, 128),      # split OUTPUT (a data-gradient landing in d(skip), d(up))
]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", SPLITK_CASES)
def test_splitk_conv_vs_cpu_and_vs_unsplit(case, dt):
    from im2im_uq_amd import hip_ops, nn_ops
    from im2im_uq_amd._lib import lib
    b, h, w_, ci, co, split_in, split_out = case
    assert lib.im2im_conv_splitk_workspace_bytes(b, h, w_, ci, co, 9) > 0, "case must be one the library splits"
    g = torch.Generator().manual_seed(11)
    cin = ci // 2 if split_in else ci
    x = torch.randn(b, h, w_, cin, generator=g)
    xh = torch.randn(b, h, w_, cin, generator=g) if split_in else None
    wt = torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (3 * ci ** 0.5))
    bias = torch.randn(co, generator=g)
    ss = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3])
    xd, xhd = x.to(DEV, dt), (xh.to(DEV, dt) if split_in else None)
    wf, _ = nn_ops.pack_weight(wt.to(DEV), dt)
    want_stats = not split_out
    kw = dict(bias=None if split_out else bias.to(DEV), want_stats=want_stats, in_ss=None if split_out else ss.to(DEV),
              x_hi=xhd, in_ss_hi=None, split_out=split_out)
    try:
        hip_ops.set_option("conv_splitk", 0)
        ref = nn_ops.conv_fwd(xd, wf, **kw)
        hip_ops.set_option("conv_splitk", 3)
        got = nn_ops.conv_fwd(xd, wf, **kw)
        got2 = nn_ops.conv_fwd(xd, wf, **kw)
    finally:
        hip_ops.set_option("conv_splitk", 3)
    # CPU restatement on the operands as the kernel sees them (rounded to the compute dtype, lazy BatchNorm+ReLU on the low half)
    xq = x.to(dt).float()
    if not split_out:
        xq = torch.clamp_min(xq * ss[0] + ss[1], 0).to(dt).float()
    xin = torch.cat([xq, xh.to(dt).float()], -1) if split_in else xq
    want = F.conv2d(xin.permute(0, 3, 1, 2), wt.to(dt).float(), None if split_out else bias, padding=1).permute(0, 2, 3, 1)
    tol = 2e-5 if dt == F32 else 6e-3
    if split_out:
        y = torch.cat([got[0], got[1]], -1).float().cpu()
        yr = torch.cat([ref[0], ref[1]], -1).float().cpu()
        assert torch.equal(got[0], got2[0]) and torch.equal(got[1], got2[1])           # deterministic
    else:
        y, yr = got[0].float().cpu(), ref[0].float().cpu()
        assert torch.equal(got[0], got2[0]) and torch.equal(got[1], got2[1])
    assert rel_l2(y, want) < tol, rel_l2(y, want)
    assert rel_l2(y, yr) < (3e-6 if dt == F32 else 3e-3)        # same sums, another order: fp32 noise / one bf16 rounding apart
    if want_stats:
        n, mean, m2 = merged_moments(got[1])
        yv = got[0].double().cpu().reshape(-1, co)              # statistics describe the STORED values
        assert torch.equal(n, torch.full_like(n, float(b * h * w_)))
        assert torch.allclose(mean, yv.mean(0), rtol=0, atol=2e-5 * float(yv.abs().max()))
        assert torch.allclose(m2, ((yv - yv.mean(0)) ** 2).sum(0), rtol=2e-4)
        assert got[1].shape == ref[1].shape                     # one row per conv tile, as the one-kernel epilogue


def test_splitk_model_step_matches_unsplit_step():
    """a whole train step of the depth-4 UNet at the per-GPU batch 10 with and without split-K: the 20x20 / 40x40 levels take the
    split path (forward and data-gradient); losses equal to fp32-rounding, gradients to the bf16 / fp32 noise of a re-ordered sum."""
    from im2im_uq_amd import hip_ops, nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from oracle import model as om
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    nn_ops.set_compute_dtype("fp32")
    try:
        x, y = om.det_images(10, 1, 160, 160, salt=5)           # 160 -> 10x10 at the bottom: every level below 64 px is "small"
        out = {}
        for mode in (0, 3):
            hip_ops.set_option("conv_splitk", mode)
            model = add_uncertainty(UNet(1, 1), dict(params))
            model.load_state_dict(om.det_state(1, 1))
            model = model.to(DEV).train()
            loss = model.loss_fn(model(x.to(DEV)), y.to(DEV))
            loss.backward()
            torch.cuda.synchronize()
            out[mode] = (loss.item(), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None})
    finally:
        hip_ops.set_option("conv_splitk", 3)
        nn_ops.set_compute_dtype("bf16")
    assert abs(out[0][0] - out[3][0]) < 1e-5 * abs(out[0][0])
    worst = max(rel_l2(out[3][1][k], out[0][1][k]) for k in out[0][1] if float(out[0][1][k].abs().max()) > 0)
    # fp32 re-association through 18 layers of an ill-conditioned net: a 1e-7 relative change of the input moves the reference's OWN
    # fp32 gradients by 2e-3 (tools/debug_grad.py); measured here 4.2e-3
    assert worst < 1.5e-2, worst


def test_pack_frag_multi_bit_equal_to_single_tensor_pack():
    from im2im_uq_amd import nn_ops
    torch.manual_seed(2)
    shapes = [(64, 64), (128, 64), (64, 128), (512, 1024), (32, 96), (256, 32)]
    ws = [torch.nn.Parameter(torch.randn(co, ci, 3, 3, device=DEV)) for co, ci in shapes]
    ws.append(torch.nn.Parameter(torch.randn(32, 64, 1, 1, device=DEV)))          # 1x1: the chunked kernel in the same call
    for w in ws:
        nn_ops.packed_pair(w, BF16)                           # registers it; the LAST call packs all of them in one batch
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5)                                       # bumps the version counters: everything is stale
    got = [nn_ops.packed_pair(w, BF16) for w in ws]
    for w, (wf, wd) in zip(ws, got):
        rf, rd = nn_ops.pack_weight(w, BF16)
        assert torch.equal(wf, rf) and torch.equal(wd, rd), tuple(w.shape)


def test_bn_reductions_many_launches_of_changing_shapes():
    from im2im_uq_amd import nn_ops
    g = torch.Generator().manual_seed(4)
    for it in range(40):
        c = [32, 64, 128, 192, 512, 1024][it % 6]
        rows = int(torch.randint(1, 3000, (1,), generator=g))
        stats = torch.empty(rows, 3, c)
        stats[:, 0] = torch.randn(rows, c, generator=g) * 2 + 5
        stats[:, 1] = torch.rand(rows, c, generator=g) * 30
        stats[:, 2] = torch.randint(1, 257, (rows, 1), generator=g).float()
        gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        mi, ss = nn_ops.bn_finalize(stats.to(DEV), int(stats[:, 2, 0].sum()), gamma.to(DEV), beta.to(DEV), rm, rv, 0.1, 1e-5,
                                    num_batches_tracked=nbt)
        n, mean, m2 = merged_moments(stats)
        var = m2 / n
        assert int(nbt) == 1
        assert torch.allclose(mi[0].double().cpu(), mean, rtol=1e-6, atol=1e-6), it
        assert torch.allclose(mi[1].double().cpu(), 1 / torch.sqrt(var + 1e-5), rtol=2e-6), it
        assert torch.allclose(ss[0].double().cpu(), gamma.double() / torch.sqrt(var + 1e-5), rtol=2e-6), it
        assert torch.allclose(rm.double().cpu(), 0.1 * mean, rtol=2e-6, atol=1e-7), it
        # backward sums on a small activation of this width
        m = int(torch.randint(64, 6000, (1,), generator=g))
        z = torch.randn(m, c, generator=g)
        da = torch.randn(m, c, generator=g)
        mu, istd = z.mean(0), 1 / torch.sqrt(z.var(0, unbiased=False) + 1e-5)
        sc, sh = gamma * istd, beta - mu * gamma * istd
        dz, dg, db = nn_ops.bn_relu_bwd(da.to(DEV).view(1, 1, m, c), z.to(DEV).view(1, 1, m, c), torch.stack([sc, sh]).to(DEV),
                                        torch.stack([mu, istd]).to(DEV))
        gg = (da * ((z * sc + sh) > 0)).double()
        xhat = ((z - mu) * istd).double()
        assert torch.allclose(db.double().cpu(), gg.sum(0), rtol=1e-4, atol=1e-3), it
        assert torch.allclose(dg.double().cpu(), (gg * xhat).sum(0), rtol=1e-4, atol=1e-3), it
        want = sc.double() * (gg - gg.mean(0) - xhat * (gg * xhat).mean(0))
        assert rel_l2(dz.double().cpu().view(m, c), want) < 2e-5, it


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_eval_conv_with_fused_maxpool_equals_conv_then_maxpool(dt):
    """[r4] im2im_conv_fwd_eval_pool: the eval-mode conv + folded BatchNorm + ReLU of a block that also feeds MaxPool2d(2)
    (unet_parts.py:33-36) hands back the pooled tensor from its epilogue -- same bits as the separate max-pool pass, for every tile
    family (32x16x64, 16x16x128, 8x8 patches of several images), overhanging tiles included; and a whole eval forward with and
    without the fusion is bit-identical."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype(dt)
    cdt = BF16 if dt == "bf16" else F32
    try:
        g = torch.Generator().manual_seed(8)
        for (b, h, w_, ci, co) in [(2, 64, 96, 64, 64), (3, 80, 72, 64, 128), (5, 40, 24, 128, 256), (1, 20, 36, 256, 256), (2, 70, 66, 64, 32)]:
            x = torch.randn(b, ci, h, w_, generator=g).to(DEV).to(cdt).contiguous(memory_format=torch.channels_last)
            conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
            bn = torch.nn.BatchNorm2d(co).to(DEV)
            with torch.no_grad():
                bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
                args = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, cdt)
                a, pooled = nn_ops.conv_bn_relu_eval(x, *args, pool=True)
                a0 = nn_ops.conv_bn_relu_eval(x, *args)
                assert torch.equal(a, a0)
                assert torch.equal(pooled, nn_ops.MaxPool2.apply(a0)), (b, h, w_, ci, co)
        torch.manual_seed(1)
        params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
        model = add_uncertainty(UNet(1, 1), dict(params)).to(DEV)
        with torch.no_grad():
            model.train()
            model(torch.randn(2, 1, 64, 64, device=DEV))             # non-trivial BatchNorm buffers
            model.eval()
            xin = torch.randn(3, 1, 96, 80, device=DEV)
            fused = model(xin)
            real = nn_ops.conv_bn_relu_eval
            nn_ops.conv_bn_relu_eval = lambda *a_, pool=False, **k: real(*a_, **k)
            try:
                plain = model(xin)
            finally:
                nn_ops.conv_bn_relu_eval = real
        assert torch.equal(fused, plain)
    finally:
        nn_ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_eval_last_block_with_fused_outconv_equals_the_two_kernels(dt):
    """[r4] im2im_conv_fwd_eval_tail: in inference the trunk's last conv (-> 64 channels) evaluates OutConv's 1x1 (unet_parts.py:87-94)
    on its epilogue tile; the model output is bit-identical to the path with the separate 1x1 kernel, for the 32x16, 16x16 and
    8x8-patch tile families, depth 2 and 4."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype(dt)
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    was = nn_ops.FUSE_EVAL_OUTCONV
    try:
        for depth, shape in ((4, (3, 1, 96, 80)), (2, (2, 1, 64, 128)), (2, (5, 1, 40, 24)), (4, (1, 2, 160, 176))):
            torch.manual_seed(depth)
            model = add_uncertainty(UNet(shape[1], 1, depth=depth), dict(params)).to(DEV)
            with torch.no_grad():
                model.train()
                model(torch.randn(2, shape[1], 64, 64, device=DEV))
                model.eval()
                x = torch.randn(*shape, device=DEV)
                calls = []
                real = nn_ops.Conv1x1.apply
                nn_ops.FUSE_EVAL_OUTCONV = True
                fused = model(x)
                feat = model.baseModel(x)
                assert getattr(feat, "_im2im_tail_done", False), shape         # the fused kernel took it
                nn_ops.FUSE_EVAL_OUTCONV = False
                plain = model(x)
                assert not getattr(model.baseModel(x), "_im2im_tail_done", False)
            assert torch.equal(fused, plain), shape
            # grad-enabled eval and train mode never take the fused path
            model.train()
            nn_ops.FUSE_EVAL_OUTCONV = True
            assert not getattr(model.baseModel(x), "_im2im_tail_done", False)
    finally:
        nn_ops.FUSE_EVAL_OUTCONV = was
        nn_ops.set_compute_dtype("bf16")


@pytest.mark.parametrize("rows,c", [(1, 64), (45, 512), (250, 256), (256, 72), (257, 64)])
def test_bn_sums_in_one_launch_for_few_partial_rows(rows, c):
    """[r4] bn_stats_fused_kernel / bn_bwd_sums_fused_kernel (<= 256 partial rows; 257 stays two-stage): the statistics, the
    running buffers and the backward coefficients equal the two-stage path's to fp64-summation-order noise and a float64
    evaluation of the same merge."""
    from im2im_uq_amd import hip_ops, nn_ops
    g = torch.Generator().manual_seed(rows * 1000 + c)
    n = torch.randint(1, 300, (rows, 1, c), generator=g).float()
    mean = torch.randn(rows, 1, c, generator=g) * 0.3 + 1.5
    m2 = torch.rand(rows, 1, c, generator=g) * n * 0.8
    stats = torch.cat([mean, m2, n], dim=1).contiguous().to(DEV)                 # [R][3][C]: (mean, M2, n) per tile
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    count = int(n[:, 0, 0].sum())
    out = {}
    try:
        for mode in (0, 1):
            hip_ops.set_option("bn_fused_small", mode)
            rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
            nbt = torch.zeros((), dtype=torch.int64, device=DEV)
            mi, ss = nn_ops.bn_finalize(stats, count, gamma, beta, rm, rv, 0.1, 1e-5, num_batches_tracked=nbt)
            out[mode] = (mi.cpu(), ss.cpu(), rm.cpu(), rv.cpu(), int(nbt))
    finally:
        hip_ops.set_option("bn_fused_small", 1)
    nd, md, qd = n.double()[:, 0], mean.double()[:, 0], m2.double()[:, 0]
    N = nd.sum(0)
    mu = (nd * md).sum(0) / N
    var = (qd + nd * (md - mu) ** 2).sum(0) / N
    assert out[0][4] == out[1][4] == 1
    for mode in (0, 1):
        mi, ss, rm, rv, _ = out[mode]
        torch.testing.assert_close(mi[0].double(), mu, rtol=2e-7, atol=1e-7)
        torch.testing.assert_close(mi[1].double(), 1.0 / torch.sqrt(var + 1e-5), rtol=3e-7, atol=0)
        torch.testing.assert_close(rm.double(), 0.1 * mu, rtol=3e-7, atol=1e-8)
        torch.testing.assert_close(rv.double(), 0.9 + 0.1 * var * N / (N - 1), rtol=3e-7, atol=0)
    for a, b in zip(out[0][:4], out[1][:4]):
        torch.testing.assert_close(a, b, rtol=2.4e-7, atol=1e-7)                 # one fp32 rounding of differently-ordered fp64 sums
    # backward: m pixels chosen so that the reduce kernel writes `rows` partial rows (64 pixels per row)
    m = rows * 64
    da = torch.randn(m, c, generator=g).to(DEV).bfloat16()
    z = torch.randn(m, c, generator=g).to(DEV).bfloat16()
    mi, ss = out[1][0].to(DEV), out[1][1].to(DEV)
    res = {}
    try:
        for mode in (0, 1):
            hip_ops.set_option("bn_fused_small", mode)
            res[mode] = [t.float().cpu() for t in nn_ops.bn_relu_bwd(da, z, ss, mi)]
    finally:
        hip_ops.set_option("bn_fused_small", 1)
    for a, b in zip(res[0], res[1]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    mask = (z.float() * ss[0] + ss[1] > 0).double()
    gd = da.double() * mask
    torch.testing.assert_close(res[1][2].double(), gd.sum(0).cpu(), rtol=1e-5, atol=1e-4)          # dbeta


# ------------------------------------------------------------------------------------------------
# conv_wgrad_roll_kernel / conv_wgrad_fp8_roll_kernel (csrc/conv_wgrad.hip): the weight gradient of nn.Conv2d under loss.backward()
# (reference: core/models/trunks/unet_parts.py:16,19 at core/scripts/train.py:159).  Same LDS images, fragments and accumulation order as
# the pipe kernels they replace => the SAME BITS, for every form: 128- / 64-output-channel tiles, lazy / plain / split input, tiles that
# hang over the image (40- and 20-pixel rows, odd extents), splits of one or two tiles (the pipeline's prologue and its repeated last tile).
WG_CASES = [
    # (B, H, W, Ci, Co, split_in, lazy)
    (3, 32, 32, 64, 128, False, True),      # 128-wide form, full tiles
    (2, 40, 40, 128, 128, False, True),     # ... tiles hang over the 40-pixel rows (PARTIAL)
    (2, 20, 20, 64, 256, False, False),     # ... 20x20, plain input
    (1, 24, 48, 128, 128, True, True),      # ... split input: the low half lazy, the high half plain
    (2, 64, 64, 64, 64, False, True),       # 64-output-channel form on 256-pixel swizzled tiles
    (1, 64, 80, 128, 64, True, True),       # ... split input
    (1, 72, 88, 64, 64, False, False),      # ... overhanging 16x16 tiles, plain input
    (1, 8, 16, 64, 128, False, True),       # ONE tile in the whole problem
    (1, 16, 16, 64, 128, False, False),     # two tiles
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_roll_weight_gradient_bit_identical_to_the_pipe_kernel(case):
    from im2im_uq_amd import hip_ops, nn_ops
    b, h, w, ci, co, split, lazy = case
    g = torch.Generator(device=DEV).manual_seed(7)
    cin = ci // 2 if split else ci
    x = torch.randn(b, h, w, cin, device=DEV, generator=g).to(torch.bfloat16)
    xh = torch.randn(b, h, w, cin, device=DEV, generator=g).to(torch.bfloat16) if split else None
    dz = torch.randn(b, h, w, co, device=DEV, generator=g).to(torch.bfloat16)
    ss = torch.stack([torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g)]).contiguous() if lazy else None
    outs = {}
    try:
        for mode in (0, 1):
            hip_ops.set_option("wgrad_roll", mode)
            outs[mode] = nn_ops.conv_wgrad(x, dz, 9, x_ss=ss, x_hi=xh).clone()
    finally:
        hip_ops.set_option("wgrad_roll", 1)
    assert torch.equal(outs[0], outs[1])
    # and against torch on the CPU (fp32 math on the bf16 operands; the kernels accumulate in fp32 in another order)
    a = x.float()
    if lazy:
        a = torch.clamp_min(a * ss[0] + ss[1], 0).to(torch.bfloat16).float()
    if split:
        a = torch.cat([a, xh.float()], 3)
    ref = torch.nn.grad.conv2d_weight(a.permute(0, 3, 1, 2).cpu(), (co, ci, 3, 3), dz.float().permute(0, 3, 1, 2).cpu(), padding=1)
    got = outs[1].reshape(co, ci, 3, 3).cpu()
    assert float((got - ref).norm() / ref.norm()) < 2e-5


@pytest.mark.parametrize("case", [c for c in WG_CASES if c[3] % 64 == 0 and c[4] % 64 == 0][:7], ids=lambda c: "x".join(str(v) for v in c))
def test_fp8_roll_weight_gradient_bit_identical_to_the_fp8_pipe_kernel(case):
    from im2im_uq_amd import hip_ops, nn_ops
    b, h, w, ci, co, split, lazy = case
    g = torch.Generator(device=DEV).manual_seed(11)
    cin = ci // 2 if split else ci
    x = torch.randn(b, h, w, cin, device=DEV, generator=g).to(torch.bfloat16)
    xh = torch.randn(b, h, w, cin, device=DEV, generator=g).to(torch.bfloat16) if split else None
    dz = (torch.randn(b, h, w, co, device=DEV, generator=g) * 1e-3).to(torch.bfloat16)
    amax = dz.float().abs().max().reshape(1).contiguous()
    ss = torch.stack([torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g)]).contiguous() if lazy else None
    outs = {}
    try:
        for mode in (0, 1):
            hip_ops.set_option("wgrad_roll", mode)
            outs[mode] = nn_ops.conv_wgrad_fp8(x, dz, amax.data_ptr(), x_ss=ss, x_hi=xh).clone()
    finally:
        hip_ops.set_option("wgrad_roll", 1)
    assert torch.equal(outs[0], outs[1])
