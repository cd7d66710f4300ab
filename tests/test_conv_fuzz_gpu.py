"""Seeded random geometries through the 3x3 conv kernels (bf16, and fp32 with tight tolerances) (forward with statistics, lazy BatchNorm+ReLU input, split input,
data-gradient incl. the split result, weight gradient) against torch on the CPU with identically rounded operands: the tile
choice depends on (B, H, W, Co) -- 16x16 / 32x16 / four- or two-image 8x8 tiles, 128- / 64- / 32-wide, weights straight from L2
or staged through LDS -- so random shapes walk every combination incl. overhanging tiles and odd batches.
Replaces nn.Conv2d 3x3 pad 1 and its autograd backward (core/models/trunks/unet_parts.py:16,19)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16, F32 = torch.bfloat16, torch.float32


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cases(n=28, seed=20260929):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        ci = int(rs.choice([32, 64, 64, 96, 128, 128, 192, 256]))
        co = int(rs.choice([32, 64, 64, 128, 128, 192, 256, 512]))
        if i % 4 == 0:
            h, w = int(rs.choice([32, 64, 96])), int(rs.randint(64, 100))           # H % 32 == 0: the 32x16 tile when co == 64
        elif i % 4 == 1:
            h, w = int(rs.randint(5, 48)), int(rs.randint(5, 48))                   # small-extent levels: 8x8 tiles
        else:
            h, w = int(rs.randint(20, 90)), int(rs.randint(20, 90))
        b = int(rs.randint(1, 6)) if i % 5 else int(rs.randint(12, 20))             # some batches large enough for four-image tiles
        out.append((b, h, w, ci, co))
    return out


@pytest.mark.parametrize("case", [c + (BF16,) for c in _cases()] + [c + (F32,) for c in _cases(10, seed=7)],
                         ids=lambda c: "x".join(map(str, c[:5])) + ("_bf16" if c[5] == BF16 else "_fp32"))
def test_random_geometry_conv_matches_torch(case):
    from im2im_uq_amd import nn_ops
    b, h, w, ci, co, DT = case
    TOL = 1.2e-2 if DT == BF16 else 2e-5
    g = torch.Generator().manual_seed(hash(case[:5]) % 100000)
    x = torch.randn(b, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5
    gy = torch.randn(b, co, h, w, generator=g)
    ss = torch.stack([torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.5])
    q = lambda t: t.to(DT).to(F32)
    xq, wq, gyq = q(x), q(wt).requires_grad_(True), q(gy)
    a = q(torch.relu(xq * ss[0][None, :, None, None] + ss[1][None, :, None, None])).requires_grad_(True)   # the lazy input as the kernels form it
    ref = F.conv2d(a, wq, None, padding=1)
    ref.backward(gyq)
    x_d = x.to(DEV).permute(0, 2, 3, 1).contiguous().to(DT)
    gy_d = gy.to(DEV).permute(0, 2, 3, 1).contiguous().to(DT)
    ss_d = ss.to(DEV).contiguous()
    wf, wd = nn_ops.pack_weight(wt.to(DEV), DT)
    # forward, lazy BatchNorm+ReLU input, statistics of the stored values
    y, stats = nn_ops.conv_fwd(x_d, wf, None, want_stats=True, in_ss=ss_d)
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref.detach()) < TOL
    st = stats.double().cpu()
    n = st[:, 2].sum(0)
    mean = (st[:, 2] * st[:, 0]).sum(0) / n
    yst = y.double().cpu().reshape(-1, co)
    assert float((n - yst.shape[0]).abs().max()) == 0.0
    np.testing.assert_allclose(mean.numpy(), yst.mean(0).numpy(), rtol=1e-4, atol=1e-5)
    # the same with the input split over two tensors (the Up blocks' concatenation that is never materialised)
    if ci % 64 == 0:
        lo, hi = x_d[..., :ci // 2].contiguous(), x_d[..., ci // 2:].contiguous()
        ss_lo = ss_d[:, :ci // 2].contiguous()
        ss_hi = ss_d[:, ci // 2:].contiguous()
        y2 = nn_ops.conv_fwd(lo, wf, None, in_ss=ss_lo, x_hi=hi, in_ss_hi=ss_hi)
        assert torch.equal(y2, y)
    # data-gradient (plain, and split into two result tensors where the halves are 64-channel multiples)
    dx = nn_ops.conv_fwd(gy_d, wd)
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2), a.grad) < TOL
    if ci % 128 == 0:
        d_lo, d_hi = nn_ops.conv_fwd(gy_d, wd, split_out=ci // 2)
        assert torch.equal(torch.cat([d_lo, d_hi], dim=-1), dx)
    # weight gradient (lazy input re-formed in its staging)
    dw = nn_ops.conv_wgrad(x_d, gy_d, 9, x_ss=ss_d)
    assert rel_l2(dw.cpu(), wq.grad) < TOL
