"""Round 6 kernels (GPU): the one-launch BatchNorm reductions (last block finalizes) and the multi-tensor split-K reduction of
the weight gradients give the bits of the two-launch / per-layer forms they replace, leave their ticket counters zero, and are
reproducible launch after launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,c", [(300, 64), (4000, 64), (31200, 64), (7800, 128), (1950, 256), (520, 512), (1000, 1024), (257, 40)])
def test_bn_statistics_one_launch_equals_two_launches(rows, c):
    """im2im_bn_finalize with ticket counters (bn_stats_onelaunch_kernel) vs without (bn_stats_stage1_kernel + bn_finalize_kernel):
    mean / invstd / scale / shift / running statistics / num_batches_tracked bit for bit, counters back at zero, ten launches in a
    row identical."""
    from im2im_uq_amd import hip_ops, nn_ops
    g = torch.Generator().manual_seed(rows + c)
    n = torch.randint(200, 257, (rows, 1, c), generator=g).float()
    stats = torch.cat([torch.randn(rows, 1, c, generator=g) * 0.3 + 1.5, torch.rand(rows, 1, c, generator=g) * 40 + 5, n], dim=1).to(DEV)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    count = int(n.sum().item() // c)

    def run(one):
        hip_ops.set_option("bn_onelaunch", one)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        mi, ss = nn_ops.bn_finalize(stats, count, gamma, beta, rm, rv, 0.1, 1e-5, num_batches_tracked=nbt)
        torch.cuda.synchronize()
        return [t.clone() for t in (mi, ss, rm, rv, nbt)]
    try:
        ref = run(0)
        for rep in range(10):
            got = run(1)
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
            assert int(nn_ops._Scratch.counters(DEV).abs().sum()) == 0
    finally:
        hip_ops.set_option("bn_onelaunch", 0)            # the library's default (the one-launch form measured slower)
    assert int(ref[4]) == 1 and bool(torch.isfinite(ref[0]).all())


@pytest.mark.parametrize("shape", [(78, 80, 80, 64), (10, 160, 160, 128), (6, 40, 40, 512), (3, 36, 28, 256)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bn_backward_one_launch_equals_two_launches(shape, dtype):
    """im2im_bn_relu_bwd and im2im_bn_relu_pool_bwd with ticket counters vs without: dz, dgamma, dbeta bit for bit."""
    from im2im_uq_amd import hip_ops, nn_ops
    b, h, w, c = shape
    g = torch.Generator().manual_seed(c + h)
    z = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    da = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    dpool = torch.randn(b, h // 2, w // 2, c, generator=g).to(dtype).to(DEV)
    ss = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3]).to(DEV)
    mi = torch.stack([torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5]).to(DEV)

    def run(one):
        hip_ops.set_option("bn_onelaunch", one)
        hip_ops.set_option("bn_fused_small", 0)          # the small-grid one-launch form [r4] off: every row count takes the path under test
        out = list(nn_ops.bn_relu_bwd(da, z, ss, mi))
        if nn_ops.can_fuse_pool_bwd(z.permute(0, 3, 1, 2)) and h % 2 == 0 and w % 2 == 0:
            out += list(nn_ops.bn_relu_pool_bwd(da, dpool, z, ss, mi))
        torch.cuda.synchronize()
        return [t.clone() for t in out]
    try:
        ref = run(0)
        for _ in range(3):
            got = run(1)
            assert len(got) == len(ref)
            for a, r in zip(got, ref):
                assert torch.equal(a, r)
            assert int(nn_ops._Scratch.counters(DEV).abs().sum()) == 0
    finally:
        hip_ops.set_option("bn_onelaunch", 0)
        hip_ops.set_option("bn_fused_small", 1)


def test_deferred_multi_tensor_wgrad_reduce_equals_per_layer_reduce():
    """conv_wgrad(defer=True) x several layers + ONE im2im_wgrad_reduce_multi == conv_wgrad's own reduction, bit for bit, at both
    launch widths; 3x3 lazy / split-input layers and a 1x1; more layers than one launch holds."""
    from im2im_uq_amd import nn_ops
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    layers = [(4, 64, 64, 64, 64, 9, False), (4, 32, 32, 128, 256, 9, False), (2, 16, 16, 512, 512, 9, False), (4, 64, 64, 128, 64, 9, True),
              (3, 40, 24, 256, 128, 9, False), (4, 64, 64, 64, 32, 1, False)] * 6            # 36 > REDUCE_MULTI_MAX = 32
    ops = []
    for b, h, w, ci, co, taps, split in layers:
        cin = ci // 2 if split else ci
        x = torch.randn(b, h, w, cin, generator=g).to(BF).to(DEV)
        xh = torch.randn(b, h, w, cin, generator=g).to(BF).to(DEV) if split else None
        dz = torch.randn(b, h, w, co, generator=g).to(BF).to(DEV)
        ss = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5]).to(DEV) if taps == 9 else None
        ops.append((x, dz, taps, ss, xh))
    side = nn_ops.side_stream(DEV)
    for on_side in (False, True):                           # width 256 (alone) and 128 (from the weight-gradient stream)
        st = side if on_side else torch.cuda.current_stream(DEV)
        st.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(st):
            ref = [nn_ops.conv_wgrad(x, dz, taps, x_ss=ss, x_hi=xh, scratch_key="t") for x, dz, taps, ss, xh in ops]
            outs = [nn_ops.conv_wgrad(x, dz, taps, x_ss=ss, x_hi=xh, scratch_key=f"t{i}", defer=True) for i, (x, dz, taps, ss, xh) in enumerate(ops)]
            assert len(nn_ops._pending_reduce[0]) == len(ops)
            nn_ops.flush_wgrad_reduce(0)
            assert not nn_ops._pending_reduce.get(0)
        torch.cuda.synchronize()
        for a, r in zip(outs, ref):
            assert torch.equal(a, r) and bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0
    nn_ops._wgrad_calls[0] = 0


def test_train_step_with_deferred_reduce_and_one_launch_sums_is_bit_identical_to_the_round5_launch_sequence():
    """a whole bf16 train step (depth-4 UNet, 64x64, batch 6): losses, every gradient and every updated weight / buffer equal with the
    round-6 launch sequence (one-launch BatchNorm sums, deferred multi-tensor weight-gradient reduce) and with the per-layer one."""
    from im2im_uq_amd import hip_ops, nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    g = torch.Generator().manual_seed(12)
    x, y = torch.randn(6, 1, 64, 64, generator=g).to(DEV), torch.rand(6, 1, 64, 64, generator=g).to(DEV)

    def run(new):
        nn_ops.set_compute_dtype("bf16")
        hip_ops.set_option("bn_onelaunch", int(new))
        was = nn_ops.WGRAD_DEFER_REDUCE
        nn_ops.WGRAD_DEFER_REDUCE = bool(new)
        try:
            torch.manual_seed(7)
            model = add_uncertainty(UNet(1, 1), dict(params)).to(DEV).train()
            opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
            losses, grads = [], None
            for _ in range(3):
                loss = model.loss_fn(model(x), y)
                opt.zero_grad()
                loss.backward()
                nn_ops.join_side_streams()
                grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
                opt.step()
                losses.append(float(loss))
            torch.cuda.synchronize()
            return losses, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}
        finally:
            nn_ops.WGRAD_DEFER_REDUCE = was
            hip_ops.set_option("bn_onelaunch", 0)
    ref, new = run(False), run(True)
    assert ref[0] == new[0]
    for k in ref[1]:
        assert torch.equal(ref[1][k], new[1][k]), k
    for k in ref[2]:
        assert torch.equal(ref[2][k], new[2][k]), k


def test_first_conv_weight_gradient_keeps_its_bits_beside_another_process():
    """[r6] smallconv_wgrad_vec_kernel with the compiler's v_pk_fma_f32 op_sel:[0,1,0] returned a different weight gradient on every
    launch while ANOTHER PROCESS kept the GPU busy (profiles/r06_multiprocess_determinism.txt); the scalar form must not.  A second
    process runs bf16 training steps (tools/debug_victim.py aggressor); this one launches the kernel on fixed inputs and compares bits."""
    import os
    import subprocess
    import sys
    import time
    from im2im_uq_amd import nn_ops
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    log = open(os.devnull, "w")
    aggressor = subprocess.Popen([sys.executable, os.path.join(root, "tools", "debug_victim.py"), "aggressor", "28"], stdout=log, stderr=log)
    try:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(10, 1, 320, 320, generator=g).to(DEV)
        cases = {dt: (torch.randn(10, 320, 320, 64, generator=g) * 1e-6).to(DEV).to(dt) for dt in (torch.bfloat16, torch.float32)}
        first = {dt: nn_ops.smallconv_wgrad(x, dz, True, False)[0].clone() for dt, dz in cases.items()}
        torch.cuda.synchronize()
        time.sleep(14)                                          # the other process has imported torch and is stepping
        assert aggressor.poll() is None, "the aggressor process ended early"
        bad = 0
        t0 = time.time()
        while time.time() - t0 < 8:
            for dt, dz in cases.items():
                bad += int(not torch.equal(nn_ops.smallconv_wgrad(x, dz, True, False)[0], first[dt]))
            torch.cuda.synchronize()
        assert bad == 0, f"{bad} launches differ from the first"
    finally:
        aggressor.wait(timeout=120)
        log.close()
