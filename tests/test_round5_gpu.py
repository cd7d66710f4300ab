"""Round 5: the configurations the earlier suites ran only in part.

(a) BASELINE configs[3] AS NAMED -- `UNet(1, 1, depth=5)` on a full 1024 x 1024 tile (the reference's TEMCA setting,
    experiments/temca_test/config.yml:74-77: 1024-pixel crops; a deeper trunk than core/models/trunks/unet.py:20-31 hard-codes):
    the 32 x 32 bottleneck, the 1024 + 1024-channel concatenation (K = 18,432) and 4,096 pixel tiles per image per layer.
    The CPU oracle runs the WHOLE image (bilinear upsampling with align_corners=True is not local: a band computed on its
    own is a different function, so a band oracle would not be one), ~10 s on 8 cores.
(b) fp8 twin of tests/test_train_parity_gpu.py: 600 Adam steps at TWO input channels (configs[4] is multi-channel,
    experiments/bsbcm_test/config.yml:16-17), fp8 vs fp32 with the fp32-vs-fp32' yardstick, then calibration
    (core/scripts/train.py:141-165, calibrate_model.py:89-145).
(c) a gradient spike (x100) through the delayed e5m2 scale of the fp8 backward: finite, and back to normal two steps later.
"""
import numpy as np
import pytest
import torch
from torch.utils.data import TensorDataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1,
              alpha=0.1, delta=0.1, num_lambdas=100, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6,
              device=DEV, dataset="synthetic", batch_size=16, lr=3e-4, input_normalization="standard",
              output_normalization="min-max", num_validation_images=2)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(autouse=True)
def _restore_dtype():
    from im2im_uq_amd import nn_ops
    yield
    nn_ops.set_compute_dtype("bf16")


def _build(depth, dt, n_in=1):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from oracle import model as om
    nn_ops.set_compute_dtype(dt)
    model = add_uncertainty(UNet(n_in, 1, depth=depth), dict(PARAMS))
    model.load_state_dict(om.det_state(n_in, 1, depth=depth))
    return model.to(DEV)


# ------------------------------------------------------------------------------------------------ (a) configs[3]
@pytest.fixture(scope="module")
def temca_oracle():
    """the CPU oracle on ONE full 1024 x 1024 image through the depth-5 network, eval mode (whole image: see the docstring)"""
    from oracle import model as om
    x, y = om.det_images(2, 1, 1024, 1024, salt=4)
    with torch.no_grad():
        ref = om.model_forward(x[:1], om.det_state(1, 1, depth=5), training=False)
    return x, y, ref


def test_configs3_depth5_1024_fp32_mode_eval_forward_equals_the_oracle(temca_oracle):
    """parity mode (exact-fp32 MFMA) at the named size and depth: every pixel of the three output planes within 5e-4 of the
    oracle's (the bound the 64 x 1024 strip test holds; outputs are O(0.2))."""
    x, _, ref = temca_oracle
    model = _build(5, "fp32").eval()
    with torch.no_grad():
        out = model(x[:1].to(DEV))
    assert out.shape == (1, 3, 1, 1024, 1024)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-4)


def test_configs3_depth5_1024_bf16_eval_forward_and_train_step(temca_oracle):
    """the benchmarked mode at the named size and depth: eval forward within 3 % relative L2 of the fp32 oracle on the whole
    image AND on a 64-row band through its middle (rows 480-543: every level's interior tiles), bit-reproducible; one
    training step at batch 2 (the per-GPU batch bench.py uses for this config): finite loss and gradients for every
    parameter that has one, bit-reproducible from the same state, loss within 3 % of the oracle's train-mode forward."""
    from im2im_uq_amd import nn_ops
    from oracle import model as om
    x, y, ref = temca_oracle
    model = _build(5, "bf16").eval()
    with torch.no_grad():
        out = model(x[:1].to(DEV))
        out2 = model(x[:1].to(DEV))
    assert out.shape == (1, 3, 1, 1024, 1024) and bool(torch.isfinite(out).all())
    assert torch.equal(out, out2)
    e_all, e_band = rel_l2(out.cpu(), ref), rel_l2(out.cpu()[..., 480:544, :], ref[..., 480:544, :])
    print(f"\n[configs3 bf16 eval] rel-L2 whole image {e_all:.4f}  rows 480-543 {e_band:.4f}")
    assert e_all < 3e-2 and e_band < 3e-2
    # one training step, twice from the same state
    st0 = {k: v.clone() for k, v in model.state_dict().items()}
    xd, yd = x.to(DEV), y.to(DEV)
    runs = []
    for _ in range(2):
        model.load_state_dict(st0)
        model.train()
        for p in model.parameters():
            p.grad = None
        pred = model(xd)
        assert pred.shape == (2, 3, 1, 1024, 1024)
        loss = model.loss_fn(pred, yd)
        loss.backward()
        nn_ops.join_side_streams()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        runs.append((loss.detach().clone(), grads))
    (l0, g0), (l1, g1) = runs
    assert bool(torch.isfinite(l0)) and torch.equal(l0, l1)
    assert len(g0) >= 2 * 11 * 2 and all(bool(torch.isfinite(v).all()) for v in g0.values())
    assert all(torch.equal(g0[n], g1[n]) for n in g0)
    assert any(float(v.abs().max()) > 0 for n, v in g0.items() if "down5" in n)          # the deepest level got a gradient
    with torch.no_grad():
        ref_loss = om.quantile_loss(om.model_forward(x, om.det_state(1, 1, depth=5), training=True), y, PARAMS)
    print(f"[configs3 bf16 train] loss HIP {l0.item():.5f} oracle {ref_loss.item():.5f}")
    assert l0.item() == pytest.approx(ref_loss.item(), rel=3e-2)


# ------------------------------------------------------------------------------------------------ (b) fp8 training parity
def _run(dt, data, steps, n_in, perturb=0.0, seed=99):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.eval import eval_set_metrics
    from oracle import model as om
    nn_ops.set_compute_dtype(dt)
    model = add_uncertainty(UNet(n_in, 1), dict(PARAMS))
    st = om.det_state(n_in, 1)
    if perturb:
        g = torch.Generator().manual_seed(seed)
        st = {k: (v * (1 + perturb * torch.randn(v.shape, generator=g)) if om.is_param(k) else v) for k, v in st.items()}
    model.load_state_dict(st)
    model = model.to(DEV).train()
    opt = nn_ops.FusedAdam(model.parameters(), lr=PARAMS["lr"])
    (xt, yt), (xc, yc), (xv, yv) = data
    nb = xt.shape[0] // 16
    losses = []
    for step in range(steps):
        s = (step % nb) * 16
        loss = model.loss_fn(model(xt[s:s + 16]), yt[s:s + 16])
        losses.append(loss.detach())
        opt.zero_grad(); loss.backward(); opt.step()
    losses = torch.stack(losses).cpu().numpy()
    cfg = dict(PARAMS)
    model, _ = calibrate_model(model, TensorDataset(xc, yc), cfg)
    torch.manual_seed(0); np.random.seed(0)
    risk = eval_set_metrics(model, TensorDataset(xv, yv), cfg)[0]
    with torch.no_grad():
        lo, mid, hi = model.nested_sets((xv,))
    return dict(losses=losses, lhat=float(model.lhat), risk=float(risk), lo=lo.float().cpu(), mid=mid.float().cpu(),
                hi=hi.float().cpu())


def _distance(a, ref):
    dl = 6.0 / 99
    return dict(loss=abs(a["losses"][-200:].mean() / ref["losses"][-200:].mean() - 1.0), lhat=abs(a["lhat"] - ref["lhat"]) / dl,
                mid=rel_l2(a["mid"], ref["mid"]), lo=rel_l2(a["lo"], ref["lo"]), hi=rel_l2(a["hi"], ref["hi"]))


def test_fp8_training_tracks_fp32_training_then_calibrates_alike():
    """600 steps, 2 input channels, 64 x 64: fp8 mode (e4m3 forward, e5m2 data-gradient under delayed scaling, fp8 weight gradient
    where routed) against fp32 mode, with fp32 against a 1e-4-perturbed fp32 run as the yardstick and the bf16 run beside it.
    Measured on MI355X (profiles/r05_tests_round5.txt; fp32' | bf16 | fp8, all vs fp32): tail-200 loss 1.8 % | 5.3 % | 4.7 %;
    lambda-hat 3 | 2 | 3 grid steps of 100; prediction images (rel. L2) 4.6 % | 3.3 % | 5.5 %; calibrated lower edge 7.9 % | 4.9 % |
    8.7 %; upper edge 13.2 % | 5.6 % | 9.6 %; mean calibrated interval size 0.148 | 0.117 | 0.136 (fp32 0.130); validation risk
    0.047-0.050 for all four (alpha = 0.1): after 600 steps the fp8 run is as far from the fp32 run as a second fp32 run is.
    The bounds asserted are the bf16 test's absolute ones widened for e4m3 / e5m2 operands (method: module docstring of
    tests/test_train_parity_gpu.py)."""
    from im2im_uq_amd.core.datasets.synthetic import SyntheticDenoiseDataset
    steps, n_in = 600, 2
    ds = SyntheticDenoiseDataset(num_images=96 * 3, num_inputs=n_in, side=64, noise=0.1, seed=5)
    x, y = ds.x.to(DEV), ds.y.to(DEV)
    data = ((x[:96], y[:96]), (x[96:192], y[96:192]), (x[192:], y[192:]))
    r32 = _run("fp32", data, steps, n_in)
    r32b = _run("fp32", data, steps, n_in, perturb=1e-4)
    r16 = _run("bf16", data, steps, n_in)
    r8 = _run("fp8", data, steps, n_in)
    d8, d16, dself = _distance(r8, r32), _distance(r16, r32), _distance(r32b, r32)
    size = {k: float((r["hi"] - r["lo"]).mean()) for k, r in (("fp32", r32), ("fp32'", r32b), ("bf16", r16), ("fp8", r8))}
    print(f"\n[fp8 train parity] loss0 fp32 {r32['losses'][0]:.4f} fp8 {r8['losses'][0]:.4f}  tail200 fp32 {r32['losses'][-200:].mean():.5f} "
          f"fp32' {r32b['losses'][-200:].mean():.5f} bf16 {r16['losses'][-200:].mean():.5f} fp8 {r8['losses'][-200:].mean():.5f}\n"
          f"  lhat fp32 {r32['lhat']:.4f} fp32' {r32b['lhat']:.4f} bf16 {r16['lhat']:.4f} fp8 {r8['lhat']:.4f}   "
          f"val risk {r32['risk']:.4f} / {r32b['risk']:.4f} / {r16['risk']:.4f} / {r8['risk']:.4f}\n"
          f"  fp32' vs fp32: " + "  ".join(f"{k} {v:.4f}" for k, v in dself.items()) + "\n"
          f"  bf16  vs fp32: " + "  ".join(f"{k} {v:.4f}" for k, v in d16.items()) + "\n"
          f"  fp8   vs fp32: " + "  ".join(f"{k} {v:.4f}" for k, v in d8.items()) + "\n"
          f"  mean calibrated interval size: " + "  ".join(f"{k} {v:.4f}" for k, v in size.items()))
    for r in (r32, r32b, r16, r8):
        assert np.isfinite(r["losses"]).all()
        assert r["losses"][-200:].mean() < 0.1 * r["losses"][0]                       # actually trained
        assert 0 < r["lhat"] < 6                                                      # the scan stopped inside the grid
        assert r["risk"] <= PARAMS["alpha"]                                           # the calibrated sets hold the risk
    assert r8["losses"][0] == pytest.approx(r32["losses"][0], rel=5e-2)               # same start (e4m3 operands: percents)
    # absolute bounds: the bf16 test's (10 % / 12 % / 20 % / 15 %) widened for e4m3 / e5m2 operands
    assert d8["loss"] < 0.20
    assert d8["mid"] < 0.16 and d8["lo"] < 0.25 and d8["hi"] < 0.20
    # [r6] relative to the yardstick (see tests/test_train_parity_gpu.py: two fp32 runs alone can sit at a ratio of 0.75), plus a sanity range
    assert abs(np.log(size["fp8"] / size["fp32"])) <= 2.5 * abs(np.log(size["fp32'"] / size["fp32"])) + np.log(1.3), size
    assert 0.5 < size["fp8"] / size["fp32"] < 2.0
    # and relative to the yardstick: no further from fp32 than 2.5x what a second fp32 run is, plus a margin
    assert d8["loss"] <= 2.5 * dself["loss"] + 0.10
    assert d8["lhat"] <= 2.5 * dself["lhat"] + 4.0
    for k in ("mid", "lo", "hi"):
        assert d8[k] <= 2.5 * dself[k] + 0.06, (k, d8[k], dself[k])


# ------------------------------------------------------------------------------------------------ (c) gradient spike
def test_fp8_backward_survives_a_gradient_spike_and_recovers_within_two_steps():
    """delayed scaling sizes the e5m2 operand of step t from max|dz| of step t-1 with 3.5x headroom (csrc/conv_fp8.hip GRAD form,
    nn_ops.Fp8GradScale).  A loss that jumps x100 for ONE step (a bad batch) overshoots that headroom 28-fold: the conversion must
    saturate, not overflow to inf / NaN; the step after runs on a scale 100x too large (values 6-7 binades further down e5m2's
    16-binade range); two steps after the spike the error is back at the steady-state level.  Weights are fixed, so every step's
    exact answer is the same bf16-backward gradient times the step's factor.  Measured (profiles/r05_tests_round5.txt): median
    relative L2 of the parameter gradients vs the bf16 backward 0.133 at every normal step -- the step right after the spike
    included: e5m2's range absorbs a 100x too large scale -- and 0.914 on the spike step itself (saturated, finite)."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("fp8")
    torch.manual_seed(0)
    model = add_uncertainty(UNet(2, 1), dict(PARAMS)).to(DEV).train()
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(4, 2, 64, 64, generator=g).to(DEV), torch.rand(4, 1, 64, 64, generator=g).to(DEV)

    def grads_of(factor):
        for p in model.parameters():
            p.grad = None
        loss = model.loss_fn(model(x), y) * factor
        loss.backward()
        nn_ops.join_side_streams()
        return {n: (p.grad / factor).float().cpu() for n, p in model.named_parameters() if p.grad is not None}

    was = nn_ops.FP8_DGRAD
    try:
        nn_ops.FP8_DGRAD = False                                   # reference: the same fp8 forward, bf16 backward
        ref = grads_of(1.0)
        nn_ops.FP8_DGRAD = True
        med, worst, finite = [], [], []
        for factor in (1.0, 1.0, 100.0, 1.0, 1.0, 1.0):
            got = grads_of(factor)
            finite.append(all(bool(torch.isfinite(v).all()) for v in got.values()))
            errs = sorted(rel_l2(got[n], ref[n]) for n in got if float(ref[n].abs().max()) > 0)
            med.append(errs[len(errs) // 2]); worst.append(errs[-1])
    finally:
        nn_ops.FP8_DGRAD = was
    print("\n[fp8 spike] factor 1,1,100,1,1,1: median rel-L2 vs bf16 backward " + " ".join(f"{m:.3f}" for m in med)
          + "   worst " + " ".join(f"{m:.3f}" for m in worst))
    assert all(finite), finite                                     # the spike step saturates, it does not poison
    assert med[0] < 0.15 and med[1] < 0.15                         # steady state (tests/test_fp8_gpu.py holds the same bound)
    assert med[4] < 0.15 and med[5] < 0.15                         # recovered two steps after the spike
    assert med[4] <= med[1] * 1.5 + 0.02
    assert med[3] < 0.3                                            # the step on the stale (too large) scale (measured: no worse)


# ------------------------------------------------------------------------------------------------ ADVICE r4 (host logic on the GPU)
def test_fused_adam_capturable_with_two_param_groups_equals_torch_adam():
    """two param groups on ONE device at the same step count each own a device step counter (round 4 keyed them by
    (device, step): the second group's lookup missed, allocated a fresh counter and dropped the first one's memory)."""
    from im2im_uq_amd import nn_ops
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(n, generator=g) for n in (1000, 37, 4096, 5)]
    grads = [[torch.randn(w.shape, generator=g) for w in ws] for _ in range(4)]
    mine = [w.clone().to(DEV).requires_grad_(True) for w in ws]
    ref = [w.clone().requires_grad_(True) for w in ws]
    opt = nn_ops.FusedAdam([dict(params=mine[:2], lr=1e-2), dict(params=mine[2:], lr=3e-3, betas=(0.8, 0.99))], capturable=True)
    topt = torch.optim.Adam([dict(params=ref[:2], lr=1e-2), dict(params=ref[2:], lr=3e-3, betas=(0.8, 0.99))])
    ctr_ids = None
    for step in grads:
        for p, q, gr in zip(mine, ref, step):
            p.grad, q.grad = gr.to(DEV), gr.clone()
        opt.step(); topt.step()
        ids = sorted(c[0].data_ptr() for c in opt._ctrs.values())
        assert len(ids) == 2 and len(set(ids)) == 2                # one live counter per group ...
        assert ctr_ids is None or ids == ctr_ids                   # ... and the same two device words at every step
        ctr_ids = ids
    assert opt._ctr_allocs == 2
    assert sorted(int(c[0].item()) for c in opt._ctrs.values()) == [4, 4]
    for p, q in zip(mine, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=2e-6, atol=1e-7)
    assert opt.hyper_key() == ((1e-2, 0.9, 0.999, 1e-8), (3e-3, 0.8, 0.99, 1e-8))


def test_graphed_step_follows_a_learning_rate_change():
    """lr / betas / eps enter the captured Adam launch by value: GraphedStep re-captures when `group['lr']` changes (an LR
    scheduler), so a graphed run with a schedule stays bit-identical to the eager loop -- and releases its pin on the scratch
    buffers when it drops a graph."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import GraphedStep
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    nn_ops.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(12)
    batches = [(torch.randn(8, 1, 32, 32, generator=g), torch.rand(8, 1, 32, 32, generator=g)) for _ in range(12)]

    def run(graph):
        torch.manual_seed(4)
        model = add_uncertainty(UNet(1, 1, depth=2), dict(params)).to(DEV).train()
        opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
        gs = GraphedStep(model, opt) if graph else None
        losses, captures = [], 0
        for i, (x, y) in enumerate(batches):
            if i in (6, 9):
                opt.param_groups[0]["lr"] *= 0.1               # what a StepLR does
            x, y = x.to(DEV), y.to(DEV)
            before = gs.graph if gs else None
            loss = gs.step((x,), y) if gs else None
            if gs and gs.graph is not None and gs.graph is not before:
                captures += 1
            if loss is None:
                loss = model.loss_fn(model(x), y)
                opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        if graph:
            assert captures == 3 and gs.replays >= 8           # first capture + one per lr change
            assert nn_ops._Scratch.graphs >= 1
            gs._drop_graph()
            assert gs.graph is None
        return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    graphs_before = nn_ops._Scratch.graphs
    le, se = run(False)
    lg, sg = run(True)
    assert nn_ops._Scratch.graphs == graphs_before and (graphs_before > 0 or not nn_ops._Scratch.pinned)
    assert torch.equal(le, lg), (le, lg)
    for k in se:
        assert torch.equal(se[k], sg[k]), k


def test_eval_fused_outconv_respects_hooks_and_a_dtype_override_on_outconv():
    """UNet.forward hands OutConv's 1x1 to the last block's epilogue only when nobody would notice: a forward hook on `net.out`
    (feature extraction) still fires, and an OutConv with its own `compute_dtype` runs in that dtype -- both give the unfused
    path's result."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    nn_ops.set_compute_dtype("bf16")
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    torch.manual_seed(1)
    model = add_uncertainty(UNet(1, 1, depth=2), dict(params)).to(DEV)
    x = torch.randn(2, 1, 64, 64, device=DEV)
    with torch.no_grad():
        model.train(); model(x); model.eval()
        assert getattr(model.baseModel(x), "_im2im_tail_done", False)             # the fused path is the default
        plain_was = nn_ops.FUSE_EVAL_OUTCONV
        try:
            nn_ops.FUSE_EVAL_OUTCONV = False
            ref = model(x)
        finally:
            nn_ops.FUSE_EVAL_OUTCONV = plain_was
        seen = []
        h = model.baseModel.out.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
        out = model(x)
        h.remove()
        assert seen == [(2, 32, 64, 64)] and torch.equal(out, ref)
        model.baseModel.out.compute_dtype = torch.float32                          # per-module override (unet_parts._cdt)
        feat = model.baseModel(x)
        assert not getattr(feat, "_im2im_tail_done", False) and feat.dtype == torch.float32
        model.baseModel.out.compute_dtype = None
        assert torch.equal(model(x), ref)
        # [r6] a hook on the LAST Up block (or a module inside it) must see that block's own 64-channel activation, and a
        # global module forward hook must see every module's real output: both switch the fused tail off
        seen = []
        h = model.baseModel.up2.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
        out = model(x)
        h.remove()
        assert seen == [(2, 64, 64, 64)] and torch.equal(out, ref)
        seen = []
        h = model.baseModel.up2.conv.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
        out = model(x)
        h.remove()
        assert seen == [(2, 64, 64, 64)] and torch.equal(out, ref)
        shapes = {}
        h = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: shapes.__setitem__(type(m).__name__, tuple(getattr(o, "shape", ()))))
        out = model(x)
        h.remove()
        assert shapes.get("Up") == (2, 64, 64, 64) and shapes.get("OutConv") == (2, 32, 64, 64) and torch.equal(out, ref)
        assert getattr(model.baseModel(x), "_im2im_tail_done", False)             # nobody listening any more: fused again


# ------------------------------------------------------------------------------------------------ conv_roll64_kernel (csrc/conv_roll.hip)
ROLL_CASES = [
    # B, H, W, Ci, Co, split input, lazy input
    (1, 64, 64, 96, 64, False, False),       # three 32-channel chunks = six units, every tile touches an image edge
    (3, 64, 128, 64, 64, False, True),       # odd batch, lazy BatchNorm+ReLU on the staged input
    (2, 96, 80, 128, 64, True, True),        # [skip, upsampled] from two tensors, only the first one lazy (the Up block's first conv)
    (1, 64, 64, 32, 192, False, False),      # three 64-channel output blocks per pixel tile, one unit pair
    (5, 320, 320, 64, 64, False, True),      # BASELINE layer shape: 1,000 tiles on 512 persistent workgroups (runs of 1-2 tiles)
]


@pytest.mark.parametrize("case", ROLL_CASES)
def test_conv_roll64_kernel_vs_cpu_and_vs_conv_igemm(case):
    """the persistent, software-pipelined kernel of the 64-output-channel full-resolution layers (option conv_roll = 1; the default
    routes these launches to conv_igemm_kernel) against F.conv2d in fp32 on operands quantised as the kernel sees them (bf16
    tolerance 1.5e-2 relative L2, worst element 6 % of the RMS) and against conv_igemm_kernel on the same inputs: same values up
    to the order of the fp32 sums (1e-4 relative L2; one bf16 ulp on single elements), same BatchNorm partial statistics
    (count exact, mean / M2 to 1e-5 after merging), forward with statistics, the folded eval epilogue and the plain
    data-gradient form."""
    import torch.nn.functional as F
    from im2im_uq_amd import _lib, hip_ops, nn_ops
    from test_kernels_gpu import merged_moments
    try:                                                  # [r6] the kernel is only in libraries built with IM2IM_BUILD_EXPERIMENTAL=1
        hip_ops.set_option("conv_roll", 1)
    except _lib.Im2ImError:
        pytest.skip("library built without the experimental kernels (IM2IM_BUILD_EXPERIMENTAL=1 python -m im2im_uq_amd.build)")
    finally:
        hip_ops.set_option("conv_roll", 0)
    b, h, w, ci, co, split, lazy = case
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    cin = ci // 2 if split else ci
    x = torch.randn(b, h, w, cin, generator=g).to(BF)
    xh = torch.randn(b, h, w, cin, generator=g).to(BF) if split else None
    wt = torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5
    bias = torch.randn(co, generator=g) * 0.1
    ss = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5]) if lazy else None
    fold = torch.stack([torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)])
    # CPU reference: the staged operand is bf16(max(x * scale + shift, 0)) for the lazy source, x itself otherwise
    a_lo = x.float()
    if lazy:
        a_lo = torch.relu(a_lo * ss[0] + ss[1]).to(BF).float()
    a_full = torch.cat([a_lo, xh.float()], dim=3) if split else a_lo
    ref = F.conv2d(a_full.permute(0, 3, 1, 2), wt.to(BF).float(), bias, padding=1)
    wf, wd = nn_ops.pack_weight(wt.to(DEV), BF)
    xd, xhd, ssd = x.to(DEV), (xh.to(DEV) if split else None), (ss.to(DEV) if lazy else None)
    got = {}
    try:
        for mode in (0, 1):
            hip_ops.set_option("conv_roll", mode)
            y, st = nn_ops.conv_fwd(xd, wf, bias.to(DEV), want_stats=True, in_ss=ssd, x_hi=xhd)
            y2 = nn_ops.conv_fwd(xd, wf, None, fold.to(DEV), relu=True, in_ss=ssd, x_hi=xhd)
            y3 = nn_ops.conv_fwd(xd, wf, in_ss=ssd, x_hi=xhd)                 # EPI 0 (what a data-gradient launch is)
            torch.cuda.synchronize()
            got[mode] = (y.float().cpu(), st.cpu(), y2.float().cpu(), y3.float().cpu())
    finally:
        hip_ops.set_option("conv_roll", 0)
    y, st, y2, y3 = got[1]
    ref_nhwc = ref.permute(0, 2, 3, 1)
    assert rel_l2(y, ref_nhwc) < 1.5e-2
    assert float((y - ref_nhwc).abs().max()) < 6e-2 * float(ref_nhwc.pow(2).mean().sqrt())
    ref2 = torch.relu((ref_nhwc - bias) * fold[0] + fold[1])
    assert rel_l2(y2, ref2) < 1.5e-2
    assert rel_l2(y3, ref_nhwc - bias) < 1.5e-2
    for k in (0, 2, 3):                                       # the same numbers as conv_igemm_kernel up to summation order
        assert rel_l2(got[1][k], got[0][k]) < 1e-4, k
    n1, m1, q1 = merged_moments(st)
    n0, m0, q0 = merged_moments(got[0][1])
    assert torch.equal(n1, n0) and float((n1 - b * h * w).abs().max()) == 0.0
    yst = y.double().reshape(-1, co)
    np.testing.assert_allclose(m1.numpy(), yst.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q1.numpy(), ((yst - yst.mean(0)) ** 2).sum(0).numpy(), rtol=2e-5)


# ------------------------------------------------------------------------------------------------ G19: the train loop on well-conditioned weights
def test_g19_adam_trajectory_on_default_init_fp32():
    """fixture G19 (tests/golden/make_golden.py g19): TEN steps of the reference's inner training loop (core/scripts/train.py:141-165,
    torch.optim.Adam lr 1e-4) on the reference's own default initialisation (depth-2 / base-32 assembly of its DoubleConv / Down /
    Up / OutConv, noise images) -- weights on which the trajectory is well-conditioned, unlike G5's closed-form ones.  fp32 mode:
    the WHOLE loss trajectory within 1e-3 relative, the eval-mode probe within 1e-3 relative L2, and after the ten steps every
    state_dict tensor equals the reference's within what ten Adam steps can differ by -- EXCEPT the documented ones: the ten
    pre-BatchNorm conv biases (the reference feeds Adam their rounding-noise gradients and they drift by up to 2.7e-4; here their
    gradient is exactly zero and they stay put) and, with them, the running means they shift (running_mean - bias,
    what eval mode uses, is compared instead)."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from conftest import load_golden
    g = load_golden("g19_adam_trajectory_default_init")
    nn_ops.set_compute_dtype("fp32")
    model = add_uncertainty(UNet(1, 1, depth=int(g["depth"]), base=int(g["base"])), dict(PARAMS))
    init = {k[len("init."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    assert list(model.state_dict().keys()) == list(init.keys())
    model.load_state_dict(init)
    model = model.to(DEV).train()
    lr = float(g["lr"])
    opt = nn_ops.FusedAdam(model.parameters(), lr=lr)
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    losses = []
    for s in range(x.shape[0]):
        loss = model.loss_fn(model(x[s]), y[s])
        losses.append(loss.item())
        opt.zero_grad(); loss.backward(); opt.step()
    print("\n[g19] losses HIP " + " ".join(f"{v:.6f}" for v in losses) + "\n      reference " + " ".join(f"{v:.6f}" for v in g["losses"]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-3)                  # the whole trajectory
    assert losses[-1] < 0.85 * losses[0]
    model.eval()
    with torch.no_grad():
        probe = model(x[0])
    assert rel_l2(probe.cpu(), g["probe_out"]) < 1e-3
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pre_bn_bias = [k for k in sd if k.endswith(".double_conv.0.bias") or k.endswith(".double_conv.3.bias")]
    assert len(pre_bn_bias) == 10
    worst = {}
    for k, v in sd.items():
        ref = torch.from_numpy(g["final." + k])
        if k in pre_bn_bias:
            assert torch.equal(v, init[k])                                      # zero gradient: Adam never moves them here
            assert 1e-5 < float((ref - init[k]).abs().max()) < 5 * 10 * lr     # ... while the reference's drift on rounding noise
            continue
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref) == x.shape[0]
            continue
        slack = 0.0
        if k.endswith("running_mean"):
            # an average of the batch means of conv + bias over the ten steps: the reference's carries its drifting bias with the
            # weights of the ten steps, so minus the FINAL bias it still differs by up to that layer's drift -- nothing else
            bias_k = k[:-len("1.running_mean")] + "0.bias" if k.endswith(".1.running_mean") else k[:-len("4.running_mean")] + "3.bias"
            ref_bias = torch.from_numpy(g["final." + bias_k])
            slack = float((ref_bias - init[bias_k]).abs().max())
            v, ref = v - sd[bias_k], ref - ref_bias
        d = (v.double() - ref.double()).abs()
        worst[k] = (float(d.max()), float(d.median()), float(ref.abs().max()), slack)
    big = sorted(worst.items(), key=lambda kv: -kv[1][0])[:4]
    print("[g19] largest per-tensor max |diff| after 10 steps: " + "; ".join(f"{k} {v[0]:.2e} (median {v[1]:.1e})" for k, v in big))
    notrm = {k: v for k, v in worst.items() if not k.endswith("running_mean")}
    print("[g19] ... outside the running means: " + "; ".join(f"{k} {v[0]:.2e} (median {v[1]:.1e})" for k, v in sorted(notrm.items(), key=lambda kv: -kv[1][0])[:4]))
    for k, (m, md, scale, slack) in worst.items():
        # a weight whose gradient is ~0 can take +-lr steps of either sign in two fp32 implementations: bound = the ten steps; the
        # typical element must agree to rounding noise (+ the bias drift the reference's running means carry, see above)
        assert m <= 2 * 10 * lr + 1e-5 * scale + slack, (k, m)
        assert md <= 2e-6 + 1e-5 * scale + slack, (k, md)


# ------------------------------------------------------------------------------------------------ the LEAN first conv (csrc/smallconv.hip)
_S2L_SNIPPET = r"""
import sys, torch
sys.path.insert(0, {root!r})
from im2im_uq_amd import nn_ops
g = torch.Generator().manual_seed(5)
x = torch.randn(3, 1, 64, 96, generator=g).cuda()
w = (torch.randn(1, 9, 64, generator=g) * 0.3).cuda()
b = (torch.randn(64, generator=g) * 0.1).cuda()
y, st = nn_ops.smallconv_s2l(x, w, b, None, 64, torch.bfloat16, want_stats=True)
fold = torch.stack([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)]).cuda()
y2 = nn_ops.smallconv_s2l(x, w, None, fold, 64, torch.bfloat16, relu=True, flip=True)      # the inference launch (conv_bn_relu_eval)
torch.cuda.synchronize()
torch.save({{"y": y.cpu(), "st": st.cpu(), "y2": y2.cpu()}}, {out!r})
"""


def test_lean_first_conv_is_bit_identical_to_the_generic_kernel(tmp_path):
    """[r5] smallconv_s2l_kernel<..., LEAN> (the training launch of the 1 -> 64 first conv, unet_parts.py:16 with in_channels = 1: eval-only
    epilogue code and bounds tests compiled out, the tile's passes in two rounds so three workgroups fit a CU; 0.45 -> 0.33 ms at batch
    78) against the generic form of the same kernel (IM2IM_SMALLCONV_VALU=16, read once per process -- hence two child processes):
    same output bits, same statistics rows."""
    import subprocess
    import sys as _sys
    from conftest import ROOT
    import os as _os
    outs = {}
    for name, mask in (("lean", "0"), ("generic", "16")):
        path = str(tmp_path / f"{name}.pt")
        env = dict(_os.environ, IM2IM_SMALLCONV_VALU=mask)
        r = subprocess.run([_sys.executable, "-c", _S2L_SNIPPET.format(root=ROOT, out=path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs[name] = torch.load(path)
    assert torch.equal(outs["lean"]["y"], outs["generic"]["y"])
    assert torch.equal(outs["lean"]["st"], outs["generic"]["st"])
    assert torch.equal(outs["lean"]["y2"], outs["generic"]["y2"]) and float(outs["lean"]["y2"].float().max()) > 0     # LEAN = 2, inference
    assert bool(torch.isfinite(outs["lean"]["y"].float()).all()) and float(outs["lean"]["y"].float().abs().max()) > 0
