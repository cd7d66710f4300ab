"""Pin the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import calibration as oc
from oracle import model as om
from conftest import load_golden

PARAMS = dict(q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1, beta=0.1)
T = torch.from_numpy


def test_g1_pinball():
    g = load_golden("g1_pinball")
    for q, tag in ((0.05, "005"), (0.95, "095")):
        o = T(g["output"]).clone().requires_grad_(True)
        loss = om.pinball(o, T(g["target"]), q)
        loss.backward()
        assert loss.item() == pytest.approx(float(g[f"loss_{tag}"]), rel=1e-6)
        np.testing.assert_allclose(o.grad.numpy(), g[f"grad_{tag}"], rtol=1e-6, atol=0)
        assert (o.grad[0, 0, :8] == 0).all()          # exact ties contribute no gradient


def test_g2_quantile_loss():
    g = load_golden("g2_quantile_loss")
    for params, lk, gk in ((PARAMS, "loss", "grad"),
                           (dict(q_lo_weight=0.5, q_hi_weight=2.0, mse_weight=3.0, q_lo=0.1, q_hi=0.8), "loss_w", "grad_w")):
        p = T(g["pred"]).clone().requires_grad_(True)
        loss = om.quantile_loss(p, T(g["target"]), params)
        loss.backward()
        assert loss.item() == pytest.approx(float(g[lk]), rel=1e-6)
        np.testing.assert_allclose(p.grad.numpy(), g[gk], rtol=1e-5, atol=1e-9)


def _part_state(g, prefix):
    st = {}
    for k, v in g.items():
        if k.startswith("state_before."):
            st[k[len("state_before."):]] = T(v).clone()
    return st


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_g3_doubleconv_down_up(mode):
    training = mode == "train"
    # DoubleConv
    g = load_golden("g3_doubleconv")
    st = {"baseModel.blk." + k: v for k, v in _part_state(g, "").items()}
    y = om.double_conv(T(g[f"{mode}.x0"]), st, "blk", training)
    np.testing.assert_allclose(y.numpy(), g[f"{mode}.y"], rtol=1e-4, atol=1e-5)
    if training:
        np.testing.assert_allclose(st["baseModel.blk.double_conv.1.running_mean"].numpy(),
                                   g["train.state_after.double_conv.1.running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st["baseModel.blk.double_conv.4.running_var"].numpy(),
                                   g["train.state_after.double_conv.4.running_var"], rtol=1e-5, atol=1e-6)
    # Down = maxpool + DoubleConv
    g = load_golden("g3_down")
    st = {"baseModel.blk." + k[len("maxpool_conv.1."):]: v for k, v in _part_state(g, "").items()}
    y = om.double_conv(torch.nn.functional.max_pool2d(T(g[f"{mode}.x0"]), 2), st, "blk", training)
    np.testing.assert_allclose(y.numpy(), g[f"{mode}.y"], rtol=1e-4, atol=1e-5)
    # Up (bilinear), with and without odd-size padding
    for name in ("g3_up_bilinear", "g3_up_bilinear_pad"):
        g = load_golden(name)
        st = {"baseModel.blk." + k[len("conv."):]: v for k, v in _part_state(g, "").items()}
        y = om.up_block(T(g[f"{mode}.x0"]), T(g[f"{mode}.x1"]), st, "blk", training)
        np.testing.assert_allclose(y.numpy(), g[f"{mode}.y"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n_in", [1, 2])
def test_g4_full_model_forward(n_in):
    g = load_golden(f"g4_model_fwd_nin{n_in}")
    st = om.det_state(n_in, 1)
    x = T(g["x"])
    with torch.no_grad():
        out = om.model_forward(x, st, training=False)
    np.testing.assert_allclose(out.numpy(), g["out_eval"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        out = om.model_forward(x, st, training=True)
    np.testing.assert_allclose(out.numpy(), g["out_train"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(st["baseModel.inc.double_conv.1.running_mean"].numpy(), g["rm_inc1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st["baseModel.up4.conv.double_conv.4.running_var"].numpy(), g["rv_up4"], rtol=1e-5, atol=1e-6)
    assert int(st["baseModel.inc.double_conv.1.num_batches_tracked"]) == int(g["nbt"])


def test_g4_state_spec_matches_reference_param_count():
    n = sum(int(np.prod(s)) for k, s in om.state_spec(1, 1) if om.is_param(k))
    assert n == 17_269_123          # BASELINE.md section 2, measured on the reference model


def test_g5_adam_trajectory():
    g = load_golden("g5_adam_trajectory")
    st = om.det_state(1, 1)
    batches = []
    for step in range(5):
        x, y = om.det_images(4, 1, 32, 32, salt=step)
        batches.append((x, y))
    losses = om.train_steps(st, batches, PARAMS, lr=float(g["lr"]))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4)
    with torch.no_grad():
        probe = om.model_forward(T(g["probe_x"]), st, training=False)
    np.testing.assert_allclose(probe.numpy(), g["probe_out"], rtol=0, atol=5e-3)
    k = "last_layer.upper.weight"
    np.testing.assert_allclose(st[k].flatten().numpy()[::max(1, st[k].numel() // 512)][:512], g["sample." + k], atol=2e-4)


def test_g19_adam_trajectory_on_default_init():
    """[r5] ten steps of the reference's inner loop (train.py:141-165) from its own default initialisation (fixture G19): the oracle
    walks the same trajectory -- every loss and every tensor of the final state_dict to 1e-6 (bit-equal on the torch build that
    generated the fixture), so the `-m gpu` test that compares the HIP path with G19 compares it with the oracle too."""
    g = load_golden("g19_adam_trajectory_default_init")
    st = {k[len("init."):]: T(v) for k, v in g.items() if k.startswith("init.")}
    assert om.unet_depth(st) == int(g["depth"]) == 2
    x, y = T(g["x"]), T(g["y"])
    losses = om.train_steps(st, [(x[i], y[i]) for i in range(x.shape[0])], PARAMS, lr=float(g["lr"]))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-6)
    for k, v in st.items():
        np.testing.assert_allclose(v.numpy(), g["final." + k], rtol=0, atol=1e-6, err_msg=k)
    with torch.no_grad():
        probe = om.model_forward(x[0], st, training=False)
    np.testing.assert_allclose(probe.numpy(), g["probe_out"], rtol=0, atol=1e-6)


def test_g6_nested_sets():
    g = load_golden("g6_nested_sets")
    out = T(g["output"])
    for i, lam in enumerate(T(g["lams"])):
        lo, mid, hi = oc.nested_sets(out, lam)
        assert np.array_equal(lo.numpy(), g["lower"][i])
        assert np.array_equal(hi.numpy(), g["upper"][i])


def _cfg(v):
    return dict(alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]), minimum_lambda=float(v[3]),
                maximum_lambda=float(v[4]), batch_size=int(v[5]))


@pytest.mark.parametrize("case", ["mid", "n130", "zero_risk", "no_stop", "c2"])
def test_g7_calibrate(case):
    g = load_golden("g7_calibrate_" + case)
    cfg = _cfg(g["cfg"])
    lhat, table, trace = oc.calibrate_from_outputs(T(g["output"]), T(g["label"]), cfg)
    assert np.array_equal(table.numpy(), g["table"])           # bit-exact, incl. zero columns (Q2)
    assert float(lhat) == float(g["lhat"])
    ref_trace = g["trace"]
    assert len(trace) == len(ref_trace)
    for (j, r, rp), (j2, r2, rp2) in zip(trace, ref_trace):
        assert j == int(j2) and r == r2 and rp == pytest.approx(rp2, abs=1e-12)


def test_g8_hb_bound():
    rows = load_golden("g8_hb_bound")["rows"]
    for muhat, n, delta, expect in rows:
        got = oc.hb_mu_plus(muhat, int(n), delta)
        assert got == pytest.approx(expect, abs=1e-12), (muhat, n, delta)
    # the reference's only known-answer style check, core/calibration/bounds.py:46
    assert oc.hb_mu_plus(0.1, 10000, 0.1) == pytest.approx(0.10551758004098838, abs=1e-12)


def test_g9_loss_table():
    g = load_golden("g9_loss_table")
    cfg = dict(num_lambdas=int(g["cfg"][0]), minimum_lambda=float(g["cfg"][1]), maximum_lambda=float(g["cfg"][2]))
    table = oc.loss_table_from_outputs(T(g["output"]), T(g["label"]), cfg)
    assert np.array_equal(table.numpy(), g["table"])


def test_g10_metrics():
    g = load_golden("g10_metrics")
    losses, spatial = oc.risk_and_miscoverage(T(g["output"]), T(g["label"]), T(g["lhat"]))
    assert np.array_equal(losses.numpy(), g["losses"])
    np.testing.assert_allclose(spatial, g["spatial"], rtol=1e-6, atol=1e-7)


UTYPES = ["quantiles_l1", "gaussian", "residual_magnitude", "residual_magnitude_l1", "inn"]


@pytest.mark.parametrize("utype", UTYPES)
def test_g12_final_layer_loss_and_gradients(utype):
    """the other final layers: heads + activation, train loss and its gradients vs the reference."""
    g = load_golden("g12_" + utype)
    st = om.det_state(1, 1, utype=utype)
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if k.startswith("last_layer.")}
    feat = T(g["feat"]).clone().requires_grad_(True)
    pred = om.final_layer(feat, leaves, utype)
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], rtol=1e-5, atol=1e-6)
    loss = om.uq_loss(pred, T(g["target"]), PARAMS, utype)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    np.testing.assert_allclose(feat.grad.numpy(), g["g_feat"], rtol=1e-4, atol=1e-7 * float(np.abs(g["g_feat"]).max()) + 1e-12)
    for k, v in leaves.items():
        ref = g["g_" + k[len("last_layer."):].replace(".", "_")]
        assert float(np.linalg.norm(v.grad.numpy() - ref) / (np.linalg.norm(ref) + 1e-30)) < 1e-4, k


@pytest.mark.parametrize("utype", UTYPES)
def test_g12_nested_sets_and_calibration(utype):
    g = load_golden("g12_" + utype)
    out = T(g["sets_output"])
    for i, lam in enumerate(T(g["lams"])):
        lo, _, hi = oc.nested_sets(out, lam, utype)
        assert np.array_equal(lo.numpy(), g["lower"][i]) and np.array_equal(hi.numpy(), g["upper"][i])
        rlo, _, rhi = oc.raw_nested_sets(out, lam, utype)
        assert np.array_equal(rlo.numpy(), g["raw_lower"][i]) and np.array_equal(rhi.numpy(), g["raw_upper"][i])
    v = g["cfg"]
    cfg = dict(alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]), minimum_lambda=float(v[3]), maximum_lambda=float(v[4]))
    lhat, table, _ = oc.calibrate_from_outputs(T(g["cal_output"]), T(g["cal_label"]), cfg, utype)
    assert np.array_equal(table.numpy(), g["table"]) and float(lhat) == float(g["lhat"])
    losses, spatial = oc.risk_and_miscoverage(T(g["cal_output"]), T(g["cal_label"]), T(g["lhat"]), utype)
    assert np.array_equal(losses.numpy(), g["risk"])
    np.testing.assert_allclose(spatial, g["spatial"], rtol=1e-6, atol=1e-7)


def test_g13_softmax_layer_loss_nested_sets_calibration():
    """softmax final layer (softmax_layer.py): class logits, cross entropy against the bucketised target, softmax-quantile
    nested sets and calibration, all vs the reference (same torch-CPU substrate -> exact where integers decide)."""
    g = load_golden("g13_softmax")
    st = om.det_state(1, 1, utype="softmax")
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if k.startswith("last_layer.")}
    feat = T(g["feat"]).clone().requires_grad_(True)
    pred = om.final_layer(feat, leaves, "softmax")
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], rtol=1e-5, atol=1e-6)
    loss = om.uq_loss(pred, T(g["target"]), PARAMS, "softmax")
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6)
    np.testing.assert_allclose(feat.grad.numpy(), g["g_feat"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(leaves["last_layer.output_layers.0.weight"].grad.numpy(), g["g_output_layers_0_weight"], rtol=1e-4, atol=1e-8)
    out = T(g["sets_output"])
    for i, lam in enumerate(T(g["lams"])):
        lo, mid, hi = oc.nested_sets(out, lam, "softmax")
        assert np.array_equal(lo.numpy(), g["lower"][i]) and np.array_equal(hi.numpy(), g["upper"][i])
        assert np.array_equal(mid.numpy(), g["prediction"][i])
        rlo, _, rhi = oc.raw_nested_sets(out, lam, "softmax")
        assert np.array_equal(rlo.numpy(), g["raw_lower"][i]) and np.array_equal(rhi.numpy(), g["raw_upper"][i])
    v = g["cfg"]
    cfg = dict(alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]), minimum_lambda_softmax=float(v[3]),
               maximum_lambda_softmax=float(v[4]))
    cout = T(g["cal_output"].astype(np.float32))
    lhat, table, _ = oc.calibrate_from_outputs(cout, T(g["cal_label"]), cfg, "softmax")
    assert np.array_equal(table.numpy(), g["table"]) and float(lhat) == float(g["lhat"])
    losses, spatial = oc.risk_and_miscoverage(cout, T(g["cal_label"]), T(g["lhat"]), "softmax")
    assert np.array_equal(losses.numpy(), g["risk"])


@pytest.mark.parametrize("name", ["doubleconv", "down", "up_bilinear", "up_bilinear_pad", "up_convT", "up_convT_pad"])
def test_g14_blocks_at_kernel_channel_counts(name):
    """the oracle's blocks against the reference's at the channel counts the GPU tests use (fixtures g14)."""
    g = load_golden("g14_" + name)
    prefix = {"doubleconv": "", "down": "maxpool_conv.1."}.get(name, "conv.")
    cin, mid, cout = {"doubleconv": (2, 32, 64), "down": (32, 64, 64)}.get(name, (128, 64, 64))
    st = {}
    if "convT" in name:                                       # Up(bilinear=False): ConvTranspose2d(128, 64, 2, 2)
        st["baseModel.blk.up.weight"] = om.det_fill(f"g14.{name}.up.weight", (128, 64, 2, 2))
        st["baseModel.blk.up.bias"] = om.det_fill(f"g14.{name}.up.bias", (64,))
    for idx, (ci, co) in ((0, (cin, mid)), (3, (mid, cout))):
        for leaf, shp in (("weight", (co, ci, 3, 3)), ("bias", (co,))):
            st[f"baseModel.blk.double_conv.{idx}.{leaf}"] = om.det_fill(f"g14.{name}.{prefix}double_conv.{idx}.{leaf}", shp)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            st[f"baseModel.blk.double_conv.{idx + 1}.{leaf}"] = om.det_fill(f"g14.{name}.{prefix}double_conv.{idx + 1}.{leaf}", (co,))
        st[f"baseModel.blk.double_conv.{idx + 1}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    for mode in ("eval", "train"):
        work = {k: v.clone() for k, v in st.items()}
        x0 = T(g[f"{mode}.x0"])
        if name == "doubleconv":
            y = om.double_conv(x0, work, "blk", mode == "train")
        elif name == "down":
            y = om.double_conv(torch.nn.functional.max_pool2d(x0, 2), work, "blk", mode == "train")
        else:
            y = om.up_block(x0, T(g[f"{mode}.x1"]), work, "blk", mode == "train")
        np.testing.assert_allclose(y.numpy(), g[f"{mode}.y"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("depth", [2, 5])
def test_g17_unet_of_other_depths(depth):
    """the oracle's depth-generic UNet against the reference's own DoubleConv/Down/Up/OutConv assembled to that depth
    (make_golden.py RefUNetDepth; BASELINE configs[0] is depth 2, configs[3] "deeper UNet" depth 5)."""
    g = load_golden(f"g17_unet_depth{depth}")
    st = om.det_state(1, 1, depth=depth)
    assert om.unet_depth(st) == depth
    x, y = T(g["x"]), T(g["y"])
    with torch.no_grad():
        out = om.model_forward(x, st, training=False)
    np.testing.assert_allclose(out.numpy(), g["out_eval"], rtol=1e-4, atol=1e-5)
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if om.is_param(k)}
    work = dict(st); work.update(leaves)
    pred = om.model_forward(x, work, training=True)
    np.testing.assert_allclose(pred.detach().numpy(), g["out_train"], rtol=1e-4, atol=2e-5)
    loss = om.quantile_loss(pred, y, PARAMS)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    for k in ("last_layer.prediction.weight", "baseModel.out.conv.weight", f"baseModel.up{depth}.conv.double_conv.3.weight",
              "baseModel.inc.double_conv.0.weight"):
        gr = leaves[k].grad.flatten()
        np.testing.assert_allclose(gr[::max(1, gr.numel() // 256)][:256].numpy(), g["gsample." + k], rtol=2e-3,
                                   atol=2e-3 * float(g["gnorm." + k]) / np.sqrt(gr.numel()))
    for k in g:
        if k.startswith("state."):
            np.testing.assert_allclose(work[k[len("state."):]].numpy(), g[k], rtol=1e-4, atol=1e-6)
    if depth == 2:
        st2 = om.det_state(1, 1, depth=2)
        losses = om.train_steps(st2, [(x, y)] * 5, PARAMS, lr=1e-3)
        np.testing.assert_allclose(losses, g["adam_losses"], rtol=2e-4)


def test_g15_evaluate_from_loss_table():
    """oracle and the C-ABI batch bound against the reference's evaluate_from_loss_table (4 re-splits incl. 'no lambda
    qualifies') and its per-lambda HB_mu_plus values on 0-dim fp32 risks."""
    g = load_golden("g15_evaluate_from_loss_table")
    table = T(g["table"])
    for (n, alpha, delta), want, hb in zip(g["cases"], g["values"], g["hb"]):
        n = int(n)
        torch.manual_seed(n)
        got = oc.evaluate_from_loss_table(table, n, alpha, delta)
        assert float(got) == pytest.approx(float(want), rel=1e-6, abs=1e-9)
        torch.manual_seed(n)
        rh = table[torch.randperm(table.shape[0])][:n].mean(dim=0)
        mine = np.array([oc.hb_mu_plus(np.float32(r.item()), n, delta) for r in rh], dtype=np.float64)
        np.testing.assert_allclose(mine, hb, rtol=0, atol=2e-7)        # the reference evaluates h1 on fp32 operands


def test_g15_batch_bound_through_the_c_abi():
    """im2im_hb_mu_plus_batch (host C++, no GPU needed) and the drop-in evaluate_from_loss_table built on it."""
    from im2im_uq_amd import hip_ops
    from im2im_uq_amd.core.calibration.calibrate_model import evaluate_from_loss_table
    g = load_golden("g15_evaluate_from_loss_table")
    table = T(g["table"])
    for (n, alpha, delta), want, hb in zip(g["cases"], g["values"], g["hb"]):
        n = int(n)
        torch.manual_seed(n)
        rh = table[torch.randperm(table.shape[0])][:n].mean(dim=0)
        got = hip_ops.hb_mu_plus_batch(rh, n, delta).numpy()
        np.testing.assert_allclose(got, hb, rtol=0, atol=2e-7)
        torch.manual_seed(n)
        val = evaluate_from_loss_table(table, n, alpha, delta)
        assert float(val) == pytest.approx(float(want), rel=1e-6, abs=1e-9)


def test_g16_wnet():
    g = load_golden("g16_wnet")
    keys = [str(k) for k in g["keys"]]
    from oracle.model import det_fill
    shapes = _wnet_shapes()
    assert [k for k in keys if k != "lhat"] == list(shapes)
    st = {k: det_fill(k, shp) for k, shp in shapes.items()}
    x, y = T(g["x"]), T(g["y"])
    with torch.no_grad():
        out = om.final_layer(om.wnet_forward(x, st, training=False), st)
    np.testing.assert_allclose(out.numpy(), g["out_eval"], rtol=1e-4, atol=1e-5)
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if om.is_param(k)}
    work = dict(st); work.update(leaves)
    pred = om.final_layer(om.wnet_forward(x, work, training=True), work)
    np.testing.assert_allclose(pred.detach().numpy(), g["out_train"], rtol=1e-4, atol=2e-5)
    loss = om.quantile_loss(pred, y, PARAMS)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    for k in ("baseModel.p1inc.double_conv.0.weight", "baseModel.p2down4.maxpool_conv.1.double_conv.3.weight", "baseModel.up1.conv.double_conv.0.weight"):
        gr = leaves[k].grad.flatten()
        np.testing.assert_allclose(gr[::max(1, gr.numel() // 256)][:256].numpy(), g["gsample." + k], rtol=2e-3,
                                   atol=2e-3 * float(g["gnorm." + k]) / np.sqrt(gr.numel()))


def _wnet_shapes():
    """state_dict (key -> shape) of add_uncertainty(WNet(1, 1), quantiles) in the reference's registration order
    (wnet.py:19-38; pinned against the fixture's recorded keys)."""
    shapes = {}

    def dc(prefix, cin, cmid, cout):
        p = f"baseModel.{prefix}.double_conv"
        for idx, (ci, co) in ((0, (cin, cmid)), (3, (cmid, cout))):
            shapes[f"{p}.{idx}.weight"] = (co, ci, 3, 3)
            shapes[f"{p}.{idx}.bias"] = (co,)
            for nm in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"{p}.{idx + 1}.{nm}"] = (co,)
            shapes[f"{p}.{idx + 1}.num_batches_tracked"] = ()
    for path in ("p1", "p2"):
        dc(path + "inc", 1, 32, 32)
        for i, (ci, co) in enumerate(((32, 64), (64, 128), (128, 256), (256, 256)), 1):
            dc(f"{path}down{i}.maxpool_conv.1", ci, co, co)
    for i, (ci, co) in enumerate(((1024, 256), (512, 128), (256, 64), (128, 64)), 1):
        dc(f"up{i}.conv", ci, ci // 2, co)
    shapes["baseModel.out.conv.weight"] = (32, 64, 1, 1)
    shapes["baseModel.out.conv.bias"] = (32,)
    for h in ("lower", "prediction", "upper"):
        shapes[f"last_layer.{h}.weight"] = (1, 32, 3, 3)
        shapes[f"last_layer.{h}.bias"] = (1,)
    return shapes


def test_g18_fastmri_pipeline_oracle():
    """oracle/fastmri.py against the reference's mask functions and UnetDataTransform (fixture G18)."""
    from oracle import fastmri as ofm
    g = load_golden("g18_fastmri_pipeline")
    for k in g:
        if k.startswith("mask.") and k != "mask.two_rates":
            _, kind, cols, fname = k.split(".", 3)
            m = ofm.seeded_mask(kind, int(cols), [0.08], [4], tuple(map(ord, fname)))
            assert np.array_equal(m.astype(np.uint8), g[k]), k
    two = np.stack([ofm.seeded_mask("equispaced", 368, [0.08, 0.04], [4, 8], (s,)) for s in range(6)]).astype(np.uint8)
    assert np.array_equal(two, g["mask.two_rates"])
    ks = T(g["small_kspace"])
    mask = T(ofm.seeded_mask("equispaced", 72, [0.08], [4], tuple(map(ord, "file1000001.h5"))))
    img = ofm.unet_data_transform(ks, mask, (48, 40))
    np.testing.assert_allclose(img.numpy(), g["small_image"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ofm.center_crop(T(g["small_target"]), (48, 40)).numpy(), g["small_target_out"], rtol=0, atol=0)
    img_f = ofm.unet_data_transform(ks, mask, (64, 80))
    assert img_f.shape == (72, 72)
    np.testing.assert_allclose(img_f.numpy(), g["small_image_flair"], rtol=1e-5, atol=1e-9)
    for cols in (368, 372):
        ks = ofm.det_kspace(1, 640, cols, salt=cols)[0]
        mask = T(ofm.seeded_mask("equispaced", cols, [0.08], [4], tuple(map(ord, "file1000277.h5"))))
        img = ofm.unet_data_transform(ks, mask, (320, 320))
        scale = float(g[f"full{cols}.max"])
        np.testing.assert_allclose(img[::4, ::4].numpy(), g[f"full{cols}.sample"], rtol=0, atol=2e-6 * scale)
        assert float(img.double().sum()) == pytest.approx(float(g[f"full{cols}.sum"]), rel=1e-6)
        assert float((img.double() ** 2).sum()) == pytest.approx(float(g[f"full{cols}.sumsq"]), rel=1e-6)
