"""Round-3 behaviours that need the GPU: the fixes for the round-2 advisor findings (NaN through the Gaussian head's ReLU,
tensor hooks on conv weights vs the weight-gradient stream, fastMRI volumes that carry their own mask) and the C-ABI
lambda scan driven by calibrate_model."""
import sys
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def test_head_relu_propagates_nan_like_torch():
    """finallayers/gaussian_layer.py:15-17 applies nn.ReLU to the variance head: torch's relu keeps NaN (a diverged head
    must surface as a NaN loss, not as var = 0)."""
    from im2im_uq_amd._lib import check, dptr, lib, stream_ptr
    b, k, p = 2, 2, 96
    out = torch.randn(b, k, p, device=DEV)
    out[0, 1, 5] = float("nan")
    out[1, 1, 7] = -3.0
    want = out.clone()
    want[:, 1] = torch.relu(want[:, 1])
    pre = torch.empty(b, p, device=DEV)
    check(lib.im2im_head_activation_fwd(dptr(out), dptr(pre), b, p, k * p, p, 0, stream_ptr(out.device)), "im2im_head_activation_fwd")
    torch.cuda.synchronize()
    assert torch.isnan(out[0, 1, 5]) and float(out[1, 1, 7]) == 0.0
    assert torch.equal(torch.nan_to_num(out, nan=123.0), torch.nan_to_num(want, nan=123.0))


def test_tensor_hook_on_conv_weight_sees_the_finished_weight_gradient():
    """a hook registered on a conv weight (wandb.watch(net) does that, reference train.py:122) reads dW the moment autograd
    hands it over on the main stream; the weight gradient must then not still be in flight on the side stream."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.trunks.unet_parts import DoubleConv
    nn_ops.set_compute_dtype("fp32")
    try:
        torch.manual_seed(0)
        blk = DoubleConv(64, 64).to(DEV).train()
        x = torch.randn(4, 64, 96, 96, device=DEV)

        def run(with_hook):
            seen = {}
            handles = []
            if with_hook:
                for name, p in blk.named_parameters():
                    if p.dim() == 4:
                        handles.append(p.register_hook(lambda g, name=name: seen.__setitem__(name, g.detach().clone())))
            for p in blk.parameters():
                p.grad = None
            blk(x).square().mean().backward()
            nn_ops.join_side_streams()
            torch.cuda.synchronize()
            for h in handles:
                h.remove()
            return seen, {n: p.grad.clone() for n, p in blk.named_parameters() if p.dim() == 4}
        _, ref = run(False)
        for _ in range(3):                                    # a race would not show every time
            seen, got = run(True)
            assert set(seen) == set(ref)
            for n in ref:
                assert torch.equal(seen[n], ref[n]) and torch.equal(got[n], ref[n]), n
    finally:
        nn_ops.set_compute_dtype("bf16")


class _FakeH5File(dict):
    """the slice of h5py.File the dataset uses: item access, `in`, .attrs, context manager."""

    def __init__(self, items, attrs):
        super().__init__(items)
        self.attrs = attrs

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_HEADER = """<ismrmrdHeader xmlns="http://www.ismrm.org/ISMRMRD"><encoding>
<encodedSpace><matrixSize><x>96</x><y>72</y><z>1</z></matrixSize></encodedSpace>
<reconSpace><matrixSize><x>48</x><y>40</y><z>1</z></matrixSize></reconSpace>
<encodingLimits><kspace_encoding_step_1><minimum>0</minimum><maximum>71</maximum><center>36</center></kspace_encoding_step_1></encodingLimits>
</encoding></ismrmrdHeader>"""


def test_fastmri_dataset_reads_volumes_and_honours_a_mask_stored_in_the_file(tmp_path, monkeypatch):
    """FastMRIDataset against an in-memory stand-in for h5py (the image has none): train volumes get a fresh mask from
    mask_func; test / challenge volumes are already sub-sampled, carry `mask`, and must NOT be masked a second time
    (reference FastMRIDataset.py:136 passes hf['mask'] to the transform, transforms.py:286-292 then skips mask_func)."""
    from oracle import fastmri as ofm
    ks = ofm.det_kspace(4, 96, 72, salt=3)
    kc = (ks[..., 0] + 1j * ks[..., 1]).numpy()
    stored_mask = ofm.seeded_mask("random", 72, [0.1], [3], 7)
    pre_masked = kc * stored_mask.reshape(1, 1, -1)
    recon = ofm.unet_data_transform(ks, torch.ones(72), (48, 40)).numpy()
    files = {
        "train_vol.h5": _FakeH5File({"kspace": kc, "reconstruction_esc": recon, "ismrmrd_header": np.array(_HEADER.encode())}, {"max": 1.0}),
        "test_vol.h5": _FakeH5File({"kspace": pre_masked, "mask": stored_mask, "reconstruction_esc": recon,
                                    "ismrmrd_header": np.array(_HEADER.encode())}, {"max": 1.0}),
    }
    for name in files:
        (tmp_path / name).write_bytes(b"")
    fake = types.ModuleType("h5py")
    fake.File = lambda fname, mode="r": files[str(fname).rsplit("/", 1)[-1]]
    monkeypatch.setitem(sys.modules, "h5py", fake)
    from im2im_uq_amd.core.datasets.fastmri.FastMRIDataset import FastMRIDataset
    ds = FastMRIDataset(str(tmp_path), "standard", "min-max", {"type": "equispaced", "center_fraction": [0.08], "acceleration": [4]},
                        slice_sample_period=1, device=DEV)
    assert len(ds) == 8
    n_test = 0
    for i in range(len(ds)):
        fname, sl, _ = ds.examples[i]
        x, y = ds[i]
        assert x.shape == (1, 48, 40) and y.shape == (1, 48, 40)
        np.testing.assert_allclose(y[0].cpu().numpy(), recon[sl], rtol=0, atol=1e-6)
        if fname.name == "test_vol.h5":
            n_test += 1
            want = ofm.unet_data_transform(T(np.stack([pre_masked[sl].real, pre_masked[sl].imag], -1).astype(np.float32)),
                                           torch.ones(72), (48, 40))
            assert float((x[0].cpu() - want).abs().max()) <= 2e-5 * float(want.max())
    assert n_test == 4


@pytest.mark.parametrize("case", ["mid", "no_stop"])
def test_calibrate_model_runs_the_c_abi_scan(case, monkeypatch):
    """calibrate_model -> scan_loss_table -> im2im_rcps_scan: the lambda-hat the Python driver sets is the C library's, and
    equals the reference's (fixture G7)."""
    import torch.nn as nn
    from torch.utils.data import TensorDataset
    from im2im_uq_amd import hip_ops
    from im2im_uq_amd.core.calibration.calibrate_model import calibrate_model
    from im2im_uq_amd.core.models.add_uncertainty import ModelWithUncertainty
    from im2im_uq_amd.core.models.finallayers.quantile_layer import quantile_regression_nested_sets_from_output
    g = load_golden("g7_calibrate_" + case)
    v = g["cfg"]
    cfg = dict(alpha=float(v[0]), delta=float(v[1]), num_lambdas=int(v[2]), minimum_lambda=float(v[3]), maximum_lambda=float(v[4]),
               batch_size=int(v[5]), rcps_loss="fraction_missed", device=DEV, uncertainty_type="quantiles", dataset="synthetic")
    calls = []
    real = hip_ops.rcps_scan
    monkeypatch.setattr(hip_ops, "rcps_scan", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ident = ModelWithUncertainty(nn.Identity(), nn.Identity(), None, quantile_regression_nested_sets_from_output, dict(cfg))
    ident, table = calibrate_model(ident, TensorDataset(T(g["output"]).clone(), T(g["label"]).clone()), cfg)
    assert calls == [1]
    assert np.array_equal(table.numpy(), g["table"]) and np.float32(float(ident.lhat)) == np.float32(g["lhat"])


def test_graphed_train_step_is_bit_identical_to_the_eager_loop():
    """core/scripts/train.py GraphedStep (HIP graph of forward + loss + backward for launch-bound batch shapes, BASELINE
    configs[0]: 32x32, depth 2): the same batches through the eager loop (train.py:141-165 of the reference) and through the
    graphed one -- three eager warm-up steps, capture, replays, one short batch of another shape in between (eager fallback),
    the parameters moved to new memory once (re-capture) -- give bit-identical losses, parameters and BatchNorm buffers."""
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import GraphedStep
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)
    nn_ops.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randn(16 if i != 5 else 7, 1, 32, 32, generator=g), torch.rand(16 if i != 5 else 7, 1, 32, 32, generator=g)) for i in range(13)]

    def run(graph):
        torch.manual_seed(3)
        model = add_uncertainty(UNet(1, 1, depth=2), dict(params)).to(DEV).train()
        opt = nn_ops.FusedAdam(model.parameters(), lr=1e-3)
        gs = GraphedStep(model, opt) if graph else None
        losses = []
        for i, (x, y) in enumerate(batches):
            if i == 7:                                        # what train_net's checkpoint does: the parameters move to new memory
                model.cpu()
                model.to(DEV)
            x, y = x.to(DEV), y.to(DEV)
            loss = gs.step((x,), y) if gs else None
            if loss is None:
                loss = model.loss_fn(model(x), y)
                opt.zero_grad()
                loss.backward()
                opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        if graph:
            assert gs.graph is not None                       # the step was captured and replayed
        return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    le, se = run(False)
    lg, sg = run(True)
    assert torch.equal(le, lg), (le, lg)
    for k in se:
        assert torch.equal(se[k], sg[k]), k
    assert GraphedStep.wanted({}, 16 * 32 * 32, 1, "bf16") and not GraphedStep.wanted({}, 10 * 320 * 320, 1, "bf16")
    # [r4] several ranks: only when asked for (the step is then forward + backward + bucket packing in the graph, the exchange
    # after the replay); auto stays eager
    assert GraphedStep.wanted({"hip_graph": True}, 1024, 2, "bf16") and not GraphedStep.wanted({}, 1024, 2, "bf16")
    assert not GraphedStep.wanted({"hip_graph": False}, 1024, 1, "bf16") and not GraphedStep.wanted({"hip_graph": True}, 1024, 1, "fp8")


def test_train_net_with_and_without_the_hip_graph_gives_the_same_model(tmp_path, monkeypatch):
    """train_net (core/scripts/train.py, the reference's train loop :141-165) on a 32x32 dataset: IM2IM_HIP_GRAPH=0 (eager loop)
    and =1 (GraphedStep: warm-up, capture, replays, the short last batch of every epoch eager, re-capture after the per-epoch
    checkpoint moved the parameters) end in bit-identical state_dicts."""
    from torch.utils.data import TensorDataset
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    from im2im_uq_amd.core.scripts.train import train_net
    params = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1, alpha=0.1, delta=0.1,
                  num_lambdas=50, rcps_loss="fraction_missed", minimum_lambda=0, maximum_lambda=6, device=DEV, dataset="synthetic",
                  batch_size=8, lr=1e-3, input_normalization="standard", output_normalization="min-max", num_validation_images=2)
    nn_ops.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(5)
    train = TensorDataset(torch.randn(44, 1, 32, 32, generator=g), torch.rand(44, 1, 32, 32, generator=g))     # 5 full batches + one of 4
    val = TensorDataset(torch.randn(8, 1, 32, 32, generator=g), torch.rand(8, 1, 32, 32, generator=g))
    states = []
    for flag in ("0", "1"):
        monkeypatch.setenv("IM2IM_HIP_GRAPH", flag)
        torch.manual_seed(9)
        model = add_uncertainty(UNet(1, 1, depth=2), dict(params))
        model = train_net(model, train, val, DEV, epochs=3, batch_size=8, lr=1e-3, load_from_checkpoint=False,
                          checkpoint_dir=str(tmp_path / f"ck{flag}"), checkpoint_every=1, validate_every=10, config=dict(params))
        torch.cuda.synchronize()
        states.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k
