"""Host-side router logic that needs no GPU: sweep expansion of the reference-schema YAML, artefact naming."""
import os

import yaml

from conftest import ROOT


def test_expand_sweep_and_names():
    from im2im_uq_amd.core.scripts.router import _suffix, expand_sweep
    doc = yaml.safe_load(open(os.path.join(ROOT, "experiments", "synthetic_fastmri", "config.yml")))
    runs = expand_sweep(doc)
    assert len(runs) == 1 and runs[0]["uncertainty_type"] == "quantiles" and runs[0]["num_lambdas"] == 1000
    doc["parameters"]["lr"] = {"values": [0.001, 0.0001]}
    doc["parameters"]["uncertainty_type"] = {"values": ["gaussian", "quantiles"]}
    runs = expand_sweep(doc)
    assert len(runs) == 4
    assert _suffix(runs[0]) == "synthetic_gaussian_16_0.001_standard_min-max"


def test_out_of_scope_heads_raise_named_error():
    import pytest
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    with pytest.raises(NotImplementedError):
        add_uncertainty(UNet(1, 1), {"uncertainty_type": "nope"})


def test_final_layer_families_keep_the_reference_state_dict_keys():
    """all seven uncertainty types of the reference factory build, and their last_layer parameters carry the reference's names
    (finallayers/*.py __init__), so reference checkpoints load."""
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty, sets_form
    from im2im_uq_amd.core.models.trunks.unet import UNet
    params = dict(q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1, num_softmax=50, beta=0.1)
    m = add_uncertainty(UNet(1, 1), dict(params, uncertainty_type="softmax"))
    assert [k for k in m.state_dict() if k.startswith("last_layer.")] == ["last_layer.output_layers.0.weight", "last_layer.output_layers.0.bias"]
    assert tuple(m.last_layer.output_layers[0].weight.shape) == (50, 32, 3, 3) and sets_form(m) == 3
    expect = {"quantiles": ("lower", "prediction", "upper"), "quantiles_l1": ("lower", "prediction", "upper"),
              "gaussian": ("mean", "variance"), "residual_magnitude": ("prediction", "residual_magnitude"),
              "residual_magnitude_l1": ("prediction", "residual_magnitude"), "inn": ("lower", "prediction", "upper")}
    for utype, heads in expect.items():
        m = add_uncertainty(UNet(1, 1), dict(params, uncertainty_type=utype))
        keys = [k for k in m.state_dict() if k.startswith("last_layer.")]
        assert keys == [f"last_layer.{h}.{p}" for h in heads for p in ("weight", "bias")], (utype, keys)
        assert sets_form(m) in (0, 1, 2)


def test_reference_whole_module_checkpoint_unpickles_into_the_hip_classes(tmp_path):
    """train.py:191-192 pickles the whole ModelWithUncertainty; im2im_uq_amd.compat aliases the reference's module paths to
    this package's mirror while unpickling.  Needs the reference tree to write such a file (build container only)."""
    import os
    import subprocess
    import sys
    import pytest
    import torch
    ref = os.environ.get("IM2IM_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "core")):
        pytest.skip("reference tree not present (GPU box)")
    ckpt = str(tmp_path / "CP_epoch1_ref.pth")
    script = (
        "import sys, torch; sys.path.insert(0, %r); sys.dont_write_bytecode = True\n"
        "from core.models.trunks.unet import UNet\n"
        "from core.models.add_uncertainty import add_uncertainty\n"
        "torch.manual_seed(3)\n"
        "p = dict(uncertainty_type='gaussian', q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)\n"
        "m = add_uncertainty(UNet(1, 1), p); m.set_lhat(torch.tensor(1.25))\n"
        "torch.save(m, %r); torch.save(m.state_dict(), %r)\n" % (ref, ckpt, ckpt + ".sd"))
    subprocess.run([sys.executable, "-c", script], check=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    from im2im_uq_amd.compat import load_reference_checkpoint
    model = load_reference_checkpoint(ckpt)
    assert type(model).__module__ == "im2im_uq_amd.core.models.add_uncertainty"
    assert type(model.baseModel).__module__ == "im2im_uq_amd.core.models.trunks.unet"
    assert type(model.last_layer).__name__ == "GaussianRegressionLayer" and type(model.last_layer).__module__.startswith("im2im_uq_amd.")
    assert model.in_train_loss_fn.__module__.startswith("im2im_uq_amd.") and float(model.lhat) == 1.25
    sd = torch.load(ckpt + ".sd")
    got = model.state_dict()
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert "core" not in sys.modules or not getattr(sys.modules["core"], "__name__", "").startswith("im2im_uq_amd")


def test_wnet_state_dict_matches_the_reference(tmp_path):
    """same parameter names, order and shapes as the reference's WNet (needs the reference tree: build container only)."""
    import os
    import subprocess
    import sys
    import pytest
    import torch
    ref = os.environ.get("IM2IM_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "core")):
        pytest.skip("reference tree not present (GPU box)")
    out = str(tmp_path / "keys.pt")
    script = ("import sys, torch; sys.path.insert(0, %r); sys.dont_write_bytecode = True\n"
              "from core.models.trunks.wnet import WNet\n"
              "torch.save({k: tuple(v.shape) for k, v in WNet(2, 1).state_dict().items()}, %r)\n" % (ref, out))
    subprocess.run([sys.executable, "-c", script], check=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    from im2im_uq_amd.core.models.trunks.wnet import WNet
    want = torch.load(out)
    got = {k: tuple(v.shape) for k, v in WNet(2, 1).state_dict().items()}
    assert list(got.items()) == list(want.items())


def test_prefetcher_is_a_pass_through_without_a_gpu_device():
    """[r6] im2im_uq_amd/prefetch.py: on a non-CUDA device (and with IM2IM_PREFETCH=0) the wrapper yields the loader's own batches"""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from im2im_uq_amd.prefetch import DevicePrefetcher, _map_tensors
    ds = TensorDataset(torch.arange(10.0).view(10, 1), torch.arange(10))
    loader = DataLoader(ds, batch_size=4)
    got = list(DevicePrefetcher(loader, "cpu"))
    want = list(loader)
    assert len(got) == 3 and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, want))
    assert len(DevicePrefetcher(loader, "cpu")) == 3
    nested = ([torch.ones(2), {"k": torch.zeros(1)}], 78, None)
    out = _map_tensors(nested, lambda t: t + 1)
    assert out[1] == 78 and out[2] is None and float(out[0][0][0]) == 2.0 and float(out[0][1]["k"][0]) == 1.0


def test_sequential_slices_are_the_unshuffled_loaders_batches():
    """[r6] prefetch.sequential_slices: for a plain host TensorDataset (or a Subset of one over a contiguous range) the views it yields are
    exactly DataLoader(shuffle=False)'s batches (incl. the short last one); anything else is left to the DataLoader"""
    import torch
    from torch.utils.data import DataLoader, Subset, TensorDataset
    from im2im_uq_amd.prefetch import sequential_slices
    x, y = torch.arange(23.0).view(23, 1), torch.arange(23)
    ds = TensorDataset(x, y)
    for d in (ds, Subset(ds, range(5, 19)), Subset(ds, range(0, 23))):
        a, b = list(sequential_slices(d, 4)), list(DataLoader(d, batch_size=4))
        assert len(a) == len(b) and all(torch.equal(p[0], q[0]) and torch.equal(p[1], q[1]) for p, q in zip(a, b))
    assert sequential_slices(Subset(ds, [1, 2, 5]), 4) is None and sequential_slices(Subset(ds, range(0, 23, 2)), 4) is None

    class Mine(TensorDataset):
        def __getitem__(self, i):
            return tuple(t[i] * 2 for t in self.tensors)
    assert sequential_slices(Mine(x, y), 4) is None
