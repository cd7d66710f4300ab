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
    with pytest.raises(NotImplementedError, match="gaussian"):
        add_uncertainty(UNet(1, 1), {"uncertainty_type": "gaussian"})
    with pytest.raises(NotImplementedError):
        add_uncertainty(UNet(1, 1), {"uncertainty_type": "nope"})
