import os
import pickle

import pytest
import torch
import yaml

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_router_end_to_end_small(tmp_path):
    """the reference's router sequence (train -> loss table -> calibrate -> metrics -> artefacts) at 64x64."""
    from im2im_uq_amd.core.scripts.router import expand_sweep, run_experiment
    doc = yaml.safe_load(open(os.path.join(ROOT, "experiments", "synthetic_fastmri", "config.yml")))
    params = expand_sweep(doc)[0]
    params.update(output_dir=str(tmp_path / "out"), checkpoint_dir=str(tmp_path / "ckpt"), num_images=32, side_length=64,
                  epochs=2, batch_size=8, num_lambdas=100, num_validation_images=2)
    res = run_experiment(params)
    assert set(res) >= {"risk", "sizes", "spearman", "size-stratified risk", "mse", "spatial_miscoverage", "inputs", "gt",
                        "predictions", "lower_edge", "upper_edge"}
    assert res["spatial_miscoverage"].shape == (64, 64)
    names = os.listdir(tmp_path / "out")
    assert any(n.startswith("loss_table_synthetic_quantiles_8_0.001") for n in names)
    assert any(n.startswith("results_synthetic_quantiles_8_0.001") for n in names)
    table = torch.load(tmp_path / "out" / [n for n in names if n.startswith("loss_table")][0])
    assert table.shape == (16, 100)                                   # calib rows + val rows
    # checkpoints: whole-module pickles with the reference's naming, resumable
    ck = os.listdir(tmp_path / "ckpt")
    assert "CP_epoch2_synthetic_quantiles_8_0.001_standard_min-max.pth" in ck
    m = torch.load(tmp_path / "ckpt" / "CP_epoch2_synthetic_quantiles_8_0.001_standard_min-max.pth", weights_only=False)
    assert hasattr(m, "baseModel") and hasattr(m, "last_layer")
    with open(tmp_path / "out" / [n for n in names if n.startswith("results")][0], "rb") as f:
        assert "risk" in pickle.load(f)
    # second call: results exist -> skipped, like the reference (router.py:40-43)
    assert run_experiment(params) is None
