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


def test_router_spawns_its_own_ranks_and_matches_the_single_process_lambda(tmp_path):
    """`python -m im2im_uq_amd.core.scripts.router --gpus 2` with no launcher around it (the reference needs none either,
    train.py:112-115): the entry point re-executes itself under torch.distributed.run, the two ranks (sharing the test box's one
    GPU over gloo) train on split batches, shard the calibration / validation sets, and rank 0 writes the reference's artefacts --
    the same files, table shape and keys as the single-process run."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["IM2IM_DIST_BACKEND"] = "gloo"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    over = [f"output_dir={tmp_path / 'out'}", f"checkpoint_dir={tmp_path / 'ckpt'}", "num_images=32", "side_length=64", "epochs=1",
            "batch_size=8", "num_lambdas=100", "num_validation_images=2"]
    cmd = [sys.executable, "-m", "im2im_uq_amd.core.scripts.router", "--config",
           os.path.join(ROOT, "experiments", "synthetic_fastmri", "config.yml"), "--gpus", "2", "--set"] + over
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    names = os.listdir(tmp_path / "out")
    table = torch.load(tmp_path / "out" / [n for n in names if n.startswith("loss_table")][0])
    assert table.shape == (16, 100)
    with open(tmp_path / "out" / [n for n in names if n.startswith("results")][0], "rb") as f:
        res = pickle.load(f)
    assert res["spatial_miscoverage"].shape == (64, 64) and 0.0 <= float(res["risk"]) <= 1.0
    assert "CP_epoch1_synthetic_quantiles_8_0.001_standard_min-max.pth" in os.listdir(tmp_path / "ckpt")
    assert r.stdout.count("Model calibrated!") == 2                          # both ranks ran the identical scan
    lhats = {line.split("lambda hat = ")[1].strip() for line in r.stdout.splitlines() if "lambda hat = " in line}
    assert len(lhats) == 1, lhats                                            # ... and landed on the identical lambda-hat
