"""Experiment entry point -- the reference's core/scripts/router.py (:30-182) sequence on the HIP path:
dataset -> UNet trunk -> add_uncertainty -> split -> train_net -> get_loss_table -> calibrate_model ->
loss_table_*.pth -> get_images -> eval_set_metrics -> results_*.pkl, with the reference's artefact names and keys.

The reference is driven by `wandb sweep`; offline this module expands the same sweep YAML itself:

    python -m im2im_uq_amd.core.scripts.router --config experiments/synthetic_fastmri/config.yml

Like the reference (`if torch.cuda.device_count() > 1: net = DataParallelPassthrough(net)`, train.py:112-115) the entry point
uses every GPU of the node by itself: with more than one GPU visible and no rendezvous environment it re-executes itself
under torch.distributed.run, one process per GPU over RCCL (`--gpus N` picks another count, `--gpus 1` stays single).
Under torchrun (`torchrun --nproc-per-node 8 -m im2im_uq_amd.core.scripts.router --config ...`) it is a plain rank.
"""
import argparse
import itertools
import os
import pickle as pkl
import random
import warnings

import numpy as np
import torch
import yaml

from .. import _pkg  # noqa: F401
from ... import nn_ops
from ..calibration.calibrate_model import calibrate_model
from ..models.add_uncertainty import add_uncertainty
from ..models.trunks.unet import UNet
from ._wandb import wandb
from .eval import eval_set_metrics, get_images, get_loss_table
from .train import train_net


def fix_randomness(seed=0):
    """core/utils.py:15-19"""
    np.random.seed(seed=seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    random.seed(seed)


def expand_sweep(doc):
    """wandb sweep YAML -> list of flat param dicts (grid over every `values:` list)."""
    if "parameters" not in doc:
        return [dict(doc)]                                     # already a flat config (tests/.../config.yml style)
    fixed, grid = {}, {}
    for k, v in doc["parameters"].items():
        if isinstance(v, dict) and "values" in v:
            grid[k] = v["values"]
        else:
            fixed[k] = v["value"] if isinstance(v, dict) and "value" in v else v
    runs = []
    for combo in itertools.product(*grid.values()) if grid else [()]:
        p = dict(fixed)
        p.update(dict(zip(grid.keys(), combo)))
        runs.append(p)
    return runs


def _suffix(params):
    return (params['dataset'] + "_" + params['uncertainty_type'] + "_" + str(params['batch_size']) + "_" + str(params['lr']) + "_"
            + params['input_normalization'] + "_" + params['output_normalization'].replace('.', '_'))


def build_dataset(params):
    if params["dataset"] == "synthetic":
        from ..datasets.synthetic import SyntheticDenoiseDataset
        return SyntheticDenoiseDataset(params.get("num_images", 64), params["num_inputs"], params.get("side_length", 320))
    raise NotImplementedError(
        f"dataset {params['dataset']!r}: the fastMRI / TEMCA / BSBCM loaders are outside this build's scope (SURVEY.md 2.1); "
        "pass any torch Dataset returning (input CxHxW, target CxHxW) to run_experiment(params, dataset=...)")


def run_experiment(params, dataset=None):
    fix_randomness()
    warnings.filterwarnings("ignore")
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    output_dir = params['output_dir']
    results_fname = output_dir + '/results_' + _suffix(params) + '.pkl'
    if os.path.exists(results_fname):
        print(f"Results already precomputed and stored in {results_fname}!")
        return None
    wandb.init(config=params)
    if "compute_dtype" in params:
        nn_ops.set_compute_dtype(params["compute_dtype"])
    if dataset is None:
        dataset = build_dataset(params)
    if params["model"] == "UNet":
        trunk = UNet(params["num_inputs"], 1)
    else:
        raise NotImplementedError
    model = add_uncertainty(trunk, params)

    lengths = np.round(len(dataset) * np.array(params["data_split_percentages"])).astype(int)
    lengths[-1] = len(dataset) - (lengths.sum() - lengths[-1])
    train_dataset, calib_dataset, val_dataset, _ = torch.utils.data.random_split(dataset, lengths.tolist())

    model = train_net(model, train_dataset, val_dataset, params['device'], params['epochs'], params['batch_size'], params['lr'],
                      params['load_from_checkpoint'], params['checkpoint_dir'], params['checkpoint_every'],
                      params['validate_every'], params)
    print("Done training!")
    model.eval()
    with torch.no_grad():
        val_loss_table = get_loss_table(model, val_dataset, params)
        model, calib_loss_table = calibrate_model(model, calib_dataset, params)
        print(f"Model calibrated! lambda hat = {model.lhat}")
        if rank == 0 and output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
            torch.save(torch.cat((calib_loss_table, val_loss_table), dim=0), output_dir + '/loss_table_' + _suffix(params) + '.pth')
        images = get_images(model, val_dataset, params['device'], list(range(params['num_validation_images'])), params)
        raw_images_dict = images[-1]
        risk, sizes, spearman, stratified_risk, mse, spatial_miscoverage = eval_set_metrics(model, val_dataset, params)
        print(f"Risk: {risk}  |  Mean size: {sizes.mean()}  |  Spearman: {spearman}  |  Size-stratified risk: {stratified_risk} | "
              f"MSE: {mse} | Spatial miscoverage: (mu, sigma, min, max) = ({spatial_miscoverage.mean()}, {spatial_miscoverage.std()}, "
              f"{spatial_miscoverage.min()}, {spatial_miscoverage.max()})")
        wandb.log({"epoch": params['epochs'] + 1, "risk": risk, "mean_size": sizes.mean(), "Spearman": spearman,
                   "Size-Stratified Risk": stratified_risk, "mse": mse, "spatial_miscoverage": spatial_miscoverage})
        results = {"risk": risk, "sizes": sizes, "spearman": spearman, "size-stratified risk": stratified_risk, "mse": mse,
                   "spatial_miscoverage": spatial_miscoverage}
        results.update({k: [t.cpu() if torch.is_tensor(t) else t for t in v] for k, v in raw_images_dict.items()})
        if rank == 0 and output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
            with open(results_fname, 'wb') as handle:
                pkl.dump(results, handle, protocol=pkl.HIGHEST_PROTOCOL)
            print(f'Results saved to file {results_fname}!')
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, help="wandb-sweep style YAML (experiments/*/config.yml) or a flat YAML")
    ap.add_argument("--set", nargs="*", default=[], help="overrides key=value (yaml-parsed)")
    ap.add_argument("--gpus", type=int, default=None, help="processes (one per GPU); default: every visible GPU, as the reference")
    args = ap.parse_args()
    with open(args.config) as f:
        doc = yaml.safe_load(f)
    from ... import launch
    if not launch.in_rendezvous_env():
        want = args.gpus if args.gpus is not None else torch.cuda.device_count()
        if want > 1:
            import sys
            sys.exit(launch.spawn_ranks(want, sys.argv[1:], module="im2im_uq_amd.core.scripts.router"))
    _dist, _rank, world, _dev, _backend = launch.init_distributed(expected_world=args.gpus if launch.in_rendezvous_env() and args.gpus else None)
    for params in expand_sweep(doc):
        for kv in args.set:
            k, v = kv.split("=", 1)
            params[k] = yaml.safe_load(v)
        if world > 1:
            params["device"] = str(_dev)
        run_experiment(params)


if __name__ == "__main__":
    main()
