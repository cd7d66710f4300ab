"""Evaluation helpers -- drop-in for the reference's core/scripts/eval.py: get_loss_table (:86-127),
eval_set_metrics (:130-157), eval_net (:159-186), get_images (:24-84), transform_output (:14-22).

The model outputs of the evaluated set stay in HBM and the [N, num_lambdas] table comes from one pass of
the HIP scoring kernel (the reference loops lambdas x batches-of-4 with ~15 elementwise kernels each)."""
import numpy as np
import torch
from torch.utils.data import TensorDataset

from .. import _pkg  # noqa: F401
from ... import hip_ops
from ..calibration.calibrate_model import (collect_outputs, fraction_missed_loss, gather_rows, get_rcps_loss_fn,
                                           get_rcps_losses_from_outputs, get_rcps_metrics_from_outputs, lambda_grid, _dist)
from ..models.add_uncertainty import sets_form
from ._wandb import wandb


def transform_output(x, self_normalize=True):
    if self_normalize:
        x = x - x.min()
        x = x / x.max()
    x = np.maximum(0, np.minimum(255 * x.cpu().squeeze(), 255))
    if len(x.shape) == 3:
        x = x.permute(1, 2, 0)
    return x.numpy().astype(np.uint8)


def get_images(model, val_dataset, device, idx_iterator, config):
    with torch.no_grad():
        model = model.to(device)
        lam = None
        if model.lhat == None:
            lam = 1.0 if config["uncertainty_type"] != "softmax" else 0.99
        try:
            my_iter = iter(val_dataset)
            val_dataset = [next(my_iter) for img_idx in idx_iterator]
        except Exception:  # noqa: BLE001
            pass
        examples_output = [tuple(t.float() for t in model.nested_sets((val_dataset[img_idx][0].unsqueeze(0).to(device, torch.float32),), lam=lam))
                           for img_idx in idx_iterator]
        examples_gt = [val_dataset[img_idx][1] for img_idx in idx_iterator]
        if val_dataset[0][0].shape[0] > 1:
            inputs = [val_dataset[img_idx][0][0] for img_idx in idx_iterator]
        else:
            inputs = [val_dataset[img_idx][0] for img_idx in idx_iterator]
        raw_images_dict = {'inputs': inputs, 'gt': examples_gt,
                           'predictions': [example[1] for example in examples_output],
                           'lower_edge': [example[0] for example in examples_output],
                           'upper_edge': [example[2] for example in examples_output]}
        examples_input = [wandb.Image(transform_output(i)) for i in inputs]
        examples_lower_edge = [wandb.Image(transform_output(example[0])) for example in examples_output]
        examples_prediction = [wandb.Image(transform_output(example[1])) for example in examples_output]
        examples_upper_edge = [wandb.Image(transform_output(example[2])) for example in examples_output]
        examples_ground_truth = [wandb.Image(transform_output(val_dataset[img_idx][1])) for img_idx in idx_iterator]
        spans = [(e[1].max() - e[1].min()) for e in examples_output]
        lower_lengths = [transform_output((e[1] - e[0]) / s, self_normalize=False) for e, s in zip(examples_output, spans)]
        upper_lengths = [transform_output((e[2] - e[1]) / s, self_normalize=False) for e, s in zip(examples_output, spans)]
        examples_lower_length = [wandb.Image(ll) for ll in lower_lengths]
        examples_upper_length = [wandb.Image(ul) for ul in upper_lengths]
        try:
            val_dataset.reset()
        except Exception:  # noqa: BLE001
            pass
        return (examples_input, examples_lower_edge, examples_prediction, examples_upper_edge, examples_ground_truth,
                examples_lower_length, examples_upper_length, raw_images_dict)


def _outputs_for(model, dataset, config, device):
    try:
        dataset.reset()
    except Exception:  # noqa: BLE001
        pass
    cfg = dict(config)
    cfg.setdefault('batch_size', 64)
    return collect_outputs(model, dataset, cfg, device)


def get_loss_table(model, dataset, config):
    """[N, num_lambdas] loss of every image at every (un-shifted) lambda."""
    with torch.no_grad():
        lambdas = lambda_grid(config)
        model.eval()
        device = config['device']
        rcps_loss_fn = get_rcps_loss_fn(config)
        model = model.to(device)
        outputs, labels = _outputs_for(model, dataset, config, device)
        form = sets_form(model)
        if rcps_loss_fn is fraction_missed_loss and form is not None:
            table = hip_ops.rcps_loss_table(outputs, labels, lambdas, form=form)
        else:
            ds = TensorDataset(outputs, labels)
            table = torch.stack([get_rcps_losses_from_outputs(model, ds, rcps_loss_fn, lam, device) for lam in lambdas], dim=1).to(device)
        if _dist() is not None:
            cnt = torch.tensor([table.shape[0]], device=table.device)
            _dist().all_reduce(cnt)
            table = gather_rows(table, int(cnt.item()))
        return table.cpu()


def eval_set_metrics(model, dataset, config):
    """(risk, sizes, spearman, stratified_risks, mse, spatial_miscoverage) at the calibrated lhat."""
    with torch.no_grad():
        model.eval()
        device = config['device']
        rcps_loss_fn = get_rcps_loss_fn(config)
        model = model.to(device)
        cfg = dict(config)
        cfg.setdefault('batch_size', 64)
        try:
            dataset.reset()
        except Exception:  # noqa: BLE001
            pass
        outputs, labels = collect_outputs(model, dataset, cfg, device, shard=False)
        losses, sizes, spearman, stratified_risks, mse, spatial_miscoverage = get_rcps_metrics_from_outputs(
            model, TensorDataset(outputs, labels), rcps_loss_fn, device)
        return losses.mean(), sizes, spearman, stratified_risks, mse, spatial_miscoverage


def eval_net(net, loader, device):
    with torch.no_grad():
        net.eval()
        net.to(device=device)
        val_loss = torch.zeros((), dtype=torch.float64, device=device)
        num_val = 0
        for batch in loader:
            labels = batch[-1].to(device=device)
            x = [batch[i].to(device=device, dtype=torch.float32) for i in range(len(batch) - 1)]
            labels_pred = net(*x)
            num_val += labels.shape[0]
            val_loss += net.loss_fn(labels_pred, labels)
        net.train()
        if num_val == 0:
            return 0
        return val_loss.item() / num_val
