"""Evaluation helpers -- drop-in for the reference's core/scripts/eval.py: get_loss_table (:86-127),
eval_set_metrics (:130-157), eval_net (:159-186), get_images (:24-84), transform_output (:14-22).

The model outputs of the evaluated set stay in HBM and the [N, num_lambdas] table comes from one pass of
the HIP scoring kernel (the reference loops lambdas x batches-of-4 with ~15 elementwise kernels each)."""
import numpy as np
import torch
from torch.utils.data import TensorDataset

from .. import _pkg  # noqa: F401
from ... import hip_ops
from ...prefetch import to_device
from ..calibration.calibrate_model import (collect_outputs, fraction_missed_loss, gather_rows, get_rcps_loss_fn,
                                           get_rcps_losses_from_outputs, get_rcps_metrics_from_outputs, lambda_grid, _dist)
from ..models.add_uncertainty import sets_form
from ._wandb import wandb


def transform_output(x, self_normalize=True):
    """tensor -> uint8 image (channels last when it has several): optional min-max stretch to [0,1], then 255*x clipped
    to [0,255] and truncated (reference :14-22)."""
    img = x.detach().to("cpu", torch.float32)
    if self_normalize:
        img = img - img.min()
        img = img / img.max()
    img = (255 * img.squeeze()).clamp(0, 255)
    if img.dim() == 3:
        img = img.permute(1, 2, 0)
    return img.numpy().astype(np.uint8)


def get_images(model, val_dataset, device, idx_iterator, config):
    """Example panels of a few validation images (reference :24-84): the calibrated (or lam = 1 / 0.99) nested sets of
    the selected images, as wandb images and, last, as the raw tensors that results_*.pkl stores (keys inputs / gt /
    predictions / lower_edge / upper_edge).  The selected images go through the network as ONE batch (eval-mode outputs
    do not depend on the batch composition); every returned list still has one [1,C,H,W] entry per image."""
    picks = list(idx_iterator)
    with torch.no_grad():
        model = model.to(device)
        lam = None
        if model.lhat is None:
            lam = 0.99 if config["uncertainty_type"] == "softmax" else 1.0
        if hasattr(val_dataset, "__getitem__"):
            samples = {i: val_dataset[i] for i in picks}
        else:                                                  # iterable dataset: the first len(picks) samples of the stream
            stream = iter(val_dataset)
            drawn = [next(stream) for _ in picks]
            samples = {i: drawn[i] for i in picks}
        x = torch.stack([samples[i][0] for i in picks]).to(device, torch.float32)
        lower, pred, upper = (t.float() for t in model.nested_sets((x,), lam=lam))
        per_image = [(lower[j:j + 1], pred[j:j + 1], upper[j:j + 1]) for j in range(len(picks))]
        gt = [samples[i][1] for i in picks]
        inputs = [samples[i][0][0] if samples[i][0].shape[0] > 1 else samples[i][0] for i in picks]
        raw_images_dict = {'inputs': inputs, 'gt': gt, 'predictions': [e[1] for e in per_image],
                           'lower_edge': [e[0] for e in per_image], 'upper_edge': [e[2] for e in per_image]}

        def panel(tensors, **kw):
            return [wandb.Image(transform_output(t, **kw)) for t in tensors]

        span = [e[1].max() - e[1].min() for e in per_image]   # interval lengths are shown on the prediction's own scale
        lower_len = panel([(e[1] - e[0]) / s for e, s in zip(per_image, span)], self_normalize=False)
        upper_len = panel([(e[2] - e[1]) / s for e, s in zip(per_image, span)], self_normalize=False)
        if hasattr(val_dataset, "reset"):
            try:
                val_dataset.reset()
            except Exception:  # noqa: BLE001
                pass
        return (panel(inputs), panel(raw_images_dict['lower_edge']), panel(raw_images_dict['predictions']),
                panel(raw_images_dict['upper_edge']), panel(gt), lower_len, upper_len, raw_images_dict)


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
        if outputs is None:                                   # empty shard of a multi-rank run (fewer images than ranks)
            table = torch.zeros((0, lambdas.numel()), dtype=torch.float32, device=device)
        elif rcps_loss_fn is fraction_missed_loss and form is not None:
            table = hip_ops.rcps_loss_table(outputs, labels, lambdas, form=form)
        else:
            ds = TensorDataset(outputs, labels)
            table = torch.stack([get_rcps_losses_from_outputs(model, ds, rcps_loss_fn, lam, device) for lam in lambdas], dim=1).to(device)
        return gather_rows(table).cpu()


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
        # with torch.distributed active every rank runs the forward over its own contiguous block of the set only; the
        # per-image rows are gathered and the [H,W] miss counts all-reduced inside get_rcps_metrics_from_outputs
        outputs, labels = collect_outputs(model, dataset, cfg, device, shard=True)
        losses, sizes, spearman, stratified_risks, mse, spatial_miscoverage = get_rcps_metrics_from_outputs(
            model, TensorDataset(outputs, labels) if outputs is not None else None, rcps_loss_fn, device, sharded=True)
        return losses.mean(), sizes, spearman, stratified_risks, mse, spatial_miscoverage


def eval_net(net, loader, device):
    with torch.no_grad():
        net.eval()
        net.to(device=device)
        val_loss = torch.zeros((), dtype=torch.float64, device=device)
        num_val = 0
        for batch in to_device(loader, device):              # [r6] uploads overlap the previous batch's kernels (im2im_uq_amd/prefetch.py)
            labels = batch[-1].to(device=device)
            x = [batch[i].to(device=device, dtype=torch.float32) for i in range(len(batch) - 1)]
            labels_pred = net(*x)
            num_val += labels.shape[0]
            val_loss += net.loss_fn(labels_pred, labels)
        net.train()
        if num_val == 0:
            return 0
        return val_loss.item() / num_val
