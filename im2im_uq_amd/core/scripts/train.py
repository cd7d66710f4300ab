"""Training driver -- drop-in for the reference's core/scripts/train.py (train_net :62-197,
run_validation :29-60).  Same signature, checkpoint naming and resume rules; what changes:

  * the forward/backward/optimizer arithmetic runs in the HIP kernels (UNet modules + FusedAdam);
  * multi-GPU is one process per GPU (torch.distributed, backend nccl == RCCL over xGMI) instead of the
    reference's single-process nn.DataParallel (:22-27,:112-115): the global batch is split across ranks,
    gradients are averaged with ONE flat all-reduce per step, BatchNorm uses per-rank batch statistics
    exactly like DataParallel replicas do;
  * the per-step `loss.item()` host sync (:155) is replaced by a device-side accumulator read once per
    epoch; the logged value `epoch_loss / num_examples` is the same quantity.
"""
import logging
import os

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .. import _pkg  # noqa: F401
from ... import nn_ops
from ...compat import load_reference_checkpoint
from ._wandb import wandb
from .eval import eval_net, get_images


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class DataParallelPassthrough(nn.Module):
    """Kept for API compatibility (reference :22-27).  With one process per GPU there is nothing to wrap:
    this is a transparent holder whose attribute lookups fall through to `.module`."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


def allreduce_gradients(params):
    """average gradients over ranks with one flat all-reduce (17.27 M fp32 = 69 MB for the reference model);
    afterwards each p.grad is a view into the flat buffer."""
    dist = _dist()
    if dist is None:
        return
    params = [p for p in params if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


def _ckpt_name(checkpoint_dir, epoch, config):
    return (checkpoint_dir + f'/CP_epoch{epoch}_' + config['dataset'] + "_" + config['uncertainty_type'] + "_"
            + str(config['batch_size']) + "_" + str(config['lr']) + "_" + config['input_normalization'] + "_"
            + config['output_normalization'].replace('.', '_') + '.pth')


def run_validation(net, val_loader, val_dataset, device, global_step, epoch, config):
    with torch.no_grad():
        net.eval()
        try:
            (examples_input, examples_lower_edge, examples_prediction, examples_upper_edge, examples_ground_truth,
             examples_ll, examples_ul, results_list) = get_images(net, val_dataset, device,
                                                                  list(range(config['num_validation_images'])), config)
            wandb.log({"epoch": epoch, "iter": global_step, "examples_input": examples_input})
            wandb.log({"epoch": epoch, "iter": global_step, "Lower edge": examples_lower_edge})
            wandb.log({"epoch": epoch, "iter": global_step, "Predictions": examples_prediction})
            wandb.log({"epoch": epoch, "iter": global_step, "Upper edge": examples_upper_edge})
            wandb.log({"epoch": epoch, "iter": global_step, "Ground truth": examples_ground_truth})
            wandb.log({"epoch": epoch, "iter": global_step, "Lower length": examples_ll})
            wandb.log({"epoch": epoch, "iter": global_step, "Upper length": examples_ul})
        except Exception:  # noqa: BLE001  (the reference swallows image-logging failures too, :56-57)
            print("Failed logging images.")
        val_loss = eval_net(net, val_loader, device)
        wandb.log({"epoch": epoch, "iter": global_step, "val_loss": val_loss})
        print(f"Val loss: {val_loss}")
    net.train()


def train_net(net, train_dataset, val_dataset, device, epochs, batch_size, lr, load_from_checkpoint, checkpoint_dir,
              checkpoint_every, validate_every, config=None):
    starting_epoch = 0
    if config is None:
        config = wandb.config
    if load_from_checkpoint:
        checkpoint_final_path = _ckpt_name(checkpoint_dir, epochs, config)
        if os.path.exists(checkpoint_final_path):
            try:
                net = load_reference_checkpoint(checkpoint_final_path)      # this package's or the reference's pickle
                net.eval()
                print(f"Model loaded from checkpoint {checkpoint_final_path}")
                return net
            except Exception:  # noqa: BLE001
                print(f"Final model cannot be loaded from checkpoint {checkpoint_final_path}. Training now, for {epochs} epochs.")
        else:
            print(f"Final model cannot be loaded from checkpoint {checkpoint_final_path}. Training now, for {epochs} epochs.")
            for e in reversed(range(epochs)):
                checkpoint_intermediate_path = _ckpt_name(checkpoint_dir, e, config)
                if os.path.exists(checkpoint_intermediate_path):
                    net = load_reference_checkpoint(checkpoint_intermediate_path)
                    starting_epoch = e
                    print(f"Starting from epoch {e}.")
                    break

    global_step = 0
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    local_bs = max(1, batch_size // world)                   # the global batch is split over ranks, as DataParallel does
    sampler = None
    try:
        if dist:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True, seed=0)
            train_loader = DataLoader(train_dataset, batch_size=local_bs, sampler=sampler, num_workers=0)
        else:
            train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=0)
    except Exception:  # noqa: BLE001  (iterable datasets cannot be shuffled, reference :105-106)
        train_loader = DataLoader(train_dataset, batch_size=local_bs, shuffle=False, num_workers=0)
    val_loader = DataLoader(val_dataset, batch_size=batch_size, shuffle=False, num_workers=0)

    net = net.to(device=device)
    optimizer = nn_ops.FusedAdam(net.parameters(), lr=lr)    # torch.optim.Adam defaults, reference :120
    if starting_epoch == 0:
        try:
            wandb.watch(net, log_freq=100)
        except Exception:  # noqa: BLE001
            wandb.init(config=config)
            wandb.watch(net, log_freq=100)

    params = [p for p in net.parameters() if p.requires_grad]
    print("Start Training!")
    for epoch in range(starting_epoch, epochs):
        net = net.to(device)
        net.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        print('epoch ' + str(epoch + 1) + '\n')
        epoch_loss = torch.zeros((), dtype=torch.float64, device=device)
        num_examples = 0
        for batch in train_loader:
            labels = batch[-1].to(device=device)
            x = tuple([batch[i].to(device=device, dtype=torch.float32) for i in range(len(batch) - 1)])

            labels_pred = net(*x)
            loss = net.loss_fn(labels_pred, labels)
            epoch_loss += loss.detach()

            optimizer.zero_grad()
            loss.backward()
            allreduce_gradients(params)
            optimizer.step()

            global_step += 1
            num_examples += labels.shape[0]

        wandb.log({"iter": global_step, "train_loss": epoch_loss.item() / max(num_examples, 1)})

        with torch.no_grad():
            if (epoch) % validate_every == 0:
                run_validation(net, val_loader, val_dataset, device, global_step, epoch, config)

            if (epoch + 1) % checkpoint_every == 0 and rank == 0:
                print('saving checkpoint')
                if checkpoint_dir is not None:
                    try:
                        os.makedirs(checkpoint_dir, exist_ok=True)
                        logging.info('Created checkpoint directory')
                    except OSError:
                        pass
                    checkpoint_fname = _ckpt_name(checkpoint_dir, epoch + 1, config)
                    # whole-module pickle like the reference (:191); works on one device too (the reference's
                    # `.module` only exists on its DataParallel wrapper)
                    torch.save(getattr(net, "module", net).cpu(), checkpoint_fname)
                    net = net.to(device)
                    logging.info(f'Checkpoint {epoch + 1} saved !')
        net.eval()
    return net
