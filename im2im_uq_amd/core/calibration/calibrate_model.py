"""RCPS calibration on MI355X -- drop-in for the reference's core/calibration/calibrate_model.py.

Same public functions and return conventions as the reference; what changes is where the
arithmetic runs:

  * model outputs of the calibration set stay resident in HBM (the reference round-trips them
    through host memory and re-uploads 16 B/pixel for every lambda, calibrate_model.py:111-123,
    :21-29);
  * the whole [N, num_lambdas] loss table comes from ONE pass of the HIP scoring kernel
    (csrc/rcps.hip) instead of one elementwise pass per lambda;
  * the descending Hoeffding-Bentkus scan (calibrate_model.py:134-144) then runs on the host over
    that table, with the reference's quirks kept: the `lam - dlambda` shift, zero columns left of
    the break, `Rhat >= alpha or RhatPlus > alpha`, default lhat = last + dlambda - 1e-9;
  * with torch.distributed initialised the calibration set is sharded by contiguous index range
    per rank and the table rows are all-gathered (RCCL) so every rank runs the identical scan and
    lands on the identical lhat.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import spearmanr
from torch.utils.data import DataLoader, Subset, TensorDataset

from .. import _pkg  # noqa: F401
from ... import hip_ops
from ...prefetch import loader as prefetch_loader, sequential_slices, to_device
from ..models.add_uncertainty import calibration_repr, sets_form
from .bounds import HB_mu_plus  # noqa: F401  (re-exported: the reference's module exposes it here too)


# ------------------------------------------------------------------ distributed helpers
def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def shard_bounds(n: int, rank: int, world: int):
    """contiguous index range of `rank` (row order of the gathered table == dataset order)."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def gather_rows(local: torch.Tensor, n_total=None) -> torch.Tensor:
    """all-gather [n_local, ...] row blocks -> [sum n_local, ...] in rank order.  The ranks first exchange their row
    counts, so any partition works (contiguous `shard_bounds` shards, caller-provided local shards of unequal size,
    empty shards); `n_total`, when given, is checked against the gathered count."""
    dist = _dist()
    if dist is None:
        return local
    world = dist.get_world_size()
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    if n_total is not None and sum(counts) != n_total:
        raise ValueError(f"gather_rows: ranks hold {counts} rows, expected {n_total} in total")
    per = max(max(counts), 1)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: counts[r]] for r in range(world)], dim=0)


# ------------------------------------------------------------------ losses
def fraction_missed_loss(pset, label):
    """Per-image fraction of pixels outside [lower, upper] (reference :76-80).  pset = (lower, pred, upper).
    A batch of one image returns shape [1] (the reference's squeeze returns [H] there)."""
    lower, upper = pset[0], pset[2]
    return hip_ops.fraction_missed(lower, upper, label.to(lower.device, torch.float32))


def get_rcps_loss_fn(config):
    string = config['rcps_loss']
    if string == 'fraction_missed':
        return fraction_missed_loss
    else:
        raise NotImplementedError


def _as_device_pair(out_dataset, device, model=None):
    outputs, labels = out_dataset.tensors
    outputs = outputs.to(device=device, dtype=torch.float32).contiguous()
    if model is not None:
        outputs = calibration_repr(model, outputs)            # softmax layer: class logits -> quantile summary planes
    return outputs, labels.to(device=device, dtype=torch.float32).contiguous()


def get_rcps_losses_from_outputs(model, out_dataset, rcps_loss_fn, lam, device):
    """Losses [N] (cpu) of every image at ONE lambda (reference :21-29)."""
    outputs, labels = _as_device_pair(out_dataset, device, model)
    form = sets_form(model)
    if rcps_loss_fn is fraction_missed_loss and form is not None:
        lam_t = torch.as_tensor(lam, dtype=torch.float32).reshape(1)
        return hip_ops.rcps_loss_table(outputs, labels, lam_t, form=form)[:, 0].cpu()
    model = model.to(device)
    losses = []
    for s in range(0, outputs.shape[0], 64):          # user-supplied loss: same batching as the reference
        sets = model.nested_sets_from_output(outputs[s:s + 64].clone(), lam)
        losses.append(rcps_loss_fn(sets, labels[s:s + 64]).cpu())
    return torch.cat(losses, dim=0)


def get_rcps_losses(model, dataset, rcps_loss_fn, lam, device):
    """The reference's version of this (:13-19) is dead code referencing an undefined name; this one
    runs the model over (input, label) pairs and scores them at `lam`."""
    outputs, labels = collect_outputs(model, dataset, {'batch_size': 64, 'dataset': None}, device, shard=False)
    return get_rcps_losses_from_outputs(model, TensorDataset(outputs, labels), rcps_loss_fn, lam, device)


def shard_offset(n_local: int, device):
    """(first global row of this rank's block, total rows) for row blocks held in rank order."""
    dist = _dist()
    if dist is None:
        return 0, n_local
    cnt = torch.tensor([n_local], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    return sum(counts[: dist.get_rank()]), sum(counts)


def get_rcps_metrics_from_outputs(model, out_dataset, rcps_loss_fn, device, sharded=False):
    """Metrics at the model's calibrated lhat (reference :31-60).  Returns
    (losses[N], sizes[N], spearman, stratified_risks[4], mse, spatial_miscoverage[H,W]).
    numpy/torch RNG is consumed in the reference's order (one np.random.choice per batch of 64, then one
    torch.rand), so with the same seed the sampled-pixel statistics are the reference's.

    sharded=True (torch.distributed active): `out_dataset` holds only this rank's contiguous block of the evaluated set
    (or None for an empty block).  Each rank scores its own images; the per-image rows are all-gathered in rank order
    and the int32 per-pixel miss counts [C,H*W] are all-reduced, after which every rank evaluates the
    same scalars.  The random pixels are drawn for the WHOLE set on every rank (same seed => same draws as one GPU)."""
    dist = _dist() if sharded else None
    model = model.to(device)
    if model.lhat is None:
        raise Exception("You have to specify lambda unless your model is already calibrated.")
    lhat = float(model.lhat)
    form = sets_form(model)
    if form is None:
        raise NotImplementedError("get_rcps_metrics_from_outputs needs one of this package's nested-set functions")
    if out_dataset is not None:
        outputs, labels = _as_device_pair(out_dataset, device, model)
        n_loc = outputs.shape[0]
        k, c = outputs.shape[1], outputs.shape[2]
        p = outputs[0, 0].numel()
        hw_shape = tuple(outputs.shape[3:])
    else:
        outputs = labels = None
        n_loc = k = c = p = 0
        hw_shape = ()
    lo_row, n = shard_offset(n_loc, device) if dist is not None else (0, n_loc)
    if dist is not None:                                      # an empty rank learns the image geometry from the others
        geo = torch.tensor([c, p] + list(hw_shape) + [0] * (4 - len(hw_shape)), dtype=torch.int64, device=device)
        dist.all_reduce(geo, op=dist.ReduceOp.MAX)
        c, p = int(geo[0]), int(geo[1])
        hw_shape = tuple(int(v) for v in geo[2:] if int(v) > 0)
    # one random pixel per image: interval size and |residual| (reference :43-46)
    idx = np.concatenate([np.random.choice(p, size=min(64, n - s)) for s in range(0, n, 64)]) if n else np.zeros(0, int)
    if n_loc:
        losses = hip_ops.rcps_loss_table(outputs, labels, torch.tensor([lhat], dtype=torch.float32), form=form)[:, 0]
        idx_t = torch.from_numpy(idx[lo_row:lo_row + n_loc]).to(device)
        rows = torch.arange(n_loc, device=device)
        picked = outputs.flatten(start_dim=2)[rows, :, idx_t].reshape(n_loc, k, 1).contiguous()        # [N,K,1]
        lo, mid, up = hip_ops.nested_sets(picked, lhat, form=form)
        sizes = (up - lo).reshape(n_loc)
        residuals = (labels.flatten(start_dim=1)[rows, idx_t] - mid.reshape(n_loc)).abs()
        counts = hip_ops.rcps_miscoverage(outputs, labels, lhat, form=form)
    else:
        losses = sizes = residuals = torch.zeros((0,), dtype=torch.float32, device=device)
        counts = torch.zeros((c, p), dtype=torch.int32, device=device)
    if dist is not None:
        rows3 = gather_rows(torch.stack([losses, sizes, residuals], dim=1), n)
        losses, sizes, residuals = rows3[:, 0], rows3[:, 1], rows3[:, 2]
        dist.all_reduce(counts)                               # integer per-pixel miss counts: exact in any order
    sizes = sizes.cpu()
    # the reference iterates one DataLoader here, whose iterator draws its base seed from torch's default
    # generator (one int64 random_()); consume the same draw so the jitter below is the reference's.
    torch.empty((), dtype=torch.int64).random_()
    sizes = sizes + torch.rand(size=sizes.shape).to(sizes.device) * 1e-6
    residuals = residuals.detach().cpu().numpy()
    spearman = spearmanr(residuals, sizes)[0]
    mse = (residuals * residuals).mean().item()
    counts = counts.reshape((c,) + hw_shape)
    # reference: float32 mean over images, then float32 mean over the channel axis (:55)
    spatial_miscoverage = (counts.cpu().numpy().astype(np.float32) / np.float32(n)).mean(axis=0)
    size_bins = torch.tensor([0, torch.quantile(sizes, 0.25), torch.quantile(sizes, 0.5), torch.quantile(sizes, 0.75)])
    buckets = torch.bucketize(sizes, size_bins) - 1
    losses_cpu = losses.cpu()
    stratified_risks = torch.tensor([losses_cpu[buckets == bucket].mean() for bucket in range(size_bins.shape[0])])
    return losses_cpu, sizes, spearman, stratified_risks, mse, spatial_miscoverage


def evaluate_from_loss_table(loss_table, n, alpha, delta):
    """Monte-Carlo re-split of a saved loss table (reference :62-74): shuffle the rows, calibrate on the first n, return
    the mean validation loss at the first lambda whose Hoeffding-Bentkus bound is <= `delta` (sic: the reference compares
    against delta, not alpha -- kept).  The num_lambdas bounds are solved in one multi-threaded C++ call
    (im2im_hb_mu_plus_batch) instead of one scipy root-find per lambda; like the reference's list of Python floats they
    are compared as float32."""
    with torch.no_grad():
        perm = torch.randperm(loss_table.shape[0])
        loss_table = loss_table[perm]
        calib_table, val_table = loss_table[:n], loss_table[n:]
        Rhats = calib_table.mean(dim=0)
        RhatPlus = hip_ops.hb_mu_plus_batch(Rhats, n, delta).to(torch.float32)
        hits = (RhatPlus <= delta).nonzero()
        if hits.numel() == 0:
            print("No rejections made!")
            idx_lambda = 0
        else:
            idx_lambda = hits[0]
        return val_table[:, idx_lambda].mean()


# ------------------------------------------------------------------ phase A: outputs into HBM
def collect_outputs(model, dataset, config, device, shard=True):
    """Eval-mode forward over `dataset`; returns (outputs [n,K,C,H,W], labels [n,C,H,W]) resident on
    `device`.  With torch.distributed active and shard=True only this rank's contiguous slice is run."""
    dist = _dist() if shard else None
    if config.get('dataset') == 'temca':                      # iterable dataset, one sample at a time (:106-109)
        samples = [s for s in iter(dataset)]
        if dist is not None:
            lo, hi = shard_bounds(len(samples), dist.get_rank(), dist.get_world_size())
            samples = samples[lo:hi]
        labels = torch.cat([s[1].unsqueeze(0) for s in samples], dim=0).to(device, torch.float32)
        outputs = torch.cat([calibration_repr(model, model(s[0].unsqueeze(0).to(device, torch.float32))) for s in samples], dim=0)
        return outputs.contiguous(), labels.contiguous()
    n_total = len(dataset)
    if dist is not None and not getattr(dataset, "im2im_local_shard", False):
        lo, hi = shard_bounds(n_total, dist.get_rank(), dist.get_world_size())
        dataset = Subset(dataset, range(lo, hi))
    n = len(dataset)
    outputs = labels = None
    counter = 0
    if isinstance(dataset, TensorDataset) and dataset.tensors[0].is_cuda:
        # inputs already resident in HBM: slice batches in place, no host round trip
        bs = config['batch_size']
        loader = ((dataset.tensors[0][s:s + bs], dataset.tensors[1][s:s + bs]) for s in range(0, n, bs))
    else:
        # [r6] a host dataset: the reference's loader (:118) behind the two-deep pinned prefetcher -- batch k+1 is collated, staged and
        # uploaded on a copy stream while batch k's forward runs (im2im_uq_amd/prefetch.py)
        # (a plain host TensorDataset walked in order: the loader's batches are contiguous row ranges -- sliced, not collated)
        seq = sequential_slices(dataset, config['batch_size'])
        loader = to_device(seq, device) if seq is not None else prefetch_loader(dataset, device, num_workers=0, batch_size=config['batch_size'])
    for batch in loader:
        out = calibration_repr(model, model(batch[0].to(device=device, dtype=torch.float32)))
        if outputs is None:
            outputs = torch.empty((n,) + tuple(out.shape[1:]), dtype=torch.float32, device=device)
            labels = torch.empty((n,) + tuple(batch[1].shape[1:]), dtype=torch.float32, device=device)
        b = out.shape[0]
        outputs[counter:counter + b] = out
        labels[counter:counter + b] = batch[1].to(device=device, dtype=torch.float32)
        counter += b
    if outputs is None:
        if dist is not None:                                  # fewer images than ranks: this rank simply contributes no rows
            return None, None
        raise ValueError("calibration dataset is empty")
    return outputs, labels


def lambda_grid(config):
    if config["uncertainty_type"] == "softmax":
        return torch.linspace(config['minimum_lambda_softmax'], config['maximum_lambda_softmax'], config['num_lambdas'])
    return torch.linspace(config['minimum_lambda'], config['maximum_lambda'], config['num_lambdas'])


def scan_loss_table(table, lambdas, alpha, delta):
    """Host half of the reference's lambda loop (:130-144) over a full [N,L] table (on any device) whose column j holds
    the losses at lambdas[j] - dlambda.  Returns (lhat, calib_loss_table with unvisited columns zero, trace).

    The scan itself is the C library's `im2im_rcps_scan` (csrc/hb_bound.cpp): per visited lambda the mean of a contiguous
    [N] fp32 column and one Hoeffding-Bentkus solve, with the reference's stop rule and default -- so a non-Python caller
    of the library lands on the same lambda-hat.  The transpose that makes the columns contiguous runs where the table
    lives, and the visited columns are copied into the result in one piece afterwards."""
    n, L = table.shape
    cols = table.t().contiguous().cpu()                       # row j = contiguous [N] losses, as torch.cat builds them
    stop, stopped, lhat_f, trace = hip_ops.rcps_scan(cols, lambdas, alpha, delta)
    if stopped:
        lhat = lambdas[stop]                                  # the grid point itself (a 0-dim fp32 tensor, as the reference keeps it)
    else:
        dlambda = lambdas[1] - lambdas[0]
        lhat = lambdas[-1] + dlambda - 1e-9
        assert float(lhat) == lhat_f
    if table.is_cuda:
        visited = torch.zeros_like(table)
        visited[:, stop:] = table[:, stop:]
        calib_loss_table = visited.cpu()
    else:
        calib_loss_table = torch.zeros((n, L))
        calib_loss_table[:, stop:] = table[:, stop:]
    return lhat, calib_loss_table, trace


def calibrate_model(model, dataset, config):
    """model, calib_loss_table[N, num_lambdas] = calibrate_model(model, dataset, config)  (reference :89-145)."""
    with torch.no_grad():
        print("Calibrating...")
        model.eval()
        alpha = config['alpha']
        delta = config['delta']
        device = config['device']
        lambdas = lambda_grid(config)
        rcps_loss_fn = get_rcps_loss_fn(config)
        model = model.to(device)
        outputs, labels = collect_outputs(model, dataset, config, device)
        dlambda = lambdas[1] - lambdas[0]
        model.set_lhat(lambdas[-1] + dlambda - 1e-9)
        form = sets_form(model)
        if outputs is None:                                   # empty shard of a multi-rank run
            table = torch.zeros((0, lambdas.numel()), dtype=torch.float32, device=device)
        elif rcps_loss_fn is fraction_missed_loss and form is not None:
            table = hip_ops.rcps_loss_table(outputs, labels, lambdas - dlambda, form=form)   # one pass, all lambdas
        else:                                                 # plugin loss: per-lambda, still device-resident
            ds = TensorDataset(outputs, labels)
            table = torch.stack([get_rcps_losses_from_outputs(model, ds, rcps_loss_fn, lam - dlambda, device)
                                 for lam in lambdas], dim=1).to(device)
        table = gather_rows(table)                            # row order = rank order = dataset order for contiguous shards
        lhat, calib_loss_table, trace = scan_loss_table(table, lambdas, alpha, delta)
        model.set_lhat(lhat)
        j, rhat, rhat_plus = trace[-1]
        print(f"Lambda: {float(lambdas[j]):.4f}  |  Rhat: {rhat:.4f}  |  RhatPlus: {rhat_plus:.4f}")
        print(f"Model's lhat set to {model.lhat}")
        return model, calib_loss_table
