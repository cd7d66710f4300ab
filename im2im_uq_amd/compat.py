"""Loading checkpoints written by the reference (core/scripts/train.py:191-192 pickles the whole ModelWithUncertainty
with `torch.save(net.cpu().module, ...)`).

Those pickles name their classes by the reference's module paths (`core.models.trunks.unet.UNet`, ...).  This package
mirrors that module tree with the same class / function names and the same attribute names, so aliasing `core` to
`im2im_uq_amd.core` while unpickling turns a reference checkpoint into a model whose forward, loss and calibration run on
the HIP kernels -- no conversion step, the weights are the pickled tensors themselves.
"""
from __future__ import annotations

import contextlib
import importlib
import sys

import torch

_MIRRORED = (
    "core", "core.models", "core.models.add_uncertainty", "core.models.trunks", "core.models.trunks.unet",
    "core.models.trunks.unet_parts", "core.models.trunks.wnet", "core.models.finallayers", "core.models.finallayers.quantile_layer",
    "core.models.finallayers.quantile_l1_layer", "core.models.finallayers.gaussian_layer",
    "core.models.finallayers.residual_magnitude_layer", "core.models.finallayers.residual_magnitude_l1_layer",
    "core.models.finallayers.softmax_layer", "core.models.finallayers.inn_layer", "core.models.losses",
    "core.models.losses.pinball", "core.models.losses.inn", "core.calibration", "core.calibration.calibrate_model",
    "core.calibration.bounds", "core.utils",
)


@contextlib.contextmanager
def reference_module_aliases():
    """inside the block `import core.models...` (and therefore unpickling) resolves to this package's mirror; whatever
    `core*` modules were loaded before are restored afterwards."""
    saved = {k: v for k, v in sys.modules.items() if k == "core" or k.startswith("core.")}
    for k in saved:
        del sys.modules[k]
    try:
        for name in _MIRRORED:
            sys.modules[name] = importlib.import_module("im2im_uq_amd." + name)
        yield
    finally:
        for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def upgrade_(model):
    """attributes this package's modules set in __init__ but a reference pickle does not carry (unpickling does not run
    __init__)."""
    for m in model.modules():
        if not hasattr(m, "compute_dtype") and type(m).__module__.startswith("im2im_uq_amd."):
            m.compute_dtype = None
    return model


def load_reference_checkpoint(path, map_location=None):
    """torch.load of a whole-module checkpoint written by the reference (or by this package's train_net, whose files
    carry the same names) -> ModelWithUncertainty on the HIP kernels."""
    with reference_module_aliases():
        model = torch.load(path, map_location=map_location, weights_only=False)
    return upgrade_(model)
