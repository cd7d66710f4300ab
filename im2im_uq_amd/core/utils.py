"""fix_randomness -- the one helper of the reference's core/utils.py (:15-19) the training / calibration path uses (the
rest of that file is matplotlib plotting, out of scope)."""
import random

import numpy as np
import torch


def fix_randomness(seed=0):
    np.random.seed(seed=seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    random.seed(seed)
