"""
helpers.py -- counterparts of the reference's helpers.py:14-25.

`set_seeds` seeds numpy's global legacy stream and torch exactly like the reference, because
in rng="compat" mode the epoch shuffle (problem.py:146) and the sampler's `sel`
(nn_modules.py:88) are drawn from that one stream.  `to_numpy` is the working version of the
reference helper (which recurses forever on torch >= 0.4).
"""
import numpy as np
import torch


def set_seeds(seed=0):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def to_numpy(x):
    if hasattr(x, "materialize"):
        x = x.materialize()
    return x.detach().cpu().float().numpy() if x.is_floating_point() else x.detach().cpu().numpy()
