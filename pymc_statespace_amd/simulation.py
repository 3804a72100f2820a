"""Host-side mirror of ``pymc_statespace.utils.simulation.conditional_simulation`` backed by
``kf_conditional_simulation``."""
from __future__ import annotations

import numpy as np
import torch

from .engine import get_engine


def conditional_simulation(mus, covs, n, k, n_simulations=100, jitter=1e-10, generator=None,
                           engine=None):
    """Reference signature (utils/simulation.py:17-26): ``mus`` (n_samples, n*k) or
    (n_samples, n, k); ``covs`` (n_samples, n, k, k).  Returns (n_samples*n_simulations, n, k):
    numpy in -> numpy out, torch(HBM) in -> torch out."""
    eng = engine or get_engine()
    as_torch = isinstance(mus, torch.Tensor)
    m_ = mus if as_torch else np.asarray(mus, dtype=np.float64)
    m_ = m_.reshape(m_.shape[0], n, k)
    out, info = eng.conditional_simulation(m_, covs, n_simulations, jitter, generator)
    bad = info.cpu().numpy()
    if np.any(bad != 0):
        raise np.linalg.LinAlgError(
            f"sample {int(np.flatnonzero(bad)[0])}: covariance block not positive definite")
    return out if as_torch else out.cpu().numpy()
