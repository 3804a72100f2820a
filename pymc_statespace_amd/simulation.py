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


def simulate_statespace(T, Z, R, H, Q, n_steps, x0=None, generator=None, engine=None):
    """Reference signature (utils/simulation.py:29-64), one trajectory of one parameter set:
    returns ``(simulated_states (n_steps, m), simulated_obs (n_steps, p))``."""
    eng = engine or get_engine()
    as_torch = isinstance(T, torch.Tensor)
    arrs = [x if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64) for x in (T, Z, R, H, Q)]
    x0b = None if x0 is None else (x0 if isinstance(x0, torch.Tensor) else np.asarray(x0, dtype=np.float64))[None]
    states, obs, info = eng.simulate_statespace(*[a[None] for a in arrs], n_steps, 1, x0b, generator)
    if int(info[0].item()) != 0:
        raise np.linalg.LinAlgError("Q or H is not positive definite")
    return (states[0], obs[0]) if as_torch else (states[0].cpu().numpy(), obs[0].cpu().numpy())


def unconditional_simulations(T, Z, R, H, Q, n_steps=100, n_simulations=100, x0=None, generator=None,
                              engine=None):
    """Batched counterpart of reference ``unconditional_simulations`` (utils/simulation.py:67-79):
    that function rebuilds the matrices per posterior sample from ``thetas`` with the model's
    compiled update functions and loops over samples and simulations in Python; here the matrices
    of all samples are passed with a leading axis (e.g. from ``assembly.BatchedStateSpace``) and
    every trajectory is one GPU lane.  Returns ``(states (samples*n_simulations, n_steps, m),
    observed (samples*n_simulations, n_steps, p))``, sample-major like the reference."""
    eng = engine or get_engine()
    as_torch = isinstance(T, torch.Tensor)
    states, obs, info = eng.simulate_statespace(T, Z, R, H, Q, n_steps, n_simulations, x0, generator)
    bad = info.cpu().numpy()
    if np.any(bad != 0):
        raise np.linalg.LinAlgError(f"sample {int(np.flatnonzero(bad)[0])}: Q or H not positive definite")
    return (states, obs) if as_torch else (states.cpu().numpy(), obs.cpu().numpy())
