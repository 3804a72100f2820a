"""CPU oracle for the step before the filter: per-draw restatement of the reference models'
``update(theta)`` (models/SARIMAX.py:59-107, models/VARMAX.py:95-150) with scipy's
``solve_discrete_lyapunov`` for the stationary P0.  TEST INFRASTRUCTURE ONLY.

Parity status: the reference models need pymc/pytensor (absent), so this is a line-by-line
restatement anchored on the reference's own structural tests (tests/test_VARMAX.py compares
these matrices with statsmodels; not runnable here) -- "parity unpinned" for this file."""
import numpy as np
from scipy.linalg import solve_discrete_lyapunov


def arma_update(theta, order, stationary_initialization=True):
    p, q = order
    k = max(p, q + 1)
    Z = np.r_[[1.0], np.zeros(k - 1)][None]
    T = np.eye(k, k=1)
    R = np.r_[[[1.0]], np.zeros(k - 1)[:, None]]
    a0, P0 = np.zeros(k), np.eye(k)
    Q, H = np.zeros((1, 1)), np.zeros((1, 1))
    cur = 0
    a0 = theta[cur: cur + k].copy(); cur += k
    if not stationary_initialization:
        P0 = theta[cur: cur + k * k].reshape(k, k).copy(); cur += k * k
    Q[0, 0] = theta[cur]; cur += 1
    T[np.arange(p), np.zeros(p, dtype=int)] = theta[cur: cur + p]; cur += p
    R[np.arange(1, q + 1), np.zeros(q, dtype=int)] = theta[cur: cur + q]; cur += q
    if stationary_initialization:
        P0 = solve_discrete_lyapunov(T, R @ Q @ R.T, method="bilinear")
    return dict(a0=a0, P0=P0, T=T, Z=Z, R=R, H=H, Q=Q)


def varmax_update(theta, k_obs, order, stationary_initialization=True, measurement_error=True):
    p, q = order
    k_order = max(p, 1) + q
    m = k_obs * k_order
    Z = np.zeros((k_obs, m)); Z[np.diag_indices(k_obs)] = 1
    T = np.zeros((m, m))
    if p > 1:
        T[k_obs: k_obs * p, 0: k_obs * (p - 1)] = np.eye(k_obs * (p - 1))
    if q > 1:
        T[-k_obs * (q - 1):, -k_obs * q: -k_obs] = np.eye(k_obs * (q - 1))
    R = np.zeros((m, k_obs)); R[0:k_obs, :] = np.eye(k_obs)
    if q > 0:
        end = -k_obs * (q - 1) if q > 1 else None
        R[k_obs * -q: end, :] = np.eye(k_obs)
    H = np.zeros((k_obs, k_obs)); P0 = np.zeros((m, m))
    cur = 0
    a0 = theta[cur: cur + m].copy(); cur += m
    if not stationary_initialization:
        P0 = theta[cur: cur + m * m].reshape(m, m).copy(); cur += m * m
    if p > 0:
        T[0:k_obs, 0: k_obs * p] = theta[cur: cur + k_obs * k_obs * p].reshape(k_obs, k_obs * p); cur += k_obs * k_obs * p
    if q > 0:
        T[0:k_obs, k_obs * max(1, p):] = theta[cur: cur + k_obs * k_obs * q].reshape(k_obs, k_obs * q); cur += k_obs * k_obs * q
    Q = theta[cur: cur + k_obs * k_obs].reshape(k_obs, k_obs).copy(); cur += k_obs * k_obs
    if measurement_error:
        H[np.diag_indices(k_obs)] = theta[cur: cur + k_obs]; cur += k_obs
    if stationary_initialization:
        P0 = solve_discrete_lyapunov(T, R @ Q @ R.T, method="bilinear")
    return dict(a0=a0, P0=P0, T=T, Z=Z, R=R, H=H, Q=Q)
