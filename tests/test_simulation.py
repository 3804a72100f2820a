"""batched conditional simulation vs a numpy restatement of the reference's
numba_mvn_draws / conditional_simulation (utils/simulation.py:7-26) on the SAME normals, plus the
reference's own statistical check (tests/test_simulations.py: sample covariance within 1%)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W

pytestmark = pytest.mark.gpu


def _oracle(mus, covs, z, jitter):
    """x = mu + chol(blockdiag(cov) + jitter I) z  -- block by block (a block-diagonal Cholesky
    is the Cholesky of its blocks)."""
    B, n, k = mus.shape
    S = z.shape[1]
    out = np.empty((B * S, n, k))
    for b in range(B):
        for t in range(n):
            L = np.linalg.cholesky(covs[b, t] + jitter * np.eye(k))
            for j in range(S):
                out[b * S + j, t] = mus[b, t] + L @ z[b, j, t]
    return out


def test_simulation_matches_oracle_on_filter_tapes():
    from oracle import kalman_numpy as kn
    from pymc_statespace_amd import KalmanEngine

    eng = KalmanEngine()
    w = W.varmax(B=5, n=20, order=(2, 0))
    res = eng.forward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    rng = np.random.default_rng(0)
    for mu, cov in ((res.filtered_states, res.filtered_covs), (sa, sP)):
        z = rng.standard_normal((5, 7, 20, 10))
        out, info = eng.conditional_simulation(mu, cov, 7, jitter=1e-9, z=z)
        ref = _oracle(mu.cpu().numpy(), cov.cpu().numpy(), z, 1e-9)
        assert not info.cpu().numpy().any()
        assert_allclose(out.cpu().numpy(), ref, rtol=1e-9, atol=1e-9)
    assert kn is not None


def test_simulation_reference_signature_and_statistics():
    from pymc_statespace_amd.simulation import conditional_simulation

    rng = np.random.default_rng(1)
    n, k, S = 3, 4, 20000
    A = rng.standard_normal((1, n, k, k))
    covs = A @ np.swapaxes(A, -1, -2) + 0.5 * np.eye(k)
    mus = rng.standard_normal((1, n * k))
    sims = conditional_simulation(mus, covs, n, k, n_simulations=S)
    assert sims.shape == (S, n, k)
    for t in range(n):
        emp = np.cov(sims[:, t].T)
        assert np.max(np.abs(emp - covs[0, t])) / np.max(np.abs(covs[0, t])) < 0.05
        assert_allclose(sims[:, t].mean(0), mus.reshape(n, k)[t], atol=0.06)


def test_simulation_singular_covariance_needs_jitter():
    """Filtered covariances with H = 0 are exactly singular: the reference adds a jitter for
    this; without it the factorisation is flagged."""
    from pymc_statespace_amd import KalmanEngine

    eng = KalmanEngine()
    mu = np.zeros((1, 2, 2))
    cov = np.zeros((1, 2, 2, 2))
    cov[0, :, 0, 0] = 1.0
    z = np.ones((1, 1, 2, 2))
    out, info = eng.conditional_simulation(mu, cov, 1, jitter=1e-10, z=z)
    assert int(info[0].item()) == 0 and np.isfinite(out.cpu().numpy()).all()
    out, info = eng.conditional_simulation(mu, cov, 1, jitter=0.0, z=z)
    assert int(info[0].item()) != 0
