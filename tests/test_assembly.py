"""theta -> matrices assembly maps (host logic, CPU) and, on the GPU, the
batched scatter + batched discrete Lyapunov solve against the per-draw oracle."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import assembly_numpy as an
from pymc_statespace_amd.assembly import ORDER, arma_layout, varmax_layout


def _stable_theta_arma(rng, B, lay, order):
    p, q = order
    th = rng.standard_normal((B, lay.n_theta))
    m = lay.m
    off = m + (0 if lay.stationary else m * m)
    th[:, off] = 0.3 + np.abs(th[:, off])              # sigma_state > 0
    th[:, off + 1: off + 1 + p] = rng.uniform(-0.4, 0.4, (B, p)) / max(p, 1)
    return th


def _stable_theta_varmax(rng, B, lay, k, order):
    p, q = order
    th = 0.2 * rng.standard_normal((B, lay.n_theta))
    m = lay.m
    cur = m + (0 if lay.stationary else m * m)
    th[:, cur: cur + k * k * p] *= 0.5 / max(p, 1)
    cur += k * k * (p + q)
    L = np.tril(0.2 * rng.standard_normal((B, k, k)))
    L[:, np.arange(k), np.arange(k)] = 0.5 + np.abs(L[:, np.arange(k), np.arange(k)])
    th[:, cur: cur + k * k] = (L @ np.swapaxes(L, 1, 2)).reshape(B, k * k)
    cur += k * k
    if "obs_cov" in lay.param_names:
        th[:, cur: cur + k] = 0.1 + np.abs(th[:, cur: cur + k])
    return th


ARMA_ORDERS = [(1, 1), (2, 0), (0, 2), (3, 2), (1, 4)]
VARMAX_ORDERS = [(1, 0), (2, 0), (2, 1), (1, 2), (0, 1), (3, 2)]


@pytest.mark.parametrize("order", ARMA_ORDERS)
@pytest.mark.parametrize("stationary", [True, False])
def test_arma_maps_match_reference_update(order, stationary):
    lay = arma_layout(order, stationary)
    rng = np.random.default_rng(0)
    th = _stable_theta_arma(rng, 5, lay, order)
    got = lay.apply_numpy(th)
    for b in range(5):
        ref = an.arma_update(th[b], order, stationary)
        for k in ORDER:
            if k == "P0" and stationary:
                continue  # filled by the Lyapunov solve
            assert_allclose(got[k][b], ref[k].reshape(got[k][b].shape), err_msg=k)
    assert lay.param_names == (["x0", "sigma_state", "rho", "theta"] if stationary else
                               ["x0", "P0", "sigma_state", "rho", "theta"])


@pytest.mark.parametrize("order", VARMAX_ORDERS)
@pytest.mark.parametrize("measurement_error", [True, False])
def test_varmax_maps_match_reference_update(order, measurement_error):
    k = 3
    lay = varmax_layout(k, order, True, measurement_error)
    rng = np.random.default_rng(1)
    th = _stable_theta_varmax(rng, 4, lay, k, order)
    got = lay.apply_numpy(th)
    for b in range(4):
        ref = an.varmax_update(th[b], k, order, True, measurement_error)
        for name in ORDER:
            if name == "P0":
                continue
            assert_allclose(got[name][b], ref[name], err_msg=name)
    # parameter counts of the reference (models/VARMAX.py:32-40)
    p, q = order
    m = k * (max(p, 1) + q)
    assert lay.n_theta == m + k * k * p + k * k * q + k * k + (k if measurement_error else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["arma11", "arma32", "varmax_k5_21", "varmax_k3_20"])
def test_hip_assembly_and_lyapunov_match_oracle(case):
    import torch

    from pymc_statespace_amd import KalmanEngine
    from pymc_statespace_amd.assembly import BatchedStateSpace

    rng = np.random.default_rng(2)
    B = 67
    if case.startswith("arma"):
        order = (1, 1) if case == "arma11" else (3, 2)
        lay = arma_layout(order, True)
        th = _stable_theta_arma(rng, B, lay, order)
        ref = [an.arma_update(th[b], order, True) for b in range(B)]
    else:
        k, order = (5, (2, 1)) if case == "varmax_k5_21" else (3, (2, 0))
        lay = varmax_layout(k, order, True, True)
        th = _stable_theta_varmax(rng, B, lay, k, order)
        ref = [an.varmax_update(th[b], k, order, True, True) for b in range(B)]
    eng = KalmanEngine()
    mats = BatchedStateSpace(lay, eng)(th)
    torch.cuda.synchronize()
    assert int(mats["lyapunov_info"].abs().sum().item()) == 0
    for name in ORDER:
        got = mats[name].cpu().numpy()
        want = np.stack([r[name].reshape(got.shape[1:]) for r in ref])
        assert_allclose(got, want, rtol=1e-10, atol=1e-12, err_msg=name)
    # the assembled batch goes straight into the filter
    y = np.random.default_rng(3).standard_normal((30, lay.p))
    res = eng.forward(y, *[mats[n] for n in ORDER], filter_type="standard")
    assert torch.isfinite(res.logp).all()


@pytest.mark.gpu
def test_hip_lyapunov_flags_nonstationary_draws():
    from pymc_statespace_amd import KalmanEngine

    eng = KalmanEngine()
    T = np.stack([0.5 * np.eye(3), 1.01 * np.eye(3)])
    S = np.stack([np.eye(3), np.eye(3)])
    P, info = eng.lyapunov(T, S=S)
    assert info.cpu().numpy().tolist() == [0, 1]
    assert_allclose(P[0].cpu().numpy(), np.eye(3) / (1 - 0.25), rtol=1e-13)
