"""BASELINE config 5 (daily seasonality, m = 367, T = 730) and the HBM-workspace path (m > 64) at REAL length.

Round 1 checked this path on 5 - 8 time steps only; the long-series guards of the register-tile kernels exist
because a smoother there amplified rounding 1.45x per step.  Here: forward, gradient and smoother of
kf_forward_big.hip / kf_backward_big.hip / kf_smoother_big.hip over 300 steps at m in {70, 130, 367} against the
numpy / torch oracles (BLAS-backed, so m = 367 finishes in seconds), with the same error-growth assertion, and one
full T = 730 run of the C5 workload at one GPU's share of the batch (128 draws).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from oracle import kalman_numpy as kn

pytestmark = pytest.mark.gpu
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


@pytest.fixture(scope="module")
def eng():
    import torch

    from pymc_statespace_amd import KalmanEngine

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return KalmanEngine()


def _args(w, y=None):
    return (w["y"] if y is None else y,) + tuple(w[k] for k in KEYS)


LONG = {
    # seasonal structural models (level + trend + dummy seasonal, r = 3: the C5 family) and a dense random model
    "seasonal_m70": lambda: W.daily_seasonal(B=2, n=300, s=69),
    "dense_m130": lambda: W.random_dense(2, 300, 1, 130, 40, seed=41),
    "c5_seasonal_m367": lambda: W.daily_seasonal(B=2, n=300),
}


@pytest.mark.parametrize("case", sorted(LONG))
def test_workspace_forward_over_300_steps(eng, case):
    w = LONG[case]()
    assert eng.forward_kernel_name(w["B"], w["n"], 1, w["m"], w["r"]) == "big_gemm_kernel"
    y = w["y"].copy()
    y[[7, 150, 299]] = np.nan
    ref = kn.kalman_filter(*_args(w, y), mvn_const="exact")
    res = eng.forward(*_args(w, y), mvn_const="exact")
    assert not res.info.cpu().numpy().any()
    assert np.max(np.abs(res.logp.cpu().numpy() - ref["logp"])) < 1e-8 * max(1.0, np.abs(ref["logp"]).max() / 1e4)
    for k in ("ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
        got = getattr(res, k).cpu().numpy()
        assert_allclose(got, ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{case} {k}")
    # the per-step error of the covariance tape must not grow along the series
    err = np.abs(res.predicted_covs.cpu().numpy() - ref["predicted_covs"]).reshape(w["B"], w["n"] + 1, -1).max(2).max(0)
    assert err[-30:].max() < 50 * max(err[1:40].max(), 1e-14), (err[:5], err[-5:])
    pc = res.predicted_covs[:, 1:]
    assert bool((pc == pc.transpose(-1, -2)).all())


@pytest.mark.parametrize("case", sorted(LONG))
@pytest.mark.parametrize("predicted", [True, False])
def test_workspace_forward_fused_steps(eng, case, predicted, monkeypatch):
    """Without the filtered covariance tape (p = 1) a step is three launches: the gain kernel (m = P z summed from the shares the
    previous step's second product left), W = T P_f with P_f formed on the way into LDS and a_hat = T a_f + c by the workgroups
    ahead of the tiles, S = W T' + R Q R' mirrored in its epilogue (kf_forward_big.hip, PFIX / MIRROR).  Against the oracle, and
    against the separate kernels (KF_BIG_FUSE=0)."""
    w = LONG[case]()
    y = w["y"].copy()
    y[[0, 7, 150, 299]] = np.nan
    # 11 draws: the XCD mapping of the workgroups with a ragged last group of eight
    rng = np.random.default_rng(5)
    big = {k: np.concatenate([w[k]] * 6, axis=0)[:11].copy() for k in KEYS}
    big["Z"] = big["Z"] + 0.05 * rng.standard_normal(big["Z"].shape)   # a dense Z: every tile's share of P z counts
    big["H"] = big["H"] * (1.0 + rng.random((11, 1, 1)))
    args = (y,) + tuple(big[k] for k in KEYS)
    ref = kn.kalman_filter(*args, mvn_const="exact")
    kw = dict(mvn_const="exact", store_filtered=False, store_predicted=predicted)
    res = eng.forward(*args, **kw)
    assert not res.info.cpu().numpy().any()
    tol = 1e-8 * max(1.0, np.abs(ref["logp"]).max() / 1e4)
    assert np.max(np.abs(res.logp.cpu().numpy() - ref["logp"])) < tol
    assert_allclose(res.ll_obs.cpu().numpy(), ref["ll_obs"], rtol=1e-9, atol=1e-9)
    if predicted:
        assert_allclose(res.predicted_states.cpu().numpy(), ref["predicted_states"], rtol=1e-9, atol=1e-9)
        assert_allclose(res.predicted_covs.cpu().numpy(), ref["predicted_covs"], rtol=1e-9, atol=1e-9)
        pc = res.predicted_covs[:, 1:]
        if eng.last_forward_kernel() == "big_gemm_kernel":   # one triangle mirrored: symmetric bit for bit
            assert bool((pc == pc.transpose(-1, -2)).all())
        else:                                                # the CSR gather (a sparse T): every element its own short sum
            assert float((pc - pc.transpose(-1, -2)).abs().max()) <= 1e-13 * float(pc.abs().max())
    monkeypatch.setenv("KF_BIG_FUSE", "0")
    sep = eng.forward(*args, **kw)
    assert_allclose(res.logp.cpu().numpy(), sep.logp.cpu().numpy(), rtol=1e-11, atol=0.0)


@pytest.mark.parametrize("case", sorted(LONG))
def test_workspace_smoother_over_300_steps(eng, case):
    w = LONG[case]()
    flt = kn.kalman_filter(*_args(w))
    sel = slice(0, 1)  # one draw: the oracle takes an SVD per step
    ra, rP = kn.kalman_smoother(w["T"][sel], w["R"][sel], w["Q"][sel], flt["filtered_states"][sel],
                                flt["filtered_covs"][sel])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    n = w["n"]
    err = np.abs(sP[sel] - rP).reshape(1, n, -1).max(axis=2).max(axis=0)
    scale = max(1.0, np.abs(rP).max())
    assert err.max() < 1e-8 * scale, (err[:5], err[-5:])
    assert err[:20].max() < 50 * max(err[-40:-1].max(), 1e-14 * scale), "error grows backwards in time"
    assert_allclose(sa[sel], ra, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(ra).max()))
    assert np.array_equal(sa[:, -1], flt["filtered_states"][:, -1])
    assert np.array_equal(sP[:, -1], flt["filtered_covs"][:, -1])
    # smoothing never increases a variance
    d_s = np.diagonal(sP, axis1=-2, axis2=-1)
    d_f = np.diagonal(flt["filtered_covs"], axis1=-2, axis2=-1)
    assert (d_s <= d_f + 1e-9 * scale).all()


@pytest.mark.parametrize("case", sorted(LONG))
def test_workspace_gradient_over_300_steps(eng, case):
    from oracle import kalman_torch as kt

    w = LONG[case]()
    one = {k: w[k][:1] for k in KEYS}
    args = (w["y"],) + tuple(one[k] for k in KEYS)
    g_logp = np.array([1.0])
    _, _, gt = kt.logp_and_grads(*args, g_logp=g_logp)
    res = eng.forward(*args)
    g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g_logp)
    for k in KEYS:
        a, b = g[k].cpu().numpy(), gt[k]
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-9, (case, k)


def test_c5_full_length_one_gpu_share(eng):
    """BASELINE config 5 as configured: daily seasonality, m = 367, T = 730, 128 draws (1024 over 8 GPUs)."""
    import torch

    w = W.WORKLOADS["c5"](B=128)
    B, n, m = w["B"], w["n"], w["m"]
    assert (n, m) == (730, 367)
    dev = [eng.to_device(x) for x in _args(w)]
    res = eng.forward(*dev, mvn_const="exact", store_filtered=False, store_predicted=False)
    torch.cuda.synchronize()
    logp = res.logp.cpu().numpy()
    assert np.isfinite(logp).all() and not res.info.cpu().numpy().any()
    assert_allclose(logp, res.ll_obs.sum(-1).cpu().numpy(), rtol=1e-12, atol=1e-9)
    idx = np.array([0, 77, 127])
    ref = kn.kalman_filter(w["y"], *[w[k][idx] for k in KEYS], mvn_const="exact")
    assert np.max(np.abs(logp[idx] - ref["logp"])) < 1e-8 * max(1.0, np.abs(ref["logp"]).max() / 1e4)
    # filtered tapes + RTS smoother, 64 draws at a time (101 GB of filtered + smoothed covariances per chunk)
    ra, rP = kn.kalman_smoother(w["T"][idx[:1]], w["R"][idx[:1]], w["Q"][idx[:1]], ref["filtered_states"][:1],
                                ref["filtered_covs"][:1])
    for lo in (0, 64):
        sl = slice(lo, lo + 64)
        part = [dev[0]] + [x[sl] for x in dev[1:]]
        flt = eng.forward(*part, mvn_const="exact", store_ll_obs=False, store_predicted=False)
        # (with the filtered covariances requested P_f is formed; without, W = T P_f comes from T P and rank corrections: last-bit differences)
        assert torch.allclose(flt.logp, res.logp[sl], rtol=1e-12, atol=0.0)
        sa, sP = eng.smooth(part[3], part[5], part[7], flt.filtered_states, flt.filtered_covs)
        torch.cuda.synchronize()
        assert torch.equal(sa[:, -1], flt.filtered_states[:, -1]) and torch.equal(sP[:, -1], flt.filtered_covs[:, -1])
        assert bool(torch.isfinite(sa).all()) and bool(torch.isfinite(sP[:, ::73]).all())
        if lo == 0:
            assert_allclose(flt.filtered_states[0].cpu().numpy(), ref["filtered_states"][0], rtol=1e-9, atol=1e-9)
            assert_allclose(flt.filtered_covs[0, -1].cpu().numpy(), ref["filtered_covs"][0, -1], rtol=1e-9, atol=1e-9)
            assert_allclose(sa[0].cpu().numpy(), ra[0], rtol=1e-8, atol=1e-8)
            for t in (0, 1, 365, 728):
                assert_allclose(sP[0, t].cpu().numpy(), rP[0, t], rtol=1e-8, atol=1e-8, err_msg=f"t={t}")
        del flt, sa, sP
        torch.cuda.empty_cache()


def test_workspace_smoother_pinv_for_singular_prediction_covariance(eng):
    """kalman_smoother.py:91-92 uses pinv(P_hat).  m = 70 on the workspace path with two perfectly correlated states
    (exactly singular P_hat at every step) in one draw and a regular draw next to it: the Cholesky pass flags the
    first, sb_pinv_redo_kernel redoes it with the Moore-Penrose inverse; the second keeps its Cholesky result."""
    n, m = 25, 70
    w = W.random_dense(2, n, 1, m, 35, seed=90)
    T, R, P0 = w["T"].copy(), w["R"].copy(), w["P0"].copy()
    T[0, 1, :] = T[0, 0, :]          # states 0 and 1 of draw 0 move together ...
    R[0, 1, :] = R[0, 0, :]
    P0[0, 1, :] = P0[0, 0, :]
    P0[0, :, 1] = P0[0, :, 0]        # ... and start perfectly correlated
    P0[0, 1, 1] = P0[0, 0, 0]
    flt = kn.kalman_filter(w["y"], w["a0"], P0, T, w["Z"], R, w["H"], w["Q"])
    ra, rP = kn.kalman_smoother(T, R, w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(T, R, w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    assert np.isfinite(sa).all() and np.isfinite(sP).all()
    # regular draw: as before (Cholesky).  Singular draw: rank decisions sit at rounding level in both codes
    assert_allclose(sa[1], ra[1], rtol=1e-8, atol=1e-8)
    assert_allclose(sP[1], rP[1], rtol=1e-8, atol=1e-8)
    assert_allclose(sa[0], ra[0], rtol=1e-6, atol=1e-6)
    assert_allclose(sP[0], rP[0], rtol=1e-6, atol=1e-6)
    assert_allclose(sP[0, :, 0, :], sP[0, :, 1, :], rtol=0, atol=1e-7)   # the two states stay perfectly correlated


@pytest.mark.parametrize("m,B,n", [(97, 1, 1), (130, 1, 3), (367, 2, 2), (200, 9, 4)])
def test_workspace_forward_fused_edge_shapes(eng, m, B, n):
    """The three-launch step at its edges: one draw (seven empty slots in the group of eight the workgroup mapping uses), one step
    (no share of P z from a previous step), a series that is missing from end to end."""
    w = W.random_dense(B, n, 1, m, 3, seed=600 + m)
    w["y"] = np.nan_to_num(w["y"], nan=0.3)   # (a one-step series standardises to NaN)
    args = (w["y"],) + tuple(w[k] for k in KEYS)
    ref = kn.kalman_filter(*args, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact", store_filtered=False, store_predicted=True)
    assert not res.info.cpu().numpy().any()
    assert_allclose(res.logp.cpu().numpy(), ref["logp"], rtol=1e-11, atol=1e-12)
    assert_allclose(res.predicted_states.cpu().numpy(), ref["predicted_states"], rtol=1e-9, atol=1e-10)
    assert_allclose(res.predicted_covs.cpu().numpy(), ref["predicted_covs"], rtol=1e-9, atol=1e-10)
    y = np.full_like(w["y"], np.nan)
    res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact", store_filtered=False, store_predicted=False)
    assert np.array_equal(res.logp.cpu().numpy(), np.zeros(B))
