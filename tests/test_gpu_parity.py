"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle.

Structure mirrors the reference's tests/test_kalman_filter.py (shape contract, missing data,
statsmodels parity on Nile, last smoothed == last filtered), plus what the reference never
pinned: p > 1, batches of draws, full BASELINE sizes.

Tolerances (fp64, stated per BASELINE.json north_star):
  * |dlogp| < 1e-8 absolute on the standardized synthetic workloads (|logp| ~ 1e2..1e4);
  * rel 1e-9 on the raw-Nile fixtures (P0 = 1e6 I, |logp| ~ 6e5: two fp64 codes differ ~1e-10);
  * tapes: rtol = atol = 1e-9.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from conftest import golden_files, h_is_diagonal, load_golden
from oracle import c_oracle as co
from oracle import kalman_numpy as kn

pytestmark = pytest.mark.gpu

FILTERS = ["standard", "cholesky", "univariate", "single"]
TAPES = ["filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs"]


@pytest.fixture(scope="module")
def eng():
    import torch

    from pymc_statespace_amd import KalmanEngine

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return KalmanEngine()


def _args(w, y=None):
    return (w["y"] if y is None else y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])


def _np(res):
    return dict(logp=res.logp.cpu().numpy(), ll_obs=res.ll_obs.cpu().numpy(),
                filtered_states=res.filtered_states.cpu().numpy(),
                predicted_states=res.predicted_states.cpu().numpy(),
                filtered_covs=res.filtered_covs.cpu().numpy(),
                predicted_covs=res.predicted_covs.cpu().numpy(), info=res.info.cpu().numpy())


# ----------------------------------------------------------------------------------------------
# golden vectors (statsmodels) -- reference tests/test_kalman_filter.py:226-241
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_files())
@pytest.mark.parametrize("ft", FILTERS)
def test_hip_matches_statsmodels_golden(eng, name, ft):
    g = load_golden(name)
    p = g["Z"].shape[1]
    if ft == "single" and p != 1:
        with pytest.raises(AssertionError, match="at most 1-dimensional"):
            eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"],
                        filter_type=ft)
        return
    if ft == "univariate" and not h_is_diagonal(g["H"]):
        pytest.skip("UnivariateFilter uses diag(H) only (reference kalman_filter.py:493)")
    out = _np(eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"],
                          filter_type=ft, mvn_const="exact"))
    tol = 2e-9 if name.startswith("nile_llt") else 1e-10
    for ours, theirs in [("filtered_states", "filtered_state"), ("predicted_states", "predicted_state"),
                         ("filtered_covs", "filtered_state_cov"),
                         ("predicted_covs", "predicted_state_cov"), ("ll_obs", "llf_obs")]:
        assert_allclose(out[ours], g[theirs], rtol=tol, atol=tol, err_msg=f"{name} {ft} {ours}")
    assert_allclose(out["logp"], g["llf_obs"].sum(-1), rtol=tol, atol=1e-8)
    assert not out["info"].any()


def test_known_answers(eng):
    """Nile LLT -606737.514026..., local level -641.5855784594156."""
    g = load_golden("nile_llt.npz")
    out = _np(eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"]))
    assert_allclose(out["logp"][0], -606737.514026642, rtol=1e-10)
    assert_allclose(out["filtered_states"][0, -1], [730.6659435031556, -17.64853509699647], rtol=1e-8)
    g = load_golden("nile_local_level.npz")
    out = _np(eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"]))
    assert_allclose(out["logp"][0], -641.5855784594156, rtol=1e-13)
    assert_allclose(out["filtered_states"][0, -1, 0], 798.3702926083578, rtol=1e-12)
    assert_allclose(out["filtered_covs"][0, -1, 0, 0], 4032.1579418087827, rtol=1e-11)


# ----------------------------------------------------------------------------------------------
# HIP vs the numpy oracle on seeded inputs, every filter, both constant conventions
# ----------------------------------------------------------------------------------------------
CASES = {
    "c2_arma11": lambda: W.arma11(B=130, n=200),
    "headline_m10": lambda: W.structural(B=67, n=150),
    "c4_varmax_m10": lambda: W.varmax(B=11, n=80, order=(2, 0)),
    "c4_varmax_m15": lambda: W.varmax(B=9, n=60, order=(2, 1)),
    "dense_p3m4": lambda: W.random_dense(19, 60, 3, 4, 2, seed=3),
    "dense_p2m6_fullH": lambda: W.random_dense(7, 40, 2, 6, 3, seed=4, h_diag=False),
    "dense_p9m12": lambda: W.random_dense(5, 30, 9, 12, 3, seed=5),
    "dense_p1m33": lambda: W.random_dense(3, 25, 1, 33, 2, seed=7),
    "dense_p1m1": lambda: W.random_dense(70, 30, 1, 1, 1, seed=8),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("ft", FILTERS)
@pytest.mark.parametrize("mvn", ["reference", "exact"])
def test_hip_matches_oracle(eng, case, ft, mvn):
    w = CASES[case]()
    if ft == "single" and w["p"] != 1:
        pytest.skip("single is p == 1 only")
    if ft == "univariate" and not h_is_diagonal(w["H"]):
        pytest.skip("univariate uses diag(H) only")
    out = _np(eng.forward(*_args(w), filter_type=ft, mvn_const=mvn))
    ref = kn.kalman_filter(*_args(w), filter_type=ft, mvn_const=mvn, cholesky_triangle="correct")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{case} {ft} {k}")
    assert_allclose(out["logp"], out["ll_obs"].sum(-1), rtol=1e-13, atol=1e-10)
    assert_allclose(out["predicted_states"][:, 0], w["a0"])
    assert_allclose(out["predicted_covs"][:, 0], w["P0"])


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate"])
def test_missing_data_rows_and_cells(eng, ft):
    """Whole-row NaN (what the reference tests) and single-cell NaN (row deletion)."""
    w = W.random_dense(6, 40, 3, 4, 2, seed=8)
    y = w["y"].copy()
    y[[0, 3, 17, 39]] = np.nan
    y[5, 0] = np.nan
    y[9, 1:] = np.nan
    out = _np(eng.forward(*_args(w, y), filter_type=ft, mvn_const="exact"))
    ref = kn.kalman_filter(*_args(w, y), filter_type=ft, mvn_const="exact",
                           cholesky_triangle="correct", partial_missing="delete")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9)
        assert not np.isnan(out[k]).any()
    assert np.all(out["ll_obs"][:, [0, 3, 17, 39]] == 0.0)


def test_per_draw_data_flag(eng):
    w = W.structural(B=9, n=40)
    rng = np.random.default_rng(0)
    yb = w["y"][None] + 0.1 * rng.standard_normal((9, 40, 1))
    out = _np(eng.forward(*_args(w, yb)))
    for b in range(9):
        ref = kn.kalman_filter(yb[b], w["a0"][b], w["P0"][b], w["T"][b], w["Z"][b], w["R"][b],
                               w["H"][b], w["Q"][b])
        assert abs(out["logp"][b] - ref["logp"][0]) < 1e-9


def test_intercepts_c_d(eng):
    w = W.random_dense(5, 30, 2, 4, 2, seed=31)
    rng = np.random.default_rng(1)
    c = 0.1 * rng.standard_normal((5, 4))
    d = 0.1 * rng.standard_normal((5, 2))
    out = _np(eng.forward(*_args(w), c=c, d=d))
    ref = kn.kalman_filter(*_args(w), c=c, d=d)
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-9
    assert_allclose(out["predicted_states"], ref["predicted_states"], rtol=1e-10, atol=1e-10)


# ----------------------------------------------------------------------------------------------
# reference interface mirror -- tests/test_kalman_filter.py:66-99,159-222
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ft", FILTERS)
@pytest.mark.parametrize("pmrn", [(1, 1, 1, 10), (1, 2, 2, 10), (1, 5, 2, 10), (1, 5, 1, 10),
                                  (5, 5, 1, 10)])
def test_reference_shapes_and_errors(ft, pmrn):
    from pymc_statespace_amd.filters import FILTER_FACTORY

    p, m, r, n = pmrn
    w = W.reference_test_inputs(p, m, r, n)
    inputs = (w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0],
              w["H"][0], w["Q"][0])
    flt = FILTER_FACTORY[ft]()
    if ft == "single" and p > 1:
        with pytest.raises(AssertionError,
                           match="UnivariateTimeSeries filter requires data be at most 1-dimensional"):
            flt(*inputs)
        return
    outs = flt(*inputs)
    shapes = [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)]
    for o, s in zip(outs, shapes):
        assert np.shape(o) == s
    # (the reference's wrong-triangle Cholesky solve for p > 1 is not reproduced: SURVEY A8)
    ref = kn.reference_outputs(*inputs, filter_type=ft, cholesky_triangle="correct")
    assert_allclose(outs[4], ref[6], rtol=1e-11)
    assert_allclose(outs[0], ref[0], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("ft", FILTERS)
@pytest.mark.parametrize("p", [1, 5])
def test_missing_row_gives_no_nan(ft, p):
    from pymc_statespace_amd.filters import FILTER_FACTORY

    if ft == "single" and p > 1:
        pytest.skip("single is p == 1 only")
    w = W.reference_test_inputs(p, 5, 1, 10, missing_rows=(4,))
    outs = FILTER_FACTORY[ft]()(w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0],
                                w["Z"][0], w["R"][0], w["H"][0], w["Q"][0])
    for o in outs:
        assert not np.any(np.isnan(o))


def test_bad_ndim_raises_value_error():
    from pymc_statespace_amd.filters import StandardFilter

    w = W.reference_test_inputs(1, 2, 1, 5)
    with pytest.raises(ValueError, match="it should either 2 .static. or 3 .time varying."):
        StandardFilter()(w["y"], w["a0"][0], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0],
                         w["H"][0], w["Q"][0][0])


def test_not_positive_definite_raises_linalgerror():
    from pymc_statespace_amd.filters import StandardFilter

    w = W.reference_test_inputs(1, 2, 1, 5)
    H = np.array([[-10.0]])
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite at time step 0"):
        StandardFilter()(w["y"], w["a0"][0], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0], H,
                         w["Q"][0])


# ----------------------------------------------------------------------------------------------
# BASELINE.json full sizes: C oracle on a sample + size-independent properties
# ----------------------------------------------------------------------------------------------
FULL = {
    "c2": dict(ft="standard", sample=256),
    "headline": dict(ft="standard", sample=96),
    "c4_m10": dict(ft="univariate", sample=48),
    # BASELINE configs 3 and 4 at their full sizes (B = 8192 x T = 2000 Cholesky; B = 16384, m = 15, p = 5)
    "c3": dict(ft="cholesky", sample=64),
    "c4": dict(ft="standard", sample=48),
}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_logp_and_properties(eng, name):
    import torch

    cfg = FULL[name]
    w = W.WORKLOADS[name]()
    B, n, m = w["B"], w["n"], w["m"]
    dev = [eng.to_device(x) for x in _args(w)]
    res = eng.forward(*dev, filter_type=cfg["ft"], mvn_const="exact", store_filtered=False,
                      store_predicted=False)
    torch.cuda.synchronize()
    logp = res.logp.cpu().numpy()
    assert np.isfinite(logp).all() and not res.info.cpu().numpy().any()
    # (1) parity on a bounded sample of draws spread over the batch, |dlogp| < 1e-8
    idx = np.linspace(0, B - 1, cfg["sample"]).astype(int)
    ref = co.kalman_filter(w["y"], *[w[k][idx] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")],
                           filter_type=cfg["ft"], mvn_const="exact", full=False)
    scale = np.maximum(1.0, np.abs(ref["logp"]) / 1e4)  # |logp| reaches 1e5 in c2 (sigma -> 0.05)
    assert np.max(np.abs(logp[idx] - ref["logp"]) / scale) < 1e-8
    # (2) logp == sum of ll_obs
    assert_allclose(logp, res.ll_obs.sum(-1).cpu().numpy(), rtol=1e-12, atol=1e-9)
    # (3) permuting the draws permutes the outputs (bit for bit) -- batch independence
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev[1].device)
    res_p = eng.forward(dev[0], *[x[perm].contiguous() for x in dev[1:]], filter_type=cfg["ft"],
                        mvn_const="exact", store_filtered=False, store_predicted=False)
    assert torch.equal(res_p.logp, res.logp[perm])
    # (4) lanes-per-draw is a pure scheduling knob: same bits
    eng.set_lanes_per_draw(16 if m <= 4 else 32)
    res_g = eng.forward(*dev, filter_type=cfg["ft"], mvn_const="exact", store_filtered=False,
                        store_predicted=False)
    eng.set_lanes_per_draw(0)
    assert torch.equal(res_g.logp, res.logp)


def test_full_size_tapes_headline(eng):
    """Full-tape run at the headline size: tapes of sampled draws vs the C oracle."""
    import torch

    w = W.WORKLOADS["headline"]()
    res = eng.forward(*_args(w), mvn_const="exact")
    torch.cuda.synchronize()
    idx = np.array([0, 1, 1777, 4095])
    ref = co.kalman_filter(w["y"], *[w[k][idx] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")],
                           mvn_const="exact", full=True)
    for k in TAPES:
        got = getattr(res, k)[torch.as_tensor(idx)].cpu().numpy()
        assert_allclose(got, ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    # covariances stay symmetric and the predicted tape starts at (a0, P0)
    pP = res.predicted_covs[:64]
    assert torch.equal(pP, pP.transpose(-1, -2))
    assert torch.equal(res.predicted_states[:, 0].cpu(), torch.as_tensor(w["a0"]))


# ----------------------------------------------------------------------------------------------
# RTS smoother -- reference kalman_smoother.py:56-104, tests/test_kalman_filter.py:212-241
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_files())
def test_hip_smoother_matches_statsmodels_golden(eng, name):
    g = load_golden(name)
    res = eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"],
                      mvn_const="exact")
    sa, sP = eng.smooth(g["T"], g["R"], g["Q"], res.filtered_states, res.filtered_covs)
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    if name.startswith("nile_llt"):
        # the reference's own tolerances: atol 1e-7, covariances from index 5 on (:236-241)
        assert_allclose(sa, g["smoothed_state"], rtol=1e-7, atol=1e-7)
        assert_allclose(sP[:, 5:], g["smoothed_state_cov"][:, 5:], rtol=1e-7, atol=1e-7)
    elif name.startswith("arma11"):
        # H = 0: P_hat is exactly rank deficient; pinv's rank decision sits at rounding level
        assert_allclose(sa, g["smoothed_state"], rtol=1e-6, atol=1e-6)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-2, atol=1e-2)
    else:
        assert_allclose(sa, g["smoothed_state"], rtol=1e-9, atol=1e-9)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-9, atol=1e-9)
    # last smoothed == last filtered (:212-222)
    assert np.array_equal(sa[:, -1], res.filtered_states[:, -1].cpu().numpy())
    assert np.array_equal(sP[:, -1], res.filtered_covs[:, -1].cpu().numpy())


SMOOTH_CASES = {
    "headline_m10": lambda: W.structural(B=33, n=120),
    "c4_varmax_m15": lambda: W.varmax(B=7, n=60, order=(2, 1)),
    "dense_p3m4": lambda: W.random_dense(19, 60, 3, 4, 2, seed=3),
    "dense_p2m20_fullrank": lambda: W.random_dense(5, 40, 2, 20, 20, seed=6),
    "dense_p1m33_fullrank": lambda: W.random_dense(3, 25, 1, 33, 33, seed=7),
    "nile_local_level": lambda: W.nile_local_level(B=9),
}


@pytest.mark.parametrize("case", sorted(SMOOTH_CASES))
@pytest.mark.parametrize("lanes", [0, 16])
def test_hip_smoother_matches_oracle(eng, case, lanes):
    w = SMOOTH_CASES[case]()
    res = eng.forward(*_args(w))
    eng.set_lanes_per_draw(lanes)
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    eng.set_lanes_per_draw(0)
    na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(),
                                res.filtered_covs.cpu().numpy())
    assert_allclose(sa.cpu().numpy(), na, rtol=1e-9, atol=1e-9)
    assert_allclose(sP.cpu().numpy(), nP, rtol=1e-9, atol=1e-9)


def test_hip_smoother_singular_prediction_covariance(eng):
    """Exactly singular P_hat (two perfectly correlated states): the pinv (Jacobi) path."""
    w = W.reference_test_inputs(1, 2, 1, 30)
    w["T"] = np.array([[[1.0, 0.0], [1.0, 0.0]]])
    w["R"] = np.array([[[1.0], [1.0]]])
    w["P0"] = np.array([[[1.0, 1.0], [1.0, 1.0]]])
    w["y"] = np.cumsum(np.random.default_rng(0).standard_normal((30, 1)), axis=0)
    res = eng.forward(*_args(w))
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(),
                                res.filtered_covs.cpu().numpy())
    assert_allclose(sa.cpu().numpy(), na, rtol=1e-9, atol=1e-9)
    assert_allclose(sP.cpu().numpy(), nP, rtol=1e-9, atol=1e-9)


def test_reference_smoother_interface():
    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    p, m, r, n = 1, 5, 1, 10
    w = W.reference_test_inputs(p, m, r, n)
    inputs = (w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0],
              w["H"][0], w["Q"][0])
    fs, ps, fc, pc, ll, llo = StandardFilter()(*inputs)
    ss, sc = KalmanSmoother()(w["T"][0], w["R"][0], w["Q"][0], fs, fc)
    assert ss.shape == (n, m, 1) and sc.shape == (n, m, m)
    assert_allclose(fs[-1], ss[-1])  # reference tests/test_kalman_filter.py:212-222
    assert_allclose(fc[-1], sc[-1])
    ref = kn.reference_outputs(*inputs)
    assert_allclose(ss, ref[2], rtol=1e-10, atol=1e-10)
    assert_allclose(sc, ref[5], rtol=1e-10, atol=1e-10)


# ----------------------------------------------------------------------------------------------
# gradient (kf_filter_backward) vs torch.autograd of the reference operation sequence
# (the reference relies on PyTensor autodiff; no reference test pins it)
# ----------------------------------------------------------------------------------------------
GRAD_CASES = {
    "c1_local_level": (lambda: W.nile_local_level(B=3), "standard"),
    "c2_arma11": (lambda: W.arma11(B=6, n=40), "standard"),
    "headline_m10": (lambda: W.structural(B=5, n=30), "standard"),
    "headline_m10_cholesky": (lambda: W.structural(B=5, n=30), "cholesky"),
    "headline_m10_single": (lambda: W.structural(B=3, n=20), "single"),
    "c4_varmax_m10": (lambda: W.varmax(B=3, n=15, order=(2, 0)), "standard"),
    "c4_varmax_m10_univariate": (lambda: W.varmax(B=3, n=15, order=(2, 0)), "univariate"),
    "c4_varmax_m15": (lambda: W.varmax(B=2, n=10, order=(2, 1)), "standard"),
    "dense_p3m4_fullH": (lambda: W.random_dense(4, 20, 3, 4, 2, seed=5, h_diag=False), "standard"),
    "dense_p9m12": (lambda: W.random_dense(2, 10, 9, 12, 3, seed=5), "standard"),
}


@pytest.mark.parametrize("case", sorted(GRAD_CASES))
@pytest.mark.parametrize("missing", [False, True])
def test_hip_gradient_matches_autograd(eng, case, missing):
    from oracle import kalman_torch as kt

    mk, ft = GRAD_CASES[case]
    w = mk()
    B, n, m, p = w["B"], w["n"], w["m"], w["p"]
    y = w["y"].copy()
    if missing:
        y[2] = np.nan
        y[5, 0] = np.nan
        y[-1] = np.nan
    rng = np.random.default_rng(0)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    args = _args(w, y)
    ft_oracle = "standard" if ft == "cholesky" else ft
    lp, _, gt = kt.logp_and_grads(*args, c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll,
                                  filter_type=ft_oracle)
    res = eng.forward(*args, c=c, d=d, filter_type=ft)
    g = eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                     filter_type=ft, want_cd=True)
    assert np.max(np.abs(res.logp.cpu().numpy() - lp)) < 1e-9
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if ft == "univariate" and k in ("P0", "Q", "H"):
            # the univariate recursion's entrywise partials w.r.t. symmetric inputs differ from
            # the standard one's; their symmetric parts (all a symmetric parameterisation can
            # see) agree -- tests/test_oracle_torch.py
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        scale = 1.0 + np.max(np.abs(b))
        assert np.max(np.abs(a - b)) / scale < 1e-10, f"{case} d/d{k}"


def test_hip_gradient_directional_derivative_full_size(eng):
    """Headline size: <grad, direction> equals the central difference of sum(logp) along a
    random direction in (T, H, Q) -- a size-independent check of the backward kernel."""
    import torch

    w = W.WORKLOADS["headline"]()
    dev = {k: eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")}
    order = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
    res = eng.forward(*[dev[k] for k in order], store_filtered=False, store_ll_obs=False)
    ones = torch.ones(w["B"], device=dev["T"].device, dtype=torch.float64)
    g = eng.backward(*[dev[k] for k in order], predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=ones)
    gen = torch.Generator(device="cpu").manual_seed(1)
    dirs = {k: torch.randn(dev[k].shape, generator=gen, dtype=torch.float64).to(dev[k].device)
            for k in ("T", "H", "Q")}
    for k in ("H", "Q"):
        dirs[k] = dirs[k] + dirs[k].transpose(-1, -2)
    analytic = sum((g[k] * dirs[k]).sum(dim=tuple(range(1, dirs[k].ndim))) for k in dirs)
    h = 1e-6

    def logp_at(s):
        pert = dict(dev)
        for k in dirs:
            pert[k] = dev[k] + s * dirs[k]
        return eng.forward(*[pert[k] for k in order], store_filtered=False, store_ll_obs=False,
                           store_predicted=False).logp

    fd = (logp_at(h) - logp_at(-h)) / (2 * h)
    rel = ((analytic - fd).abs() / (1.0 + fd.abs())).max().item()
    assert rel < 1e-5, rel


# ----------------------------------------------------------------------------------------------
# the PyTensor Op's numeric bodies (op.py) -- what perform()/grad() execute
# ----------------------------------------------------------------------------------------------
def test_op_perform_bodies_match_oracle():
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op

    g = load_golden("nile_llt.npz")
    n, m, p = g["y"].shape[0], 2, 1
    inputs = [g["y"][..., None], g["a0"][0][:, None], g["P0"][0], g["T"][0], g["Z"][0], g["R"][0],
              g["H"][0], g["Q"][0], np.zeros((m, 1)), np.zeros((p, 1))]
    outs = op.perform_filter(*inputs)
    ref = kn.reference_outputs(*inputs[:8])
    shapes = [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)]
    for o, s in zip(outs, shapes):
        assert np.shape(o) == s
    assert_allclose(outs[4], ref[6], rtol=1e-10)
    assert_allclose(outs[0], ref[0], rtol=1e-8, atol=1e-8)
    ss, sc = op.perform_smoother(inputs[3], inputs[5], inputs[7], outs[0], outs[2])
    assert ss.shape == (n, m, 1) and sc.shape == (n, m, m)
    assert_allclose(ss, ref[2], rtol=1e-6, atol=1e-6)

    w = W.structural(B=1, n=25)
    inputs = [w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0],
              w["H"][0], w["Q"][0], np.zeros((10, 1)), np.zeros((1, 1))]
    outs = op.perform_filter(*inputs)
    gl = np.linspace(-1, 1, 25)
    grads = op.perform_filter_grad(*inputs, outs[1], outs[3], np.asarray(0.7), gl)
    _, _, gt = kt.logp_and_grads(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"],
                                 g_logp=np.array([0.7]), g_ll_obs=gl[None])
    for name, gr, x in zip(op.GRAD_INPUT_NAMES, grads, inputs[1:]):
        assert gr.shape == x.shape, name
        assert_allclose(gr.reshape(gt[name][0].shape), gt[name][0], rtol=1e-9, atol=1e-10,
                        err_msg=name)


def test_op_fused_logp_and_grad_batched_and_partial_outputs():
    """The bodies behind KalmanLogpOp (value + gradient in one device round trip), KalmanFilterBatchedOp (leading
    draw axis) and the `want` mask KalmanFilterOp.perform derives from its graph's clients."""
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op

    w = W.structural(B=5, n=40)
    y = w["y"].copy()
    y[[3, 17]] = np.nan
    c = 0.05 * np.arange(10.0)[:, None]
    d = np.array([[0.1]])
    one = [y[..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0], w["H"][0], w["Q"][0], c, d]
    outs = op.perform_logp_and_grad(*one)
    lp, _, gt = kt.logp_and_grads(y, w["a0"][:1], w["P0"][:1], w["T"][:1], w["Z"][:1], w["R"][:1], w["H"][:1],
                                  w["Q"][:1], c=c[None, :, 0], d=d[None, :, 0], g_logp=np.ones(1))
    assert np.shape(outs[0]) == () and abs(float(outs[0]) - float(lp[0])) < 1e-8
    for name, gr, x in zip(op.GRAD_INPUT_NAMES, outs[1:], one[1:]):
        assert gr.shape == x.shape, name
        assert_allclose(gr.reshape(gt[name][0].shape), gt[name][0], rtol=1e-9, atol=1e-10, err_msg=name)
    # only what is wanted is computed and copied
    full = op.perform_filter(*one)
    part = op.perform_filter(*one, want=(False, False, False, False, True, False))
    assert [o is None for o in part] == [True, True, True, True, False, True]
    assert float(part[4]) == float(full[4])
    part = op.perform_filter(*one, want=(False, True, False, True, True, False))
    assert part[0] is None and part[2] is None and part[5] is None
    assert np.array_equal(part[1], full[1]) and np.array_equal(part[3], full[3])
    # leading draw axis
    cB, dB = np.repeat(c[None], 5, 0), np.repeat(d[None], 5, 0)
    bat = op.perform_filter_batched(y[..., None], w["a0"][..., None], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"],
                                    cB, dB)
    ref = kn.kalman_filter(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], c=cB[..., 0], d=dB[..., 0])
    shapes = [(5, 40, 10, 1), (5, 41, 10, 1), (5, 40, 10, 10), (5, 41, 10, 10), (5,), (5, 40)]
    assert [np.shape(o) for o in bat] == shapes
    assert_allclose(bat[4], ref["logp"], rtol=0, atol=1e-8)
    assert_allclose(bat[2], ref["filtered_covs"], rtol=1e-9, atol=1e-9)
    assert_allclose(bat[1][..., 0], ref["predicted_states"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(bat[0][0], full[0]) and np.array_equal(bat[3][0], full[3])
    # a draw that is not positive definite raises LinAlgError like LAPACK inside the reference graph
    Hb = w["H"].copy()
    Hb[3] = -100.0
    with pytest.raises(np.linalg.LinAlgError, match="draw 3"):
        op.perform_filter_batched(y[..., None], w["a0"][..., None], w["P0"], w["T"], w["Z"], w["R"], Hb, w["Q"], cB, dB)


def test_op_bodies_with_time_varying_matrices():
    """Reference layouts with time on axis 0 (T (n,m,m), c (n,m,1)): the Op bodies pass the flags on and the gradient
    of a time-varying input has that input's shape."""
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op

    n, m, p, r = 15, 4, 1, 2
    w = W.random_dense(1, n, p, m, r, seed=3)
    wave = 0.8 + 0.2 * np.cos(np.arange(n))
    Tt = w["T"][0][None] * wave[:, None, None]
    Qt = w["Q"][0][None] * (1.5 - wave)[:, None, None]
    ct = 0.1 * np.sin(np.arange(n))[:, None, None] * np.ones((n, m, 1))
    d = np.array([[0.2]])
    one = [w["y"][..., None], w["a0"][0][:, None], w["P0"][0], Tt, w["Z"][0], w["R"][0], w["H"][0], Qt, ct, d]
    tv = ("T", "Q", "c")
    lp, ll, gt = kt.logp_and_grads(w["y"], w["a0"], w["P0"], Tt[None], w["Z"], w["R"], w["H"], Qt[None],
                                   c=ct[None, ..., 0], d=d[None, :, 0], g_logp=np.ones(1), time_varying=tv)
    outs = op.perform_logp_and_grad(*one)
    assert abs(float(outs[0]) - float(lp[0])) < 1e-8
    for name, gr, x in zip(op.GRAD_INPUT_NAMES, outs[1:], one[1:]):
        assert gr.shape == x.shape, name
        assert_allclose(gr.reshape(gt[name][0].shape), gt[name][0], rtol=1e-9, atol=1e-10, err_msg=name)
    full = op.perform_filter(*one)
    assert abs(float(full[4]) - float(lp[0])) < 1e-8 and full[0].shape == (n, m, 1)
    assert_allclose(full[5], ll[0], rtol=0, atol=1e-9)
    grads = op.perform_filter_grad(*one, full[1], full[3], np.asarray(1.0), np.zeros(n))
    for name, gr, g2 in zip(op.GRAD_INPUT_NAMES, grads, outs[1:]):
        assert_allclose(gr, g2, rtol=1e-12, atol=1e-12, err_msg=name)
    ss, sc = op.perform_smoother(Tt, w["R"][0], Qt, full[0], full[2])
    ra, rP = kn.kalman_smoother(Tt[None], w["R"], Qt[None], full[0][None, ..., 0], full[2][None],
                                time_varying=("T", "Q"))
    assert_allclose(ss[..., 0], ra[0], rtol=1e-8, atol=1e-8)
    assert_allclose(sc, rP[0], rtol=1e-8, atol=1e-8)


# ----------------------------------------------------------------------------------------------
# forward kernel families: LDS-resident generic vs register-resident row / row-split (p == 1)
# ----------------------------------------------------------------------------------------------
SPLITS = {10: (1, 2, 3, 5), 8: (1, 2, 4), 6: (1, 3), 4: (1, 2), 2: (1,)}


@pytest.mark.parametrize("m", range(1, 11))
@pytest.mark.parametrize("ft", FILTERS)
def test_row_kernels_match_oracle(eng, m, ft):
    B = 41  # not a multiple of the draws-per-wave of any variant: exercises the tail clones
    w = W.random_dense(B, 70, 1, m, max(1, m // 2), seed=100 + m)
    y = w["y"].copy()
    y[[3, 17, 69]] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact")
    eng.set_forward_kernel("row")
    try:
        outs = []
        for H in SPLITS.get(m, (1,)):
            eng.set_lanes_per_draw(m * H)
            out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact"))
            assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8, (m, H)
            for k in TAPES:
                assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"m={m} H={H} {k}")
            assert not out["info"].any()
            outs.append(out)
        # lanes-per-row is a pure scheduling knob: the split variants agree bit for bit
        for o in outs[2:]:
            for k in TAPES + ["logp"]:
                assert np.array_equal(o[k], outs[1][k]), k
    finally:
        eng.set_forward_kernel("auto")
        eng.set_lanes_per_draw(0)


def test_kernel_families_agree_at_headline_size(eng):
    import torch

    w = W.WORKLOADS["headline"]()
    dev = [eng.to_device(x) for x in _args(w)]
    res = {}
    try:
        for kern, lanes in (("lds", 0), ("row", 10), ("row", 30), ("mfma", 0), ("b4", 0), ("auto", 0)):
            eng.set_forward_kernel(kern)
            eng.set_lanes_per_draw(lanes)
            r = eng.forward(*dev, mvn_const="exact", store_filtered=False, store_predicted=True)
            torch.cuda.synchronize()
            res[(kern, lanes)] = (r.logp.clone(), r.predicted_covs[::512, -1].clone(), r.ll_obs[::512].clone())
    finally:
        eng.set_forward_kernel("auto")
        eng.set_lanes_per_draw(0)
    base = res[("lds", 0)]
    for key, val in res.items():
        assert (val[0] - base[0]).abs().max().item() < 1e-8, key
        assert torch.allclose(val[1], base[1], rtol=1e-10, atol=1e-12), key
        assert torch.allclose(val[2], base[2], rtol=1e-10, atol=1e-11), key


def test_row_kernel_not_positive_definite_reports_info(eng):
    w = W.reference_test_inputs(1, 2, 1, 5)
    H = np.array([[[-10.0]]])
    res = eng.forward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    assert int(res.info[0].item()) == ((0 + 1) << 8) | 1
    assert np.isnan(res.logp[0].item())


# ----------------------------------------------------------------------------------------------
# large state dimension: covariance in an HBM workspace, fp64-MFMA GEMMs (BASELINE config C5)
# ----------------------------------------------------------------------------------------------
BIG_CASES = {
    "dense_p2m64": lambda: W.random_dense(3, 10, 2, 64, 8, seed=2),
    "dense_p3m70": lambda: W.random_dense(2, 8, 3, 70, 5, seed=3),
    "dense_p1m130": lambda: W.random_dense(2, 6, 1, 130, 3, seed=4),
    "c5_daily_seasonal_m367": lambda: W.daily_seasonal(B=3, n=8),
}


@pytest.mark.parametrize("case", sorted(BIG_CASES))
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate"])
def test_big_m_forward_matches_oracle(eng, case, ft):
    w = BIG_CASES[case]()
    y = w["y"].copy()
    y[2] = np.nan
    assert eng.forward_kernel_name(w["B"], w["n"], w["p"], w["m"], w["r"]) == "big_gemm_kernel"
    out = _np(eng.forward(*_args(w, y), filter_type=ft, mvn_const="exact"))
    ref = co.kalman_filter(*_args(w, y), filter_type=ft, mvn_const="exact")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{case} {ft} {k}")
    assert np.all(out["ll_obs"][:, 2] == 0.0) and not out["info"].any()
    # covariances stay exactly symmetric through the MFMA path
    assert np.array_equal(out["predicted_covs"], np.swapaxes(out["predicted_covs"], -1, -2))


def test_big_m_needs_workspace_and_says_so(eng):
    import ctypes as C

    import torch

    from pymc_statespace_amd import _lib

    w = W.random_dense(1, 3, 1, 80, 2, seed=9)
    dims = _lib.KfDims(1, 3, 1, 80, 2, 0, 0)
    nbytes = C.c_size_t(0)
    assert eng.lib.kf_workspace_bytes(eng._h, C.byref(dims), C.byref(nbytes)) == 0
    assert nbytes.value >= 8 * 4 * 80 * 80
    d = [eng.to_device(x) for x in _args(w)]
    logp = torch.empty(1, device=d[1].device, dtype=torch.float64)
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = eng.lib.kf_filter_forward(eng._h, C.byref(dims), *[ptr(x) for x in d], None, None, ptr(logp),
                                   None, None, None, None, None, None, None, 0, None)
    assert rc == _lib.KF_ERR_WORKSPACE
    assert b"workspace" in eng.lib.kf_last_error(eng._h)


BIG_GRAD_CASES = {
    "dense_p1m64": (lambda: W.random_dense(3, 9, 1, 64, 8, seed=12), "standard"),
    "dense_p1m70_single": (lambda: W.random_dense(2, 7, 1, 70, 5, seed=15), "single"),
    "dense_p2m70": (lambda: W.random_dense(2, 7, 2, 70, 5, seed=13), "standard"),
    "dense_p3m66_univariate": (lambda: W.random_dense(2, 6, 3, 66, 4, seed=16), "univariate"),
    "c5_daily_seasonal_m367": (lambda: W.daily_seasonal(B=2, n=5), "standard"),
}


@pytest.mark.parametrize("case", sorted(BIG_GRAD_CASES))
def test_big_m_gradient_matches_autograd(eng, case):
    """kf_backward_big.hip: HBM workspace, three fp64-MFMA GEMMs per step, scalar observations."""
    from oracle import kalman_torch as kt

    mk, ft = BIG_GRAD_CASES[case]
    w = mk()
    B, n, m, p = w["B"], w["n"], w["m"], w["p"]
    y = w["y"].copy()
    y[2] = np.nan
    if p > 1:
        y[4, 0] = np.nan
    rng = np.random.default_rng(3)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft)
    res = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft)
    g = eng.backward(*_args(w, y), c=c, d=d, predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                     filter_type=ft, want_cd=True, h_diagonal=True)
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if k == "H" and p > 1:
            assert np.all(a[:, ~np.eye(p, dtype=bool)] == 0.0)
            a, b = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b, axis1=-2, axis2=-1)
        elif k in ("P0", "Q") and (p > 1 or ft == "univariate"):
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-9, (case, k)


def test_big_m_gradient_needs_scalar_observations_and_workspace(eng):
    import ctypes as C

    from pymc_statespace_amd import KalmanError, _lib

    w = W.random_dense(1, 4, 2, 70, 2, seed=9)
    res = eng.forward(*_args(w))
    with pytest.raises(KalmanError, match="scalar observations"):
        eng.backward(*_args(w), predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=np.ones(1))
    dims = _lib.KfDims(1, 4, 2, 70, 2, 0, _lib.KF_FLAG_H_DIAGONAL)
    nbytes = C.c_size_t(0)
    assert eng.lib.kf_workspace_bytes(eng._h, C.byref(dims), C.byref(nbytes)) == 0
    assert nbytes.value >= 8 * 7 * 70 * 70


BIG_SMOOTH_CASES = {
    "dense_p1m64": lambda: W.random_dense(3, 9, 1, 64, 8, seed=12),      # exactly one Cholesky block
    # r ~ m/2 keeps cond(P_hat) ~ 1e4: with r = 3..5 it reaches 1e9 and the reference's SVD pinv and
    # a Cholesky solve legitimately differ by cond x eps
    "dense_p2m70": lambda: W.random_dense(2, 7, 2, 70, 35, seed=13),     # 64 + 6
    "dense_p1m130": lambda: W.random_dense(2, 6, 1, 130, 65, seed=14),   # 64 + 64 + 2
    "c5_daily_seasonal_m367": lambda: W.daily_seasonal(B=2, n=6),        # 5 x 64 + 47
}


@pytest.mark.parametrize("case", sorted(BIG_SMOOTH_CASES))
def test_big_m_smoother_matches_oracle(eng, case):
    """kf_smoother_big.hip: HBM workspace, blocked Cholesky and substitutions as fp64-MFMA GEMMs."""
    w = BIG_SMOOTH_CASES[case]()
    flt = kn.kalman_filter(*_args(w))
    ref_a, ref_P = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    assert_allclose(sa, ref_a, rtol=1e-8, atol=1e-8, err_msg=case)
    assert_allclose(sP, ref_P, rtol=1e-8, atol=1e-8, err_msg=case)
    # reference tests/test_kalman_filter.py: the last smoothed moments are the last filtered ones
    assert np.array_equal(sa[:, -1], flt["filtered_states"][:, -1])
    assert np.array_equal(sP[:, -1], flt["filtered_covs"][:, -1])


def test_big_m_smoother_needs_workspace_and_says_so(eng):
    import ctypes as C

    import torch

    from pymc_statespace_amd import _lib

    w = W.random_dense(1, 3, 1, 80, 2, seed=9)
    flt = kn.kalman_filter(*_args(w))
    dims = _lib.KfDims(1, 3, 1, 80, 2, 0, 0)
    nbytes = C.c_size_t(0)
    assert eng.lib.kf_workspace_bytes(eng._h, C.byref(dims), C.byref(nbytes)) == 0
    assert nbytes.value >= 8 * 6 * 80 * 80
    d = [eng.to_device(x) for x in (w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])]
    sa, sP = torch.empty_like(d[3]), torch.empty_like(d[4])
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = eng.lib.kf_smoother(eng._h, C.byref(dims), *[ptr(x) for x in d], ptr(sa), ptr(sP), None, 0, None)
    assert rc == _lib.KF_ERR_WORKSPACE
    assert b"workspace" in eng.lib.kf_last_error(eng._h)


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("ft", FILTERS)
def test_lane_kernels_match_oracle(eng, m, ft):
    """One lane per draw (p == 1, m <= 4): BASELINE configs C1 / C2."""
    B = 130
    w = W.random_dense(B, 70, 1, m, max(1, m // 2), seed=200 + m)
    y = w["y"].copy()
    y[[0, 17, 69]] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact")
    # AUTO at large batches: the lane kernel for m <= 3, the packed row kernel for m = 4 (round 4: 1.5x faster on full tapes)
    assert eng.forward_kernel_name(32768, 70, 1, m, w["r"]) == ("kf_fwd_lane_kernel" if m <= 3 else "kf_fwd_row_kernel")
    eng.set_forward_kernel("lane")
    try:
        out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact"))
        # without the ll_obs tape the kernel batches its logarithms (mantissa products)
        lp_only = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", store_ll_obs=False,
                              store_filtered=False, store_predicted=False).logp.cpu().numpy()
    finally:
        eng.set_forward_kernel("auto")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    assert np.max(np.abs(lp_only - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"m={m} {k}")
    assert not out["info"].any()


@pytest.mark.parametrize("B", [5000, 17000], ids=["dpw16", "dpw64"])
@pytest.mark.parametrize("m", [1, 2, 3, 4])
def test_lane_kernel_wide_wavefronts_per_draw_y_and_tape_subsets(eng, B, m):
    """The lane kernel packs 4 / 16 / 64 draws per wavefront depending on the batch (B = 130 above: 4) and stages tapes
    and y through LDS blocks of 4 .. 16 steps: the wider packings, y per draw, a series length that is not a multiple
    of the block, the compile-time tape subsets (none / predicted only / all) and a run-time one (filtered only), and
    the not-positive-definite report."""
    n = 37
    w = W.random_dense(B, n, 1, m, 1, seed=300 + m)
    rng = np.random.default_rng(B + m)
    y = w["y"][None, :, :] + 0.1 * rng.standard_normal((B, n, 1))
    y[:, 5] = np.nan
    y[7, 11:14] = np.nan
    H = w["H"].copy()
    H[B - 3] = -50.0  # F < 0 at the first step of that draw
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    ref = kn.kalman_filter(*args, mvn_const="exact")
    eng.set_forward_kernel("lane")
    try:
        full = _np(eng.forward(*args, mvn_const="exact"))
        lo = eng.forward(*args, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
        pr = eng.forward(*args, mvn_const="exact", store_ll_obs=False, store_filtered=False)
        fo = eng.forward(*args, mvn_const="exact", store_predicted=False)
    finally:
        eng.set_forward_kernel("auto")
    ok = np.ones(B, bool)
    ok[B - 3] = False
    assert full["info"][B - 3] == ((1 << 8) | 1) and not full["info"][ok].any()
    assert np.max(np.abs(full["logp"][ok] - ref["logp"][ok])) < 1e-8
    for k in TAPES:
        assert_allclose(full[k][ok], ref[k][ok], rtol=1e-9, atol=1e-9, err_msg=f"m={m} {k}")
    assert np.array_equal(lo.logp.cpu().numpy()[ok], full["logp"][ok])
    assert np.array_equal(pr.logp.cpu().numpy()[ok], full["logp"][ok])
    assert np.array_equal(pr.predicted_covs.cpu().numpy()[ok], full["predicted_covs"][ok])
    assert np.array_equal(pr.predicted_states.cpu().numpy()[ok], full["predicted_states"][ok])
    assert np.array_equal(fo.filtered_covs.cpu().numpy()[ok], full["filtered_covs"][ok])
    assert np.array_equal(fo.filtered_states.cpu().numpy()[ok], full["filtered_states"][ok])
    assert np.array_equal(fo.ll_obs.cpu().numpy()[ok], full["ll_obs"][ok])


# ----------------------------------------------------------------------------------------------
# time-varying system matrices (reference tests/test_kalman_filter.py:102-156)
# ----------------------------------------------------------------------------------------------
def test_time_varying_copies_equal_static_reference_shapes():
    """The reference's only time-varying test: every matrix repeated n times (shapes only there;
    here the result must also equal the static run)."""
    from pymc_statespace_amd.filters import StandardFilter

    p, m, r, n = 1, 5, 2, 10
    w = W.reference_test_inputs(p, m, r, n)
    data, a0, P0 = w["y"][..., None], w["a0"][0][:, None], w["P0"][0]
    T, Z, R, H, Q = (w[k][0] for k in ("T", "Z", "R", "H", "Q"))
    rep = lambda x: np.concatenate([np.expand_dims(x, 0)] * n, axis=0)  # noqa: E731
    flt = StandardFilter()
    outs_tv = flt(data, a0, P0, rep(T), rep(Z), rep(R), rep(H), rep(Q))
    assert flt.seq_names == ["T", "Z", "R", "H", "Q"] and flt.non_seq_names == ["c", "d"]
    outs = StandardFilter()(data, a0, P0, T, Z, R, H, Q)
    shapes = [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)]
    for a, b, s in zip(outs_tv, outs, shapes):
        assert np.shape(a) == s
        assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    with pytest.raises(AssertionError, match="time dimension"):
        flt(data, a0, P0, rep(T)[:-1], Z, R, H, Q)


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate"])
@pytest.mark.parametrize("tv", [("T",), ("Q", "H"), ("T", "Z", "R", "H", "Q", "c", "d")])
def test_time_varying_matches_oracle(eng, ft, tv):
    B, n, p, m, r = 7, 30, 2, 5, 3
    w = W.random_dense(B, n, p, m, r, seed=77)
    rng = np.random.default_rng(3)
    mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"],
                c=0.1 * rng.standard_normal((B, m)), d=0.1 * rng.standard_normal((B, p)))
    for name in tv:
        base = mats[name]
        scale = 1.0 + 0.2 * np.sin(np.arange(n))[None, :, None] * (1.0 if name in ("c", "d") else 1.0)
        x = np.repeat(base[:, None], n, axis=1)
        mats[name] = x * (scale[..., None] if x.ndim == 4 else scale)
        if name == "T":
            mats[name] = np.repeat(base[:, None], n, axis=1) * (0.8 + 0.2 * np.cos(np.arange(n)))[None, :, None, None]
    y = w["y"].copy()
    y[4] = np.nan
    res = eng.forward(y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"],
                      mats["c"], mats["d"], filter_type=ft, mvn_const="exact", time_varying=tv)
    out = _np(res)
    ref = kn.kalman_filter_time_varying(y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"],
                                        mats["Q"], mats["c"], mats["d"], time_varying=tv,
                                        filter_type=ft, mvn_const="exact")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{ft} {tv} {k}")


@pytest.mark.parametrize("tv", [("T",), ("R", "Q"), ("T", "R", "Q")])
@pytest.mark.parametrize("m", [3, 5, 10, 16, 21])
def test_time_varying_smoother_matches_oracle(eng, tv, m):
    """time-varying T / R / Q in the RTS smoother (kalman_smoother.py:17-54, 62-77): the step that
    smooths t predicts from t with slice t.  Fed with the tapes of the time-varying forward filter."""
    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    B, n, r = 5, 40, max(1, m // 2)
    w = W.random_dense(B, n, 1, m, r, seed=500 + m)
    mats = dict(T=w["T"], R=w["R"], Q=w["Q"])
    for name in tv:
        wob = (0.8 + 0.2 * np.cos(0.7 * np.arange(n) + len(name)))[None, :, None, None]
        mats[name] = np.repeat(mats[name][:, None], n, axis=1) * wob
    fwd_tv = tuple(tv)
    ref_f = kn.kalman_filter_time_varying(w["y"], w["a0"], w["P0"], mats["T"], w["Z"], mats["R"], w["H"], mats["Q"],
                                          time_varying=fwd_tv)
    res = eng.forward(w["y"], w["a0"], w["P0"], mats["T"], w["Z"], mats["R"], w["H"], mats["Q"], time_varying=fwd_tv)
    assert_allclose(res.filtered_covs.cpu().numpy(), ref_f["filtered_covs"], rtol=1e-9, atol=1e-9)
    ra, rP = kn.kalman_smoother(mats["T"], mats["R"], mats["Q"], ref_f["filtered_states"], ref_f["filtered_covs"],
                                time_varying=tv)
    sa, sP = eng.smooth(mats["T"], mats["R"], mats["Q"], res.filtered_states, res.filtered_covs, time_varying=tv)
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)
    # pairing="reference" (KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING): slice t + 1 at the step that smooths t, what the reference's own
    # scan hands its step (kalman_smoother.py:62-77) -- a different smoother, reproduced on request
    ra2, rP2 = kn.kalman_smoother(mats["T"], mats["R"], mats["Q"], ref_f["filtered_states"], ref_f["filtered_covs"],
                                  time_varying=tv, pairing="reference")
    sa2, sP2 = eng.smooth(mats["T"], mats["R"], mats["Q"], res.filtered_states, res.filtered_covs, time_varying=tv,
                          pairing="reference")
    assert_allclose(sa2.cpu().numpy(), ra2, rtol=1e-8, atol=1e-8)
    assert_allclose(sP2.cpu().numpy(), rP2, rtol=1e-8, atol=1e-8)
    assert not np.allclose(rP2, rP, rtol=1e-6, atol=1e-9)
    # constant "time-varying" matrices reproduce the static smoother bit for bit
    const = {k: np.repeat(w[k][:, None], n, axis=1) for k in ("T", "R", "Q")}
    flt = kn.kalman_filter(*_args(w))
    s1 = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    eng.set_forward_kernel("lds")
    try:
        s0 = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    finally:
        eng.set_forward_kernel("auto")
    import os

    os.environ["KF_SMOOTHER_TV_MFMA"] = "0"   # (round 5: 5 <= m <= 16 would take the register-tile kernel with per-step slices)
    try:
        s2 = eng.smooth(const["T"], const["R"], const["Q"], flt["filtered_states"], flt["filtered_covs"],
                        time_varying=("T", "R", "Q"))
    finally:
        del os.environ["KF_SMOOTHER_TV_MFMA"]
    assert np.array_equal(s2[1].cpu().numpy(), s0[1].cpu().numpy())
    assert_allclose(s2[1].cpu().numpy(), s1[1].cpu().numpy(), rtol=1e-8, atol=1e-9)
    if 5 <= m <= 16:   # ... which reproduces the static register-tile kernel (KF_SMOOTHER_B4=0) bit for bit
        s3 = eng.smooth(const["T"], const["R"], const["Q"], flt["filtered_states"], flt["filtered_covs"],
                        time_varying=("T", "R", "Q"))
        os.environ["KF_SMOOTHER_B4"] = "0"
        try:
            s4 = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        finally:
            del os.environ["KF_SMOOTHER_B4"]
        assert np.array_equal(s3[0].cpu().numpy(), s4[0].cpu().numpy()) and np.array_equal(s3[1].cpu().numpy(), s4[1].cpu().numpy())
    # the reference-shaped eager class: a per-draw 3-D matrix is a sequence (filters/utilities.py:1-20)
    ks = KalmanSmoother()
    a1, P1 = ks(mats["T"][0], mats["R"][0], mats["Q"][0], ref_f["filtered_states"][0][..., None], ref_f["filtered_covs"][0])
    assert sorted(ks.seq_names) == sorted(tv)
    assert_allclose(P1, rP[0], rtol=1e-8, atol=1e-8)
    assert a1.shape == (n, m, 1) and StandardFilter is not None


def test_time_varying_smoother_needs_the_lds_kernel(eng):
    from pymc_statespace_amd import KalmanError

    w = W.random_dense(1, 4, 1, 70, 2, seed=5)
    flt = kn.kalman_filter(*_args(w))
    Tt = np.repeat(w["T"][:, None], 4, axis=1)
    with pytest.raises(KalmanError, match="time-varying"):
        eng.smooth(Tt, w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"], time_varying=("T",))


def _tv_mats(w, tv, n, rng):
    B, m, p = w["B"], w["m"], w["p"]
    mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"],
                c=0.1 * rng.standard_normal((B, m)), d=0.1 * rng.standard_normal((B, p)))
    for k, name in enumerate(tv):
        x = np.repeat(mats[name][:, None], n, axis=1)
        wave = (0.8 + 0.2 * np.cos(np.arange(n) + k))[None, :]
        mats[name] = x * wave.reshape(wave.shape + (1,) * (x.ndim - 2))
    return mats


TVG = [("T",), ("Z", "d"), ("R",), ("Q",), ("H", "c"), ("R", "Q"), ("T", "Z", "R", "H", "Q", "c", "d")]


@pytest.mark.parametrize("ft", ["standard", "univariate"])
@pytest.mark.parametrize("tv", TVG, ids=["".join(t) for t in TVG])
@pytest.mark.parametrize("pm", [(1, 4), (2, 5), (1, 10)], ids=lambda x: f"p{x[0]}m{x[1]}")
def test_time_varying_gradient_matches_autograd(eng, ft, tv, pm):
    """reverse-mode gradient with time-varying system matrices (LDS-resident adjoint).  The gradient
    of a flagged matrix is a (B,n,.,.) tape; unflagged ones are per draw.  Oracle: torch autograd through the same
    recursion with slice t at step t (kalman_filter.py:231-252)."""
    from oracle import kalman_torch as kt

    p, m = pm
    B, n, r = 3, 12, 2
    w = W.random_dense(B, n, p, m, r, seed=31 + m)
    rng = np.random.default_rng(17)
    mats = _tv_mats(w, tv, n, rng)
    y = w["y"].copy()
    y[5] = np.nan
    g_logp = rng.standard_normal(B)
    g_ll = 0.3 * rng.standard_normal((B, n))
    args = (y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"], mats["c"], mats["d"])
    _, _, gt = kt.logp_and_grads(*args, filter_type=ft, mvn_const="exact", g_logp=g_logp, g_ll_obs=g_ll,
                                 time_varying=tv)
    res = eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv)
    g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft, mvn_const="exact", want_cd=True, time_varying=tv)
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d"):
        a, b = g[k].cpu().numpy(), gt[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if ft == "univariate" and k in ("P0", "Q", "H"):
            # entrywise partials w.r.t. symmetric inputs differ between the recursions; their symmetric parts agree
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
            if k == "H":
                a, b = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b, axis1=-2, axis2=-1)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-9, (ft, tv, k)


def test_time_varying_gradient_of_repeated_matrices_sums_to_the_static_one(eng):
    """Every matrix repeated n times: the time-summed gradient tapes are the static gradients."""
    B, n, p, m, r = 4, 20, 1, 6, 2
    w = W.random_dense(B, n, p, m, r, seed=8)
    args = _args(w)
    res = eng.forward(*args)
    g0 = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                      g_logp=np.ones(B))
    tv = ("T", "Z", "R", "H", "Q")
    rep = {k: np.repeat(w[k][:, None], n, axis=1) for k in tv}
    args_tv = (w["y"], w["a0"], w["P0"], rep["T"], rep["Z"], rep["R"], rep["H"], rep["Q"])
    res_tv = eng.forward(*args_tv, time_varying=tv)
    assert_allclose(res_tv.logp.cpu().numpy(), res.logp.cpu().numpy(), rtol=1e-12)
    g1 = eng.backward(*args_tv, predicted_states=res_tv.predicted_states, predicted_covs=res_tv.predicted_covs,
                      g_logp=np.ones(B), time_varying=tv)
    for k in ("a0", "P0"):
        assert_allclose(g1[k].cpu().numpy(), g0[k].cpu().numpy(), rtol=1e-9, atol=1e-9)
    for k in tv:
        assert g1[k].shape[:2] == (B, n)
        assert_allclose(g1[k].sum(1).cpu().numpy(), g0[k].cpu().numpy(), rtol=1e-8, atol=1e-9, err_msg=k)


def test_time_varying_gradient_needs_the_lds_kernel(eng):
    from pymc_statespace_amd import KalmanError

    n = 4
    w = W.random_dense(1, n, 1, 70, 2, seed=5)
    Tt = np.repeat(w["T"][:, None], n, axis=1)
    flt = kn.kalman_filter(*_args(w))
    with pytest.raises(KalmanError, match="time-varying"):
        eng.backward(w["y"], w["a0"], w["P0"], Tt, w["Z"], w["R"], w["H"], w["Q"],
                     predicted_states=flt["predicted_states"], predicted_covs=flt["predicted_covs"],
                     g_logp=np.ones(1), time_varying=("T",))


@pytest.mark.parametrize("ft", ["standard", "univariate", "single"])
@pytest.mark.parametrize("m", list(range(1, 15)))
def test_b4_adjoint_matches_autograd(eng, m, ft, monkeypatch):
    """kf_backward_b4.hip: the adjoint of the augmented recursion (p = 1, m <= 14) -- sweep down to step 1 on block MFMAs,
    step 0 handed to kf_backward_mfma.hip for the exact entrywise gradient of P0.  Oracle: torch autograd of the
    reference recursion.  Missing observations (first, last, block boundary), y per draw, c / d, a cotangent on every
    ll_obs term, and series of 1, 2, 3 and 70 steps."""
    from oracle import kalman_torch as kt

    monkeypatch.setenv("KF_BACKWARD_B4", "1")  # (also below the batch / m thresholds of the AUTO choice)
    for n in (70, 3, 2, 1):
        B, r = 3, max(1, m // 2)
        w = W.random_dense(B, n, 1, m, r, seed=50 + m)
        rng = np.random.default_rng(m + n)
        y = rng.standard_normal((B, n, 1))  # (the workload standardises its y: undefined for n = 1)
        if n == 70:
            y[:, [0, 5, 63, 64, 69]] = np.nan
            y[1, 20:23] = np.nan
        c = 0.1 * rng.standard_normal((B, m))
        d = 0.1 * rng.standard_normal((B, 1))
        gl = rng.standard_normal(B)
        gll = 0.3 * rng.standard_normal((B, n))
        args = (y,) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")) + (c, d)
        grads = []
        for b in range(B):  # the torch oracle shares y between draws
            one = (y[b],) + tuple(x[b:b + 1] for x in args[1:])
            grads.append(kt.logp_and_grads(*one, filter_type=ft, g_logp=gl[b:b + 1], g_ll_obs=gll[b:b + 1])[2])
        res = eng.forward(*args, filter_type=ft)
        g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl,
                         g_ll_obs=gll, filter_type=ft, want_cd=True)
        for k in ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d"):
            a_ = g[k].cpu().numpy()
            b_ = np.concatenate([gr[k] for gr in grads], axis=0)
            if ft == "univariate" and k in ("P0", "Q", "H"):
                a_, b_ = a_ + np.swapaxes(a_, -1, -2), b_ + np.swapaxes(b_, -1, -2)
            assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-10, (m, ft, n, k)


def test_b4_adjoint_equals_the_tile_adjoint_at_the_headline_shape(eng, monkeypatch):
    """4096 draws x 1000 steps x m = 10 (north star): both adjoint kernels on the same tapes."""
    w = W.WORKLOADS["headline"]()
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev, store_ll_obs=False, store_filtered=False)
    ones = np.ones(w["B"])
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("KF_BACKWARD_B4", flag)
        out[flag] = eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                                 g_logp=ones, want_cd=True)
    for k in out["1"]:
        a_, b_ = out["1"][k].cpu().numpy(), out["0"][k].cpu().numpy()
        assert np.isfinite(a_).all()
        assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-9, k


@pytest.mark.parametrize("case", ["headline_m10", "c2_m2", "varmax_p2_m6", "dense_m20"])
def test_value_and_grad_single_call(eng, case):
    """kf_filter_value_and_grad: logp, info and the nine gradients from one C-ABI call with the tapes inside the
    workspace -- bit for bit what kf_filter_forward + kf_filter_backward give, and the oracle's numbers."""
    from oracle import kalman_torch as kt

    w = {"headline_m10": lambda: W.structural(B=37, n=80), "c2_m2": lambda: W.arma11(B=70, n=50),
         "varmax_p2_m6": lambda: W.random_dense(9, 40, 2, 6, 3, seed=2),
         "dense_m20": lambda: W.random_dense(5, 30, 1, 20, 4, seed=3)}[case]()
    B = w["B"]
    rng = np.random.default_rng(1)
    y = w["y"].copy()
    y[3] = np.nan
    c, d = 0.1 * rng.standard_normal((B, w["m"])), 0.1 * rng.standard_normal((B, w["p"]))
    gl = rng.standard_normal(B)
    args = (y,) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")) + (c, d)
    logp, info, g = eng.value_and_grad(*args, g_logp=gl, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact")
    g2 = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl,
                      mvn_const="exact", want_cd=True)
    # (without the ll_obs tape some kernels sum the log terms in another order: last-bit differences in logp)
    assert_allclose(logp.cpu().numpy(), res.logp.cpu().numpy(), rtol=1e-13, atol=0)
    assert not info.cpu().numpy().any()
    # (same numbers up to rounding, not bit for bit: the one-call path hands the block kernels' own tile tape from the
    #  forward to the adjoint, and for p > 1 forward() may pick another kernel after detecting a diagonal H)
    for k in g:
        a_, b_ = g[k].cpu().numpy(), g2[k].cpu().numpy()
        if k in ("P0", "H") and w["p"] > 1:
            # round 5: p > 1 without KF_FLAG_H_DIAGONAL takes the whitened path, whose scalar-observation adjoints return the
            # SYMMETRIC part of the entrywise g_P0 and whose map back through L gives a symmetric g_H = L^-T sym(G*) L^-1
            # (include/kf.h) -- all a symmetric P0 / H can see; eng.backward ran the dense adjoint (entrywise gradients)
            a_, b_ = 0.5 * (a_ + np.swapaxes(a_, 1, 2)), 0.5 * (b_ + np.swapaxes(b_, 1, 2))
        assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-10, k
    rl, _, gt = kt.logp_and_grads(*args, mvn_const="exact", g_logp=gl)
    assert np.max(np.abs(logp.cpu().numpy() - rl)) < 1e-8
    for k in g:
        a_, b_ = g[k].cpu().numpy(), gt[k]
        if k in ("P0", "H") and w["p"] > 1:
            a_, b_ = 0.5 * (a_ + np.swapaxes(a_, 1, 2)), 0.5 * (b_ + np.swapaxes(b_, 1, 2))
        assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-9, k
    # default cotangent: ones
    _, _, g1 = eng.value_and_grad(*args, mvn_const="exact")
    g3 = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                      g_logp=np.ones(B), mvn_const="exact", want_cd=True)
    assert_allclose(g1["T"].cpu().numpy(), g3["T"].cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert_allclose(g1["Q"].cpu().numpy(), g3["Q"].cpu().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("ft", ["standard", "univariate", "single"])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8, 10, 11, 12, 14])
def test_value_and_grad_through_the_tile_tape(eng, m, ft):
    """One call, p = 1, m <= 14: where AUTO puts both passes on the 4x4x4-block kernels the forward leaves its own tile
    tape in the workspace and the adjoint reads it back (m <= 3 at this batch size: lane kernel + compact tapes -- the
    same entry point).  Oracle: torch autograd.  y per draw, missing steps, c / d, series of 70, 2 and 1 steps."""
    from oracle import kalman_torch as kt

    for n in (70, 2, 1):
        B, r = 5, max(1, m // 2)
        w = W.random_dense(B, n, 1, m, r, seed=70 + m)
        rng = np.random.default_rng(m + n)
        y = rng.standard_normal((B, n, 1))
        if n == 70:
            y[:, [0, 31, 32, 69]] = np.nan
            y[2, 40:43] = np.nan
        c = 0.1 * rng.standard_normal((B, m))
        d = 0.1 * rng.standard_normal((B, 1))
        gl = rng.standard_normal(B)
        args = (y,) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")) + (c, d)
        logp, info, g = eng.value_and_grad(*args, g_logp=gl, filter_type=ft, mvn_const="exact")
        assert not info.cpu().numpy().any()
        for b in range(B):
            one = (y[b],) + tuple(x[b:b + 1] for x in args[1:])
            rl, _, gt = kt.logp_and_grads(*one, filter_type=ft, mvn_const="exact", g_logp=gl[b:b + 1])
            assert abs(float(logp[b]) - float(rl[0])) < 1e-8
            for k in g:
                a_, b_ = g[k][b:b + 1].cpu().numpy(), gt[k]
                if ft == "univariate" and k in ("P0", "Q", "H"):
                    a_, b_ = a_ + np.swapaxes(a_, -1, -2), b_ + np.swapaxes(b_, -1, -2)
                assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-10, (m, ft, n, b, k)


def torch_equal(a, b):
    import torch

    return bool(torch.equal(a, b))


BWD_SPLITS = [(10, 1), (10, 2), (10, 3), (10, 5), (8, 1), (8, 2), (8, 4), (6, 1), (6, 3), (4, 1), (4, 2), (2, 1)]


@pytest.mark.parametrize("mh", BWD_SPLITS, ids=[f"m{m}h{h}" for m, h in BWD_SPLITS])
@pytest.mark.parametrize("ft", ["standard", "single", "univariate"])
def test_register_resident_gradient_kernels_match_autograd(eng, mh, ft):
    """kf_bwd_rowsplit_kernel (p == 1, even m <= 10), every compiled (m, lanes-per-row)."""
    from oracle import kalman_torch as kt

    m, H = mh
    B, n = 13, 25  # 13: exercises the tail clones of every draws-per-wave
    w = W.random_dense(B, n, 1, m, max(1, m // 2), seed=300 + m)
    y = w["y"].copy()
    y[2] = np.nan
    y[-1] = np.nan
    rng = np.random.default_rng(m * 10 + H)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft)
    eng.set_lanes_per_draw(m * H)
    try:
        res = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft)
        g = eng.backward(*_args(w, y), c=c, d=d, predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                         filter_type=ft, want_cd=True)
    finally:
        eng.set_lanes_per_draw(0)
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if ft == "univariate" and k in ("P0", "Q", "H"):
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-10, k


def test_gradient_kernel_families_agree_at_headline_size(eng):
    import torch

    w = W.WORKLOADS["headline"]()
    dev = [eng.to_device(x) for x in _args(w)]
    res = eng.forward(*dev, store_filtered=False, store_ll_obs=False)
    ones = torch.ones(w["B"], device=dev[1].device, dtype=torch.float64)
    out = {}
    try:
        for kern in ("lds", "auto"):
            eng.set_forward_kernel(kern)
            out[kern] = eng.backward(*dev, predicted_states=res.predicted_states,
                                     predicted_covs=res.predicted_covs, g_logp=ones)
    finally:
        eng.set_forward_kernel("auto")
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
        scale = 1.0 + out["lds"][k].abs().max().item()
        assert (out["lds"][k] - out["auto"][k]).abs().max().item() / scale < 1e-10, k


# ----------------------------------------------------------------------------------------------
# multivariate row kernel (kf_forward_rowsplit_mp.hip): p > 1 one observation at a time
# (BASELINE config C4: VARMAX, p = 5, m = 15 / 10)
# ----------------------------------------------------------------------------------------------
MP_SHAPES = [(5, 15, 5), (5, 10, 5), (2, 16, 3), (3, 13, 4), (8, 12, 6), (4, 7, 3), (2, 6, 2),
             (3, 3, 2), (2, 2, 1), (1, 12, 4), (1, 15, 3)]
MP_H = {16: (2, 4), 14: (2, 4), 12: (2, 3), 10: (2, 3), 8: (2, 4), 6: (1, 3), 4: (1, 2)}


@pytest.mark.parametrize("pmr", MP_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in MP_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("mvn", ["reference", "exact"])
def test_multivariate_row_kernels_match_oracle(eng, pmr, ft, mvn):
    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    if p == 1 and m <= 10:
        pytest.skip("p == 1, m <= 10 runs the dedicated row kernels")
    B = 23
    w = W.random_dense(B, 75, p, m, r, seed=300 + 17 * p + m)
    assert h_is_diagonal(w["H"])
    y = w["y"].copy()
    y[[3, 17, 74]] = np.nan              # whole rows
    if p > 1:
        y[5, 0] = np.nan                  # single cells: row deletion
        y[64, 1:] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    # partial rows: the reference counts log(2 pi) for the full p there, "exact" counts n_observed
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                           cholesky_triangle="correct", partial_missing="delete")
    M = max(4, m + (m & 1))
    eng.set_forward_kernel("row")
    try:
        outs = []
        for H in MP_H[M]:
            eng.set_lanes_per_draw(M * H)
            assert eng.forward_kernel_name(B, 75, p, m, r, ft, True) == "kf_fwd_rowsplit_mp_kernel"
            out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
            assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8, (M, H)
            for k in TAPES:
                assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"M={M} H={H} {k}")
            assert not out["info"].any()
            assert np.all(out["ll_obs"][:, [3, 17, 74]] == 0.0)
            outs.append(out)
        for k in TAPES + ["logp"]:  # lanes-per-row only changes the order of the row sums
            assert_allclose(outs[0][k], outs[1][k], rtol=1e-11, atol=1e-11, err_msg=k)
    finally:
        eng.set_forward_kernel("auto")
        eng.set_lanes_per_draw(0)


def test_multivariate_row_kernel_needs_diagonal_h(eng):
    """Joint filters with a dense H are whitened onto the diagonal-H kernel of the shape (auto) / are refused (forced row)."""
    w = W.random_dense(5, 20, 3, 8, 3, seed=5, h_diag=False)
    assert eng.forward_kernel_name(5, 20, 3, 8, 3, "standard", False) == "kf_fwd_b4q_kernel"   # behind kf_whiten_kernel
    assert eng.forward_kernel_name(5, 20, 3, 8, 3, "univariate", False) == "kf_fwd_b4q_kernel"
    assert eng.forward_kernel_name(16384, 20, 3, 8, 3, "univariate", False) == "kf_fwd_rowsplit_mp_kernel"
    out = _np(eng.forward(*_args(w)))  # auto-detects the dense H
    ref = kn.kalman_filter(*_args(w))
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    eng.set_forward_kernel("row")
    try:
        with pytest.raises(RuntimeError, match="KF_FLAG_H_DIAGONAL"):
            eng.forward(*_args(w), h_diagonal=False)
    finally:
        eng.set_forward_kernel("auto")


def test_multivariate_row_kernel_per_draw_data_and_partial_outputs(eng):
    w = W.varmax(B=37, n=130, k=3, order=(1, 1))
    rng = np.random.default_rng(3)
    yb = w["y"][None] + 0.1 * rng.standard_normal((37,) + w["y"].shape)
    eng.set_forward_kernel("row")
    try:
        name = eng.forward_kernel_name(37, 130, w["p"], w["m"], w["r"], "standard", True)
        assert name == "kf_fwd_rowsplit_mp_kernel"
        res = eng.forward(*_args(w, yb), store_filtered=False, store_ll_obs=False)
    finally:
        eng.set_forward_kernel("auto")
    assert res.filtered_covs is None and res.ll_obs is None
    ref = kn.kalman_filter(*_args(w, yb))
    assert np.max(np.abs(res.logp.cpu().numpy() - ref["logp"])) < 1e-8
    assert_allclose(res.predicted_covs.cpu().numpy(), ref["predicted_covs"], rtol=1e-9, atol=1e-9)
    assert_allclose(res.predicted_states.cpu().numpy(), ref["predicted_states"], rtol=1e-9, atol=1e-9)


def test_multivariate_row_kernel_not_positive_definite_reports_info(eng):
    w = W.random_dense(4, 12, 3, 6, 2, seed=9)
    H = w["H"].copy()
    H[2, 1, 1] = -50.0
    res = eng.forward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    info = res.info.cpu().numpy()
    assert info[2] == ((0 + 1) << 8) | 2 and not info[[0, 1, 3]].any()
    lp = res.logp.cpu().numpy()
    assert np.isnan(lp[2]) and np.isfinite(lp[[0, 1, 3]]).all()


def test_kernel_families_agree_at_c4_size(eng):
    import torch

    w = W.varmax(B=512, n=1000)
    dev = [eng.to_device(x) for x in _args(w)]
    res = {}
    try:
        for kern in ("lds", "row"):
            eng.set_forward_kernel(kern)
            r = eng.forward(*dev, mvn_const="exact", store_filtered=False, h_diagonal=True)
            torch.cuda.synchronize()
            res[kern] = (r.logp.clone(), r.predicted_covs[::64, -1].clone(), r.ll_obs[::64].clone())
    finally:
        eng.set_forward_kernel("auto")
    assert (res["row"][0] - res["lds"][0]).abs().max().item() < 1e-8
    assert torch.allclose(res["row"][1], res["lds"][1], rtol=1e-9, atol=1e-11)
    assert torch.allclose(res["row"][2], res["lds"][2], rtol=1e-9, atol=1e-10)


# ----------------------------------------------------------------------------------------------
# accumulator-layout kernel (kf_forward_mfma.hip): P in v_mfma_f64_16x16x4 layout, one wave per draw
# ----------------------------------------------------------------------------------------------
MFMA_SHAPES = [(1, m, max(1, m // 2)) for m in (1, 2, 3, 5, 8, 10, 11, 13, 16)] + \
              [(5, 15, 5), (5, 10, 5), (2, 16, 3), (3, 13, 4), (8, 12, 6), (4, 7, 3), (3, 3, 2)]


@pytest.mark.parametrize("pmr", MFMA_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in MFMA_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("mvn", ["reference", "exact"])
def test_mfma_kernel_matches_oracle(eng, pmr, ft, mvn):
    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    B = 19
    w = W.random_dense(B, 139, p, m, r, seed=500 + 17 * p + m)
    y = w["y"].copy()
    y[[3, 17, 64, 138]] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[70, 1:] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                           cholesky_triangle="correct", partial_missing="delete")
    eng.set_forward_kernel("mfma")
    try:
        out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
    finally:
        eng.set_forward_kernel("auto")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert not out["info"].any()
    assert np.all(out["ll_obs"][:, [3, 17, 64, 138]] == 0.0)
    # the covariance tapes are bitwise symmetric by construction
    assert np.array_equal(out["predicted_covs"][:, 1:], np.swapaxes(out["predicted_covs"][:, 1:], -1, -2))
    assert np.array_equal(out["filtered_covs"], np.swapaxes(out["filtered_covs"], -1, -2))


def test_mfma_kernel_per_draw_data_not_pd_and_partial_outputs(eng):
    w = W.varmax(B=21, n=130, k=3, order=(1, 1))
    rng = np.random.default_rng(3)
    yb = w["y"][None] + 0.1 * rng.standard_normal((21,) + w["y"].shape)
    H = w["H"].copy()
    H[7, 1, 1] = -50.0
    eng.set_forward_kernel("mfma")
    try:
        res = eng.forward(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"],
                          store_filtered=False, store_ll_obs=False, h_diagonal=True)
    finally:
        eng.set_forward_kernel("auto")
    assert res.filtered_covs is None and res.ll_obs is None
    info = res.info.cpu().numpy()
    good = np.arange(21) != 7
    assert info[7] == ((0 + 1) << 8) | 2 and not info[good].any()
    ref = kn.kalman_filter(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    lp = res.logp.cpu().numpy()
    assert np.isnan(lp[7]) and np.max(np.abs(lp[good] - ref["logp"][good])) < 1e-8
    assert_allclose(res.predicted_covs.cpu().numpy()[good], ref["predicted_covs"][good], rtol=1e-9, atol=1e-9)


def test_auto_kernel_selection(eng):
    """The documented AUTO rules (include/kf.h) -- names only, no launch."""
    name = eng.forward_kernel_name
    assert name(4096, 1000, 1, 2, 1) == "kf_fwd_lane_kernel"        # C2
    assert name(512, 1000, 1, 2, 1) == "kf_fwd_lane_kernel"         # C2, one of 8 GPUs' share: 4 draws per wavefront
    assert name(512, 1000, 1, 3, 1) == "kf_fwd_b4_kernel"
    assert name(4096, 1000, 1, 6, 3) == "kf_fwd_b4_kernel"
    assert name(16384, 1000, 1, 6, 3) == "kf_fwd_row_kernel"
    assert name(65536, 1000, 1, 6, 3) == "kf_fwd_row_kernel"
    assert name(4096, 1000, 1, 8, 3) == "kf_fwd_b4_kernel"
    assert name(65536, 1000, 1, 8, 3) == "kf_fwd_row_kernel"
    assert name(4096, 1000, 1, 10, 5) == "kf_fwd_b4_kernel"         # north star
    assert name(8192, 2000, 1, 10, 4, "cholesky") == "kf_fwd_b4_kernel"  # C3
    assert name(65536, 1000, 1, 10, 5) == "kf_fwd_b4_kernel"
    assert name(4096, 1000, 1, 14, 5) == "kf_fwd_b4_kernel"
    assert name(4096, 1000, 1, 15, 5) == "kf_fwd_mfma_kernel"       # m + 2 > 16: no room for the mean carrier
    assert name(4096, 1000, 1, 16, 5) == "kf_fwd_mfma_kernel"
    assert name(4096, 1000, 1, 17, 5) == "kf_fwd_mfma_nb_kernel"
    assert name(4096, 1000, 1, 48, 5) == "kf_fwd_mfma_nb_kernel"
    assert name(4096, 1000, 3, 30, 5, "standard", True) == "kf_fwd_mfma_nb_kernel"
    assert name(4096, 1000, 3, 30, 5, "standard", False) == "kf_fwd_mfma_nb_kernel"    # dense H: whitened (kf_forward_whiten.hip)
    assert name(4096, 1000, 1, 49, 5) == "kf_fwd_mfma_wg_kernel"
    assert name(4096, 1000, 1, 64, 5) == "kf_fwd_mfma_wg_kernel"
    assert name(4096, 1000, 3, 60, 5, "standard", False) == "kf_fwd_mfma_wg_kernel"   # dense H: whitened
    assert name(4096, 1000, 9, 60, 5, "standard", False) == "big_gemm_kernel"         # dense H, p > 8: no whitening
    assert name(4096, 1000, 1, 65, 5) == "big_gemm_kernel"
    assert name(16384, 1000, 5, 15, 5, "standard", True) == "kf_fwd_b4q_kernel"    # C4
    assert name(16384, 1000, 5, 10, 5, "standard", True) == "kf_fwd_b4q_kernel"
    assert name(16384, 1000, 5, 15, 5, "standard", False) == "kf_fwd_b4q_kernel"   # dense H: whitened (C4 with a dense H)
    assert name(16384, 1000, 5, 15, 5, "univariate", False) == "kf_fwd_b4q_kernel"
    assert name(16384, 1000, 3, 8, 3, "standard", True) == "kf_fwd_rowsplit_mp_kernel"  # m <= 8: packed rows once the batch fills the chip
    assert name(1024, 1000, 3, 8, 3, "standard", True) == "kf_fwd_b4q_kernel"
    assert name(4096, 1000, 2, 16, 3, "standard", True) == "kf_fwd_mfma_kernel"    # m + 1 > 16
    assert name(4096, 1000, 9, 12, 3, "standard", True) == "kf_fwd_lds_kernel"     # p > 8
    assert name(128, 1000, 1, 367, 3) == "big_gemm_kernel"          # C5


# ----------------------------------------------------------------------------------------------
# 4x4x4-block kernel (kf_forward_b4.hip): v_mfma_f64_4x4x4_4b_f64, mean carried in the tile
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", list(range(1, 15)))
@pytest.mark.parametrize("ft", FILTERS)
def test_b4_kernel_matches_oracle(eng, m, ft):
    """Every block count (NB = 1..4), streamed (m <= 10) and per-step (m >= 11) covariance tapes, intercepts,
    missing rows, both log(2 pi) conventions, a batch that is not a multiple of anything."""
    B = 19
    w = W.random_dense(B, 77, 1, m, max(1, min(m, 3)), seed=300 + m)
    y = w["y"].copy()
    y[[0, 3, 31, 32, 64, 76]] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    eng.set_forward_kernel("b4")
    try:
        for mvn in ("reference", "exact"):
            ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn)
            out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
            assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8, (m, mvn)
            for k in TAPES:
                assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"m={m} {mvn} {k}")
            assert not out["info"].any()
        lp_only = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", store_ll_obs=False,
                              store_filtered=False, store_predicted=False).logp.cpu().numpy()
        assert np.max(np.abs(lp_only - ref["logp"])) < 1e-8
    finally:
        eng.set_forward_kernel("auto")


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 65, 129])
@pytest.mark.parametrize("m", [2, 7, 10, 12])
def test_b4_kernel_block_boundaries_and_partial_outputs(eng, n, m):
    """Series lengths around the 32-step y block / likelihood ring and the 128-double tape windows; every subset of
    tapes; tapes that start 8 bytes off a 16-byte boundary (the streams' single-element head / tail path)."""
    import torch

    from pymc_statespace_amd.engine import FilterResult

    B = 6
    w = W.random_dense(B, n, 1, m, 2, seed=17 * n + m)
    ref = kn.kalman_filter(*_args(w), mvn_const="exact")
    eng.set_forward_kernel("b4")
    try:
        for sel in ((True, True, True), (False, True, False), (False, False, True), (True, False, False)):
            r = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=sel[0], store_filtered=sel[1],
                            store_predicted=sel[2])
            assert np.max(np.abs(r.logp.cpu().numpy() - ref["logp"])) < 1e-8
            for k in TAPES:
                v = getattr(r, k)
                if v is not None:
                    assert_allclose(v.cpu().numpy(), ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{sel} {k}")

        def odd(*shape):  # 8-byte aligned only, guard words around
            flat = torch.full((int(np.prod(shape)) + 4,), float("nan"), device="cuda", dtype=torch.float64)
            return flat, flat[1:1 + int(np.prod(shape))].view(*shape)

        bufs = {k: odd(*s) for k, s in dict(ll_obs=(B, n), filtered_states=(B, n, m), predicted_states=(B, n + 1, m),
                                            filtered_covs=(B, n, m, m), predicted_covs=(B, n + 1, m, m)).items()}
        out = FilterResult(logp=torch.empty(B, device="cuda", dtype=torch.float64),
                           info=torch.empty(B, device="cuda", dtype=torch.int32), **{k: v[1] for k, v in bufs.items()})
        eng.forward(*_args(w), mvn_const="exact", out=out)
        for k, (flat, view) in bufs.items():
            assert_allclose(view.cpu().numpy(), ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
            g = flat.cpu().numpy()
            assert np.isnan(g[0]) and np.isnan(g[-3:]).all(), f"{k}: wrote outside the tape"
    finally:
        eng.set_forward_kernel("auto")


def test_b4_kernel_per_draw_data_not_pd_and_univariate_skip(eng):
    """y per draw; a draw whose F turns non-positive reports info = (t + 1) << 8 | 1 and NaN; the Univariate filter
    skips F == 0 (:468-471) where Standard flags it."""
    B, n, m = 9, 40, 5
    w = W.random_dense(B, n, 1, m, 2, seed=77)
    yb = np.repeat(w["y"][None], B, 0) + 0.01 * np.arange(B)[:, None, None]
    H = w["H"].copy()
    H[4] = -50.0
    eng.set_forward_kernel("b4")
    try:
        res = eng.forward(yb, *_args(w)[1:6], H, w["Q"], mvn_const="exact")
        info = res.info.cpu().numpy()
        assert info[4] == ((0 + 1) << 8 | 1) and not np.delete(info, 4).any()
        good = np.arange(B) != 4
        ref = kn.kalman_filter(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], mvn_const="exact")
        lp = res.logp.cpu().numpy()
        assert np.isnan(lp[4]) and np.max(np.abs(lp[good] - ref["logp"][good])) < 1e-8
        # exactly known state and no noise: F == 0 at every step
        z = W.reference_test_inputs(1, 3, 1, 12)
        P0 = np.zeros_like(z["P0"])
        Hz, Qz = np.zeros((1, 1, 1)), np.zeros((1, 1, 1))
        ru = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate")
        rf = kn.kalman_filter(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate")
        assert not ru.info.cpu().numpy().any()
        assert_allclose(ru.logp.cpu().numpy(), rf["logp"], atol=1e-12)
        assert_allclose(ru.filtered_states.cpu().numpy(), rf["filtered_states"], atol=1e-12)
        rs = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="standard")
        assert rs.info.cpu().numpy()[0] == ((0 + 1) << 8 | 1)
    finally:
        eng.set_forward_kernel("auto")


def test_b4_kernel_steady_state_phase_matches_exact_recursion(eng, monkeypatch):
    """KF_FLAG_STEADY_STATE (exact_recursion=False; the default from 1536 steps on): once the predicted covariance has stopped
    moving (and no observation is missing ahead) the block kernel advances only the mean with the converged gain;
    KF_FLAG_EXACT_RECURSION keeps the step-by-step recursion.  Both against the oracle; a series with a late missing row
    (steady state entered only behind it) and intercepts.  A 24-draw batch runs on the pair kernel (<= 1024 draws with
    covariance tapes: always the recursion at every step), so the fused kernel is forced here."""
    w = W.structural(B=24, n=600)
    B, m = 24, w["m"]
    rng = np.random.default_rng(1)
    c = 0.05 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    monkeypatch.setenv("KF_FORWARD_B4P", "0")  # (else the pair kernel takes a batch this small over at the first hand-over point)
    for missing in ((), (5, 100, 250)):
        y = w["y"].copy()
        y[list(missing)] = np.nan
        for ft in ("standard", "univariate", "single"):
            ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact")
            outs = {}
            for exact in (True, False):
                out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", exact_recursion=exact))
                assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
                for k in TAPES:
                    assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{ft} exact={exact} {k}")
                assert not out["info"].any()
                outs[exact] = out
            # the switch did switch (some draw of this workload converges within 600 steps), and changes nothing beyond rounding
            assert not np.array_equal(outs[True]["predicted_states"], outs[False]["predicted_states"])
            assert_allclose(outs[True]["filtered_covs"], outs[False]["filtered_covs"], rtol=1e-11, atol=1e-12)
            assert_allclose(outs[True]["logp"], outs[False]["logp"], rtol=1e-12)
    # partial outputs in the steady phase: every tape subset, against the full run
    full = _np(eng.forward(*_args(w), mvn_const="exact", exact_recursion=False))
    for sel in ((False, True, False), (False, False, True), (True, False, False), (False, False, False)):
        r = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=sel[0], store_filtered=sel[1], store_predicted=sel[2],
                        exact_recursion=False)
        assert_allclose(r.logp.cpu().numpy(), full["logp"], rtol=1e-13)
        for k in TAPES:
            v = getattr(r, k)
            if v is not None:
                assert_allclose(v.cpu().numpy(), full[k], rtol=1e-12, atol=1e-13, err_msg=f"{sel} {k}")


# ----------------------------------------------------------------------------------------------
# multi-observation block kernel (kf_forward_b4q.hip): p scalar observations per step on the 4x4x4 block tiles
# ----------------------------------------------------------------------------------------------
B4Q_SHAPES = [(5, 15, 5), (5, 10, 5), (2, 15, 3), (3, 13, 4), (8, 12, 6), (4, 7, 3), (3, 3, 2), (2, 1, 1), (2, 11, 4), (6, 2, 2)]


@pytest.mark.parametrize("pmr", B4Q_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in B4Q_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate"])
@pytest.mark.parametrize("mvn", ["reference", "exact"])
def test_b4q_kernel_matches_oracle(eng, pmr, ft, mvn):
    """Every block count, whole missing rows and single missing cells (row deletion), intercepts, both log(2 pi)
    conventions of every filter."""
    p, m, r = pmr
    B = 19
    w = W.random_dense(B, 139, p, m, r, seed=500 + 17 * p + m)
    y = w["y"].copy()
    y[[3, 17, 64, 138]] = np.nan
    y[5, 0] = np.nan
    y[70, 1:] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                           cholesky_triangle="correct", partial_missing="delete")
    eng.set_forward_kernel("b4")
    try:
        assert eng.forward_kernel_name(B, 139, p, m, r, ft, True) == "kf_fwd_b4q_kernel"
        out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
    finally:
        eng.set_forward_kernel("auto")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert not out["info"].any()
    assert np.all(out["ll_obs"][:, [3, 17, 64, 138]] == 0.0)
    assert np.array_equal(out["predicted_covs"][:, 1:], np.swapaxes(out["predicted_covs"][:, 1:], -1, -2))
    assert np.array_equal(out["filtered_covs"], np.swapaxes(out["filtered_covs"], -1, -2))


@pytest.mark.parametrize("n", [2, 15, 16, 17, 33])
def test_b4q_kernel_block_boundaries_and_partial_outputs(eng, n):
    """Series lengths around the 16-step y block / likelihood ring; every subset of tapes; guard words around the tapes."""
    import torch

    from pymc_statespace_amd.engine import FilterResult

    B, p, m = 6, 3, 11
    w = W.random_dense(B, n, p, m, 2, seed=17 * n + m)
    ref = kn.kalman_filter(*_args(w), mvn_const="exact")
    eng.set_forward_kernel("b4")
    try:
        for sel in ((True, True, True), (False, True, False), (False, False, True), (True, False, False)):
            r = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=sel[0], store_filtered=sel[1],
                            store_predicted=sel[2])
            assert np.max(np.abs(r.logp.cpu().numpy() - ref["logp"])) < 1e-8
            for k in TAPES:
                v = getattr(r, k)
                if v is not None:
                    assert_allclose(v.cpu().numpy(), ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{sel} {k}")

        def guarded(*shape):
            flat = torch.full((int(np.prod(shape)) + 4,), float("nan"), device="cuda", dtype=torch.float64)
            return flat, flat[2:2 + int(np.prod(shape))].view(*shape)

        bufs = {k: guarded(*s) for k, s in dict(ll_obs=(B, n), filtered_states=(B, n, m), predicted_states=(B, n + 1, m),
                                                filtered_covs=(B, n, m, m), predicted_covs=(B, n + 1, m, m)).items()}
        out = FilterResult(logp=torch.empty(B, device="cuda", dtype=torch.float64),
                           info=torch.empty(B, device="cuda", dtype=torch.int32), **{k: v[1] for k, v in bufs.items()})
        eng.forward(*_args(w), mvn_const="exact", out=out)
        for k, (flat, view) in bufs.items():
            assert_allclose(view.cpu().numpy(), ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
            g = flat.cpu().numpy()
            assert np.isnan(g[:2]).all() and np.isnan(g[-2:]).all(), f"{k}: wrote outside the tape"
    finally:
        eng.set_forward_kernel("auto")


def test_b4q_kernel_per_draw_data_not_pd_and_univariate_skip(eng):
    """y per draw; a draw whose F turns non-positive reports info = (t + 1) << 8 | (q + 1) and NaN; the Univariate filter
    skips F == 0 (:468-471) where Standard flags it; AUTO picks the kernel for the C4 shape."""
    w = W.varmax(B=21, n=130, k=3, order=(1, 1))
    rng = np.random.default_rng(3)
    yb = w["y"][None] + 0.1 * rng.standard_normal((21,) + w["y"].shape)
    H = w["H"].copy()
    H[7, 1, 1] = -50.0
    assert eng.forward_kernel_name(21, 130, w["p"], w["m"], w["r"], "standard", True) == "kf_fwd_b4q_kernel"
    res = eng.forward(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"],
                      store_filtered=False, store_ll_obs=False, h_diagonal=True)
    assert res.filtered_covs is None and res.ll_obs is None
    info = res.info.cpu().numpy()
    good = np.arange(21) != 7
    assert info[7] == ((0 + 1) << 8) | 2 and not info[good].any()
    ref = kn.kalman_filter(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    lp = res.logp.cpu().numpy()
    assert np.isnan(lp[7]) and np.max(np.abs(lp[good] - ref["logp"][good])) < 1e-8
    assert_allclose(res.predicted_covs.cpu().numpy()[good], ref["predicted_covs"][good], rtol=1e-9, atol=1e-9)
    # exactly known state and no noise: F == 0 for every observation of every step
    z = W.reference_test_inputs(2, 3, 1, 12)
    P0 = np.zeros_like(z["P0"])
    Hz, Qz = np.zeros((1, 2, 2)), np.zeros((1, 1, 1))
    for mvn in ("reference", "exact"):
        ru = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate", mvn_const=mvn)
        rf = kn.kalman_filter(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate", mvn_const=mvn)
        assert not ru.info.cpu().numpy().any()
        assert_allclose(ru.logp.cpu().numpy(), rf["logp"], atol=1e-12)
        assert_allclose(ru.filtered_states.cpu().numpy(), rf["filtered_states"], atol=1e-12)
        assert_allclose(ru.filtered_covs.cpu().numpy(), rf["filtered_covs"], atol=1e-12)
    rs = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="standard", h_diagonal=True)
    assert rs.info.cpu().numpy()[0] == ((0 + 1) << 8 | 1)


def test_b4q_kernel_agrees_with_tile_kernel_at_c4_size(eng):
    import torch

    w = W.varmax(B=512, n=1000)
    dev = [eng.to_device(x) for x in _args(w)]
    res = {}
    try:
        for kern in ("mfma", "b4"):
            eng.set_forward_kernel(kern)
            r = eng.forward(*dev, mvn_const="exact", h_diagonal=True)
            torch.cuda.synchronize()
            res[kern] = (r.logp.clone(), r.predicted_covs[::64, -1].clone(), r.ll_obs[::64].clone(), r.filtered_states[::64].clone())
    finally:
        eng.set_forward_kernel("auto")
    # (observations four at a time -- the default for p >= 3, m >= 8 -- against one at a time)
    import os

    os.environ["KF_B4Q_GROUP"] = "0"
    try:
        eng.set_forward_kernel("b4")
        r = eng.forward(*dev, mvn_const="exact", h_diagonal=True)
        torch.cuda.synchronize()
        res["seq"] = (r.logp.clone(), r.predicted_covs[::64, -1].clone(), r.ll_obs[::64].clone(), r.filtered_states[::64].clone())
    finally:
        os.environ.pop("KF_B4Q_GROUP")
        eng.set_forward_kernel("auto")
    for other in ("mfma", "seq"):
        assert (res["b4"][0] - res[other][0]).abs().max().item() < 1e-8
        assert torch.allclose(res["b4"][1], res[other][1], rtol=1e-9, atol=1e-11)
        assert torch.allclose(res["b4"][2], res[other][2], rtol=1e-9, atol=1e-10)
        assert torch.allclose(res["b4"][3], res[other][3], rtol=1e-9, atol=1e-10)
    assert not torch.equal(res["b4"][0], res["seq"][0])  # (different arithmetic: the switch did switch)


@pytest.mark.parametrize("korder", [(5, (2, 1)), (3, (1, 1)), (4, (2, 0)), (5, (2, 0))], ids=["p5m15", "p3m6", "p4m8", "p5m10"])
def test_b4q_kernel_steady_state_phase_matches_exact_recursion(eng, korder):
    """VARMAX shapes (3 <= p <= 5): the steady-state phase (mean tile through the frozen update's MFMAs, covariance tapes
    filled in bulk, state tapes through rings) against the step-by-step kernel and the oracle; missing rows and single
    missing cells delay it; every tape subset."""
    k, order = korder
    w = W.varmax(B=12, n=400, k=k, order=order)
    B, m, p = 12, w["m"], w["p"]
    rng = np.random.default_rng(1)
    c = 0.05 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    for missing in (False, True):
        y = w["y"].copy()
        if missing:
            y[[5, 100]] = np.nan
            y[150, 1] = np.nan
        for ft in ("standard", "univariate", "cholesky"):
            ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact",
                                   cholesky_triangle="correct", partial_missing="delete")
            outs = {}
            for exact in (True, False):
                out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", exact_recursion=exact,
                                      h_diagonal=True))
                assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
                for kk in TAPES:
                    assert_allclose(out[kk], ref[kk], rtol=1e-9, atol=1e-9, err_msg=f"{ft} exact={exact} {kk}")
                assert not out["info"].any()
                outs[exact] = out
            if m >= 8:  # (m <= 7 runs without groups and without the steady phase)
                assert not np.array_equal(outs[True]["predicted_states"], outs[False]["predicted_states"])
            assert_allclose(outs[True]["filtered_covs"], outs[False]["filtered_covs"], rtol=1e-10, atol=1e-12)
    full = _np(eng.forward(*_args(w), mvn_const="exact", h_diagonal=True))
    for sel in ((False, True, False), (False, False, True), (True, False, False), (False, False, False)):
        r = eng.forward(*_args(w), mvn_const="exact", h_diagonal=True, store_ll_obs=sel[0], store_filtered=sel[1],
                        store_predicted=sel[2])
        assert_allclose(r.logp.cpu().numpy(), full["logp"], rtol=1e-13)
        for kk in TAPES:
            v = getattr(r, kk)
            if v is not None:
                assert_allclose(v.cpu().numpy(), full[kk], rtol=1e-12, atol=1e-13, err_msg=f"{sel} {kk}")


@pytest.mark.parametrize("pm", [(1, 3), (1, 9), (3, 9), (5, 12)], ids=["lane", "b4", "b4q_p3", "b4q_p5"])
def test_steady_state_with_zero_innovation_variance(eng, pm):
    """Known state, no noise: P = 0 is a fixpoint from the first step on, F == 0 at every step of the steady phase.  The
    Univariate filter skips such observations (:468-471); the others flag the draw."""
    p, m = pm
    z = W.reference_test_inputs(p, m, 1, 120)
    P0 = np.zeros_like(z["P0"])
    Hz, Qz = np.zeros((1, p, p)), np.zeros((1, 1, 1))
    for mvn in ("reference", "exact"):
        r = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate", mvn_const=mvn)
        f = kn.kalman_filter(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="univariate", mvn_const=mvn)
        assert not r.info.cpu().numpy().any()
        assert_allclose(r.logp.cpu().numpy(), f["logp"], atol=1e-12)
        for k in TAPES:
            assert_allclose(getattr(r, k).cpu().numpy(), f[k], atol=1e-12, err_msg=k)
    rs = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type="standard", h_diagonal=True)
    assert rs.info.cpu().numpy()[0] == ((0 + 1) << 8 | 1) and not np.isfinite(rs.logp.cpu().numpy()).any()


@pytest.mark.parametrize("name", ["structural", "varmax", "arma11"])
def test_steady_state_per_draw_data_not_pd_and_nan_parameters(eng, name):
    """y per draw with a late NaN in one draw (that draw enters the steady phase behind it), a draw with negative H, a draw
    with a NaN in T: same flags as the step-by-step kernel, the other draws against the oracle."""
    w = getattr(W, name)(B=9, n=300)
    kw = dict(h_diagonal=True) if name == "varmax" else {}
    yb = np.repeat(w["y"][None], 9, 0).copy()
    yb[3, 250] = np.nan
    H = w["H"].copy()
    H[4] = -np.abs(H[4]) - 50.0 * np.eye(H.shape[-1])
    T = w["T"].copy()
    T[6, 0, 0] = np.nan
    r = eng.forward(yb, w["a0"], w["P0"], T, w["Z"], w["R"], H, w["Q"], mvn_const="exact", exact_recursion=False, **kw)
    x = eng.forward(yb, w["a0"], w["P0"], T, w["Z"], w["R"], H, w["Q"], mvn_const="exact", exact_recursion=True, **kw)
    assert np.array_equal(r.info.cpu().numpy(), x.info.cpu().numpy())
    lp = r.logp.cpu().numpy()
    assert np.isnan(lp[[4, 6]]).all()
    good = np.array([0, 1, 2, 3, 5, 7, 8])
    f = kn.kalman_filter(yb[good], *[w[k][good] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")], mvn_const="exact",
                         partial_missing="delete")
    assert np.max(np.abs(lp[good] - f["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(getattr(r, k).cpu().numpy()[good], f[k], rtol=1e-9, atol=1e-9, err_msg=k)


def test_steady_state_tapes_do_not_write_outside(eng):
    """Guard words around every tape of a run that spends most of its steps in the steady phase (bulk fills and ring
    windows start and end in the middle of their aligned windows)."""
    import torch

    from pymc_statespace_amd.engine import FilterResult

    for w, kw in ((W.varmax(B=5, n=203), dict(h_diagonal=True)), (W.structural(B=5, n=450), {})):
        B, n, m = 5, w["y"].shape[0], w["m"]
        ref = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True, **kw)

        def guarded(*shape):
            flat = torch.full((int(np.prod(shape)) + 6,), float("nan"), device="cuda", dtype=torch.float64)
            return flat, flat[3:3 + int(np.prod(shape))].view(*shape)

        bufs = {k: guarded(*s) for k, s in dict(ll_obs=(B, n), filtered_states=(B, n, m), predicted_states=(B, n + 1, m),
                                                filtered_covs=(B, n, m, m), predicted_covs=(B, n + 1, m, m)).items()}
        out = FilterResult(logp=torch.empty(B, device="cuda", dtype=torch.float64),
                           info=torch.empty(B, device="cuda", dtype=torch.int32), **{k: v[1] for k, v in bufs.items()})
        eng.forward(*_args(w), mvn_const="exact", out=out, exact_recursion=False, **kw)
        for k, (flat, view) in bufs.items():
            assert_allclose(view.cpu().numpy(), getattr(ref, k).cpu().numpy(), rtol=1e-9, atol=1e-10, err_msg=k)
            gd = flat.cpu().numpy()
            assert np.isnan(gd[:3]).all() and np.isnan(gd[-3:]).all(), f"{k}: wrote outside the tape"


# ----------------------------------------------------------------------------------------------
# accumulator-layout adjoint kernel (kf_backward_mfma.hip)
# ----------------------------------------------------------------------------------------------
BWD_MFMA_SHAPES = [(1, 1, 1), (1, 2, 1), (1, 5, 2), (1, 8, 3), (1, 10, 4), (1, 11, 5), (1, 13, 4), (1, 16, 6),
                   (2, 6, 2), (3, 4, 2), (5, 10, 5), (5, 15, 5), (8, 12, 6)]


@pytest.mark.parametrize("pmr", BWD_MFMA_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in BWD_MFMA_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_mfma_gradient_kernel_matches_autograd(eng, pmr, ft):
    from oracle import kalman_torch as kt

    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    B, n = 7, 70  # 70: crosses a 64-step staging block
    w = W.random_dense(B, n, p, m, r, seed=700 + 17 * p + m)
    y = w["y"].copy()
    y[2] = np.nan
    y[-1] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[66, 1:] = np.nan
    rng = np.random.default_rng(m * 10 + p)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    ft_oracle = "standard" if ft == "cholesky" else ft
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft_oracle)
    eng.set_forward_kernel("mfma")
    try:
        res = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, h_diagonal=True)
        g = eng.backward(*_args(w, y), c=c, d=d, predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                         filter_type=ft, want_cd=True, h_diagonal=True)
    finally:
        eng.set_forward_kernel("auto")
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if k == "H" and p > 1:
            # scalar-observation adjoint: diagonal of g_H only (the caller asserted a diagonal H)
            assert np.all(a[:, ~np.eye(p, dtype=bool)] == 0.0)
            a, b = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b, axis1=-2, axis2=-1)
        elif k in ("P0", "Q") and (p > 1 or ft == "univariate"):
            # entrywise partials w.r.t. symmetric inputs differ between the joint, the sequential
            # and the univariate recursions by antisymmetric terms; the symmetric parts agree
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-10, k


def test_mfma_gradient_needs_scalar_observations(eng):
    w = W.random_dense(3, 12, 3, 6, 2, seed=5, h_diag=False)
    res = eng.forward(*_args(w))
    eng.set_forward_kernel("mfma")
    try:
        with pytest.raises(RuntimeError, match="KF_FLAG_H_DIAGONAL"):
            eng.backward(*_args(w), predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=np.ones(3))
    finally:
        eng.set_forward_kernel("auto")


def test_mfma_gradient_per_draw_data(eng):
    from oracle import kalman_torch as kt

    w = W.structural(B=4, n=40)
    rng = np.random.default_rng(5)
    yb = w["y"][None] + 0.1 * rng.standard_normal((4,) + w["y"].shape)
    g_logp = rng.standard_normal(4)
    res = eng.forward(yb, *_args(w)[1:])
    g = eng.backward(yb, *_args(w)[1:], predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=g_logp)
    for b in range(4):
        sl = slice(b, b + 1)
        _, _, gt = kt.logp_and_grads(yb[b], *[x[sl] for x in _args(w)[1:]], g_logp=g_logp[sl])
        for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
            a, ref = g[k][b].cpu().numpy(), gt[k][0]
            assert np.max(np.abs(a - ref)) / (1.0 + np.max(np.abs(ref))) < 1e-10, (b, k)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 129])
def test_register_tile_kernels_at_block_boundaries(eng, n):
    """Series lengths around the 64-step staging / likelihood blocks of the MFMA kernels."""
    from oracle import kalman_torch as kt

    B, p, m, r = 5, 1, 10, 4
    w = W.random_dense(B, n, p, m, r, seed=900 + n)
    y = w["y"].copy()
    if n == 1:
        y[:] = 0.3  # standardising a single observation gives NaN
    if n > 2:
        y[n // 2] = np.nan
    ref = kn.kalman_filter(*_args(w, y), mvn_const="exact")
    eng.set_forward_kernel("mfma")
    try:
        out = _np(eng.forward(*_args(w, y), mvn_const="exact"))
        res = eng.forward(*_args(w, y))
        g = eng.backward(*_args(w, y), predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=np.ones(B))
    finally:
        eng.set_forward_kernel("auto")
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"n={n} {k}")
    _, _, gt = kt.logp_and_grads(*_args(w, y), g_logp=np.ones(B))
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
        a, b = g[k].cpu().numpy(), gt[k]
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-10, (n, k)
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)


def test_diagonal_h_detection_is_not_fooled_by_buffer_reuse(eng):
    """The cached diagonal-H check is keyed on the tensor object: a dense H that lands on the block a
    diagonal one just vacated must still go to the joint-update kernel."""
    import torch

    wd = W.random_dense(6, 25, 3, 8, 3, seed=41)
    wf = W.random_dense(6, 25, 3, 8, 3, seed=41, h_diag=False)
    dev = [eng.to_device(x) for x in _args(wd)]
    eng.forward(*dev)  # caches "diagonal"
    addr = dev[6].data_ptr()
    dev[6] = None
    torch.cuda.empty_cache()
    Hf = eng.to_device(wf["H"])  # same shape; typically the same address
    out = eng.forward(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], Hf, dev[7]).logp.cpu().numpy()
    ref = kn.kalman_filter(wd["y"], wd["a0"], wd["P0"], wd["T"], wd["Z"], wd["R"], wf["H"], wd["Q"])
    assert np.max(np.abs(out - ref["logp"])) < 1e-8, (addr, Hf.data_ptr())


def test_plain_cpp_consumer_of_the_c_abi():
    """tests/abi_c/abi_consumer.cpp: dlopen + hipMalloc + the extern "C" entry points, no Python or
    torch in the data path -- known answers, smoother, gradient vs a finite difference."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "abi_c", "abi_consumer")
    if not os.path.exists(exe):
        pytest.fail("tests/abi_c/abi_consumer is not built: run __graft_entry__.build()")
    lib = os.path.join(root, "pymc_statespace_amd", "libkf_hip.so")
    csv = os.path.join(root, "tests", "golden", "nile.csv")
    res = subprocess.run([exe, lib, csv], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ABI_CONSUMER_OK" in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize("m", [5, 6, 7, 8, 9, 10, 11, 12, 13, 15])
def test_newton_schulz_block_smoother(eng, m, monkeypatch):
    """kf_smoother_b4.hip (5 <= m <= 15): P_hat^-1 by warm-started Newton-Schulz iterations on block MFMAs, the state
    vector carried in the tile.  Against the pinv oracle on a full-rank model, on one with a diffuse start (a cold
    restart per step near t = 0), and -- singular P_hat: the iteration cannot settle -- through the flagged-draw
    fallback; and against the elimination kernel it replaces."""
    monkeypatch.setenv("KF_SMOOTHER_B4", "1")
    w = W.random_dense(9, 120, 1, m, m, seed=60 + m)
    for scale, tol in ((1.0, 1e-10), (1e6, 1e-7)):
        # (x 1e6: a diffuse start -- P_hat changes by orders of magnitude over the first steps and has a condition number of
        #  ~1e6 there: cold restarts, and a residual floor of cond x eps, which the tolerance allows for)
        P0 = w["P0"] * scale
        flt = kn.kalman_filter(w["y"], w["a0"], P0, w["T"], w["Z"], w["R"], w["H"], w["Q"])
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        scale_a, scale_P = 1.0 + np.abs(ra).max(), 1.0 + np.abs(rP).max()
        assert np.isfinite(sP.cpu().numpy()).all()
        assert np.abs(sa.cpu().numpy() - ra).max() / scale_a < tol
        assert np.abs(sP.cpu().numpy() - rP).max() / scale_P < tol
        assert np.array_equal(sa[:, -1].cpu().numpy(), flt["filtered_states"][:, -1])
        monkeypatch.setenv("KF_SMOOTHER_B4", "0")
        ea, eP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        monkeypatch.setenv("KF_SMOOTHER_B4", "1")
        assert np.abs((sa - ea).cpu().numpy()).max() / scale_a < tol and np.abs((sP - eP).cpu().numpy()).max() / scale_P < tol
    # singular prediction covariance (no state noise on an exactly known component): flagged, redone with the pinv
    w0 = W.random_dense(4, 30, 1, m, 1, seed=80 + m)
    P00 = w0["P0"].copy()
    P00[:, 0, :] = 0.0
    P00[:, :, 0] = 0.0
    flt0 = kn.kalman_filter(w0["y"], w0["a0"], P00, w0["T"], w0["Z"], w0["R"], w0["H"], w0["Q"])
    sa0, sP0 = eng.smooth(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    # (values: rank decisions of a pseudo-inverse sit at rounding level in every implementation, the reference's included)
    assert np.isfinite(sa0.cpu().numpy()).all() and np.isfinite(sP0.cpu().numpy()).all()
    assert np.array_equal(sa0[:, -1].cpu().numpy(), flt0["filtered_states"][:, -1])


@pytest.mark.parametrize("m", [5, 6, 7, 8, 9, 11, 12, 13, 16])
def test_register_tile_smoother_every_tile_size(eng, m):
    """kf_smooth_mfma_kernel for every accumulator count (KS = 2, 3, 4), odd and even m x m."""
    w = W.random_dense(9, 77, 1, m, m, seed=60 + m)  # full-rank R Q R': P_hat safely positive definite
    flt = kn.kalman_filter(*_args(w))
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)
    # and with a singular prediction covariance (zero state noise, exactly known initial state
    # components): the flagged draws take the pinv path of the LDS kernel
    w0 = W.random_dense(4, 30, 1, m, 1, seed=80 + m)
    P0 = w0["P0"].copy()
    P0[:, 0, :] = 0.0
    P0[:, :, 0] = 0.0
    flt0 = kn.kalman_filter(w0["y"], w0["a0"], P0, w0["T"], w0["Z"], w0["R"], w0["H"], w0["Q"])
    ra0, rP0 = kn.kalman_smoother(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    sa0, sP0 = eng.smooth(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    assert np.isfinite(sa0.cpu().numpy()).all() and np.isfinite(sP0.cpu().numpy()).all()


# ----------------------------------------------------------------------------------------------
# block-tile kernel (kf_forward_mfma_nb.hip): 16 < m <= 48 as 2 x 2 / 3 x 3 blocks of 16 x 16
# ----------------------------------------------------------------------------------------------
NB_SHAPES = [(1, m, max(1, m // 3)) for m in (17, 20, 21, 24, 25, 28, 29, 32, 33, 36, 37, 40, 41, 44, 45, 48)] + \
            [(3, 18, 4), (5, 27, 9), (8, 32, 32), (2, 35, 3), (4, 43, 20), (8, 48, 7)]


@pytest.mark.parametrize("pmr", NB_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in NB_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_block_tile_kernel_matches_oracle(eng, pmr, ft):
    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    mvn = "exact" if (m + p) % 2 else "reference"
    B, n = 7, 70
    w = W.random_dense(B, n, p, m, r, seed=900 + 17 * p + m)
    y = w["y"].copy()
    y[[3, 17, 64]] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[40, 1:] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                           cholesky_triangle="correct", partial_missing="delete")
    assert eng.forward_kernel_name(B, n, p, m, r, ft, True) == "kf_fwd_mfma_nb_kernel"
    out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert not out["info"].any()
    assert np.all(out["ll_obs"][:, [3, 17, 64]] == 0.0)
    assert np.array_equal(out["predicted_covs"][:, 1:], np.swapaxes(out["predicted_covs"][:, 1:], -1, -2))
    assert np.array_equal(out["filtered_covs"], np.swapaxes(out["filtered_covs"], -1, -2))


def test_block_tile_kernel_agrees_with_lds_kernel_per_draw_data_and_partial_outputs(eng):
    B, n, p, m, r = 40, 130, 2, 30, 6
    w = W.random_dense(B, n, p, m, r, seed=77)
    rng = np.random.default_rng(5)
    yb = w["y"][None] + 0.1 * rng.standard_normal((B,) + w["y"].shape)
    H = w["H"].copy()
    H[7, 1, 1] = -500.0
    args = (yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    res = eng.forward(*args, store_filtered=False, store_ll_obs=False, h_diagonal=True)
    assert res.filtered_covs is None and res.ll_obs is None
    info = res.info.cpu().numpy()
    good = np.arange(B) != 7
    assert info[7] == ((0 + 1) << 8) | 2 and not info[good].any()
    eng.set_forward_kernel("lds")
    try:
        ref = eng.forward(*args, h_diagonal=True)
    finally:
        eng.set_forward_kernel("auto")
    lp, lr = res.logp.cpu().numpy(), ref.logp.cpu().numpy()
    assert np.isnan(lp[7]) and np.max(np.abs(lp[good] - lr[good])) < 1e-8
    assert_allclose(res.predicted_covs.cpu().numpy()[good], ref.predicted_covs.cpu().numpy()[good],
                    rtol=1e-9, atol=1e-9)
    assert_allclose(res.predicted_states.cpu().numpy()[good], ref.predicted_states.cpu().numpy()[good],
                    rtol=1e-9, atol=1e-9)


BWD_NB_SHAPES = [(1, m, max(1, m // 3)) for m in (17, 20, 23, 24, 28, 32, 33, 36, 40, 44, 45, 48)] + \
                [(3, 18, 4), (5, 27, 9), (8, 32, 10), (2, 35, 3), (4, 43, 12), (8, 48, 7)]


@pytest.mark.parametrize("pmr", BWD_NB_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in BWD_NB_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "univariate", "single"])
def test_block_tile_gradient_kernel_matches_autograd(eng, pmr, ft):
    from oracle import kalman_torch as kt

    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    if ft == "univariate" and m % 4 != 0 and p == 1:
        pytest.skip("covered by the standard filter at this shape")
    B, n = 3, 67  # 67: crosses a 64-step staging block
    w = W.random_dense(B, n, p, m, r, seed=1700 + 17 * p + m)
    y = w["y"].copy()
    y[2] = np.nan
    y[-1] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[65, 1:] = np.nan
    rng = np.random.default_rng(m * 10 + p)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft)
    res = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, h_diagonal=True)
    g = eng.backward(*_args(w, y), c=c, d=d, predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                     filter_type=ft, want_cd=True, h_diagonal=True)
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if k == "H" and p > 1:
            assert np.all(a[:, ~np.eye(p, dtype=bool)] == 0.0)
            a, b = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b, axis1=-2, axis2=-1)
        elif k in ("P0", "Q") and (p > 1 or ft == "univariate"):
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-10, k


@pytest.mark.parametrize("m", [17, 20, 22, 24, 27, 28, 32, 33, 36, 39, 40, 44, 47])
def test_block_tile_smoother_matches_oracle(eng, m):
    """kf_smooth_mfma_nb_kernel for every (blocks, slices) instantiation, odd and even m."""
    w = W.random_dense(5, 40, 1, m, m, seed=160 + m)  # full-rank R Q R': P_hat safely positive definite
    flt = kn.kalman_filter(*_args(w))
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)
    # agrees with the LDS-resident smoother (Cholesky path) it replaces
    eng.set_forward_kernel("lds")
    try:
        la, lP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    finally:
        eng.set_forward_kernel("auto")
    assert_allclose(sa.cpu().numpy(), la.cpu().numpy(), rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), lP.cpu().numpy(), rtol=1e-8, atol=1e-8)
    # singular prediction covariance: flagged draws take the pinv path of the LDS kernel
    w0 = W.random_dense(3, 20, 1, m, 1, seed=180 + m)
    P0 = w0["P0"].copy()
    P0[:, 0, :] = 0.0
    P0[:, :, 0] = 0.0
    flt0 = kn.kalman_filter(w0["y"], w0["a0"], P0, w0["T"], w0["Z"], w0["R"], w0["H"], w0["Q"])
    sa0, sP0 = eng.smooth(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    # (cond(P_hat) ~ 1e9 .. 1e11 here: every solver, the oracle's pinv included, returns noise)
    assert np.isfinite(sa0.cpu().numpy()).all() and np.isfinite(sP0.cpu().numpy()).all()


# ----------------------------------------------------------------------------------------------
# workgroup-per-draw kernel (kf_forward_mfma_wg.hip): 48 < m <= 64, four wavefronts per draw
# ----------------------------------------------------------------------------------------------
WG_SHAPES = [(1, m, max(1, m // 4)) for m in (49, 50, 52, 53, 56, 57, 60, 61, 63, 64)] + \
            [(3, 51, 4), (5, 55, 9), (8, 64, 20), (2, 59, 3), (1, 64, 64), (1, 50, 50)]


@pytest.mark.parametrize("pmr", WG_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in WG_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_workgroup_kernel_matches_oracle(eng, pmr, ft):
    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    if ft in ("cholesky", "single") and m % 4 not in (0, 1):
        pytest.skip("covered by the standard filter at this shape")
    mvn = "exact" if (m + p) % 2 else "reference"
    B, n = 5, 70
    w = W.random_dense(B, n, p, m, r, seed=2900 + 17 * p + m)
    y = w["y"].copy()
    y[[3, 17, 64]] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[40, 1:] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                           cholesky_triangle="correct", partial_missing="delete")
    assert eng.forward_kernel_name(B, n, p, m, r, ft, True) == "kf_fwd_mfma_wg_kernel"
    out = _np(eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn))
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert not out["info"].any()
    assert np.all(out["ll_obs"][:, [3, 17, 64]] == 0.0)
    assert np.array_equal(out["predicted_covs"][:, 1:], np.swapaxes(out["predicted_covs"][:, 1:], -1, -2))
    assert np.array_equal(out["filtered_covs"], np.swapaxes(out["filtered_covs"], -1, -2))


def test_workgroup_kernel_per_draw_data_not_pd_and_partial_outputs(eng):
    B, n, p, m, r = 24, 130, 2, 58, 6
    w = W.random_dense(B, n, p, m, r, seed=78)
    rng = np.random.default_rng(5)
    yb = w["y"][None] + 0.1 * rng.standard_normal((B,) + w["y"].shape)
    H = w["H"].copy()
    H[7, 1, 1] = -500.0
    args = (yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    res = eng.forward(*args, store_filtered=False, store_ll_obs=False, h_diagonal=True)
    assert res.filtered_covs is None and res.ll_obs is None
    info = res.info.cpu().numpy()
    good = np.arange(B) != 7
    assert info[7] == ((0 + 1) << 8) | 2 and not info[good].any()
    ref = kn.kalman_filter(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    lp = res.logp.cpu().numpy()
    assert np.isnan(lp[7]) and np.max(np.abs(lp[good] - ref["logp"][good])) < 1e-8
    assert_allclose(res.predicted_covs.cpu().numpy()[good], ref["predicted_covs"][good], rtol=1e-9, atol=1e-9)
    assert_allclose(res.predicted_states.cpu().numpy()[good], ref["predicted_states"][good], rtol=1e-9, atol=1e-9)


BWD_WG_SHAPES = [(1, m, max(1, m // 4)) for m in (49, 52, 53, 56, 57, 60, 61, 64)] + \
                [(3, 51, 4), (5, 55, 9), (8, 64, 12), (2, 59, 3), (1, 64, 64)]


@pytest.mark.parametrize("pmr", BWD_WG_SHAPES, ids=[f"p{p}m{m}r{r}" for p, m, r in BWD_WG_SHAPES])
@pytest.mark.parametrize("ft", ["standard", "univariate", "single"])
def test_workgroup_gradient_kernel_matches_autograd(eng, pmr, ft):
    from oracle import kalman_torch as kt

    p, m, r = pmr
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    if ft != "standard" and p == 1 and m % 4 not in (0, 1):
        pytest.skip("covered by the standard filter at this shape")
    B, n = 3, 67  # 67: crosses a 64-step staging block
    w = W.random_dense(B, n, p, m, r, seed=2700 + 17 * p + m)
    y = w["y"].copy()
    y[2] = np.nan
    y[-1] = np.nan
    if p > 1:
        y[5, 0] = np.nan
        y[65, 1:] = np.nan
    rng = np.random.default_rng(m * 10 + p)
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft)
    res = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, h_diagonal=True)
    g = eng.backward(*_args(w, y), c=c, d=d, predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                     filter_type=ft, want_cd=True, h_diagonal=True)
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), gt[k]
        if k == "H" and p > 1:
            assert np.all(a[:, ~np.eye(p, dtype=bool)] == 0.0)
            a, b = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b, axis1=-2, axis2=-1)
        elif k in ("P0", "Q") and (p > 1 or ft == "univariate"):
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-10, k


@pytest.mark.parametrize("m", [48, 49, 52, 53, 56, 57, 60, 61, 64])
def test_workgroup_smoother_matches_oracle(eng, m):
    """kf_smooth_mfma_wg_kernel for every slice count, odd and even m."""
    w = W.random_dense(4, 40, 1, m, m, seed=260 + m)  # full-rank R Q R': P_hat safely positive definite
    flt = kn.kalman_filter(*_args(w))
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)
    # exactly singular prediction covariance (no state noise, initial state known exactly): the draw is
    # flagged and redone by kf_smooth_pinv_wide_kernel (Jacobi pinv, as the reference); the others stay
    w0 = W.random_dense(3, 12, 1, m, m, seed=280 + m)
    Q = w0["Q"].copy()
    Q[1] = 0.0
    P0 = w0["P0"].copy()
    P0[1] = 0.0
    flt0 = kn.kalman_filter(w0["y"], w0["a0"], P0, w0["T"], w0["Z"], w0["R"], w0["H"], Q)
    ra0, rP0 = kn.kalman_smoother(w0["T"], w0["R"], Q, flt0["filtered_states"], flt0["filtered_covs"])
    sa0, sP0 = eng.smooth(w0["T"], w0["R"], Q, flt0["filtered_states"], flt0["filtered_covs"])
    assert_allclose(sa0.cpu().numpy(), ra0, rtol=1e-8, atol=1e-8)
    assert_allclose(sP0.cpu().numpy(), rP0, rtol=1e-8, atol=1e-8)
    # rank-deficient but not zero: state noise on 5 of m directions, P0 of rank 7
    w1 = W.random_dense(2, 10, 1, m, 5, seed=290 + m)
    rng = np.random.default_rng(m)
    Lr = rng.standard_normal((2, m, 7))
    P1 = Lr @ np.swapaxes(Lr, -1, -2)
    flt1 = kn.kalman_filter(w1["y"], w1["a0"], P1, w1["T"], w1["Z"], w1["R"], w1["H"], w1["Q"])
    ra1, rP1 = kn.kalman_smoother(w1["T"], w1["R"], w1["Q"], flt1["filtered_states"], flt1["filtered_covs"])
    sa1, sP1 = eng.smooth(w1["T"], w1["R"], w1["Q"], flt1["filtered_states"], flt1["filtered_covs"])
    # (eigenvalues of P_hat between the 1e-15 cutoff and ~1e-10 are kept by every pinv and amplify
    # rounding noise by their reciprocal: the answers agree to what such a problem determines)
    assert np.isfinite(sa1.cpu().numpy()).all() and np.isfinite(sP1.cpu().numpy()).all()
    assert_allclose(sa1.cpu().numpy(), ra1, rtol=1e-2, atol=1e-3)
    assert_allclose(sP1.cpu().numpy(), rP1, rtol=1e-2, atol=1e-3)


# ----------------------------------------------------------------------------------------------
# the C-ABI never allocates, synchronises or copies to the host: its launches can be captured
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 10, 5), (1, 3, 2), (1, 24, 6), (1, 52, 6), (2, 15, 4)],
                         ids=["tile", "lane", "block_tile", "workgroup", "tile_p2"])
def test_forward_gradient_smoother_replay_from_a_hip_graph(eng, shape):
    """Capture forward + gradient + smoother into one graph, change the data in place, replay:
    the replayed outputs equal an eager run on the new data (one NUTS-style step without any
    per-launch host work)."""
    import torch

    p, m, r = shape
    B, n = 48, 40
    w = W.random_dense(B, n, p, m, r, seed=31 + m)
    dev = [eng.to_device(x) for x in _args(w)]
    gl = torch.ones(B, dtype=torch.float64, device=dev[0].device)
    kw = dict(h_diagonal=True)

    def step():
        res = eng.forward(*dev, **kw)
        g = eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                         g_logp=gl, h_diagonal=True)
        sa, sP = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
        vg = eng.value_and_grad(*dev, h_diagonal=True)  # the one-call entry point (its own tapes in the workspace)
        return res, g, sa, sP, vg

    step()  # lazy initialisation (workspaces, function attributes) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res_c, g_c, sa_c, sP_c, vg_c = step()
    y2 = w["y"] + 0.3 * np.random.default_rng(1).standard_normal(w["y"].shape)
    dev[0].copy_(torch.as_tensor(y2, device=dev[0].device))
    graph.replay()
    torch.cuda.synchronize()
    res_e, g_e, sa_e, sP_e, vg_e = step()
    torch.cuda.synchronize()
    assert torch.equal(res_c.logp, res_e.logp) and torch.equal(res_c.predicted_covs, res_e.predicted_covs)
    for k in g_e:
        assert torch.equal(g_c[k], g_e[k]), k
    assert torch.equal(sa_c, sa_e) and torch.equal(sP_c, sP_e)
    assert torch.equal(vg_c[0], vg_e[0])
    for k in vg_e[2]:
        assert torch.equal(vg_c[2][k], vg_e[2][k]), k
        a_, b_ = vg_c[2][k].cpu().numpy(), g_e[k].cpu().numpy() if k in g_e else None
        if b_ is not None:
            assert np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-9, k
    ref = kn.kalman_filter(y2, *_args(w)[1:])
    assert np.max(np.abs(res_c.logp.cpu().numpy() - ref["logp"])) < 1e-8


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129])
@pytest.mark.parametrize("m", [21, 37, 50], ids=["block_tile_2x2", "block_tile_3x3", "workgroup"])
def test_mid_size_kernels_at_block_boundaries(eng, n, m):
    """Series lengths around the 64-step staging / likelihood blocks of the block-tile and
    workgroup kernels; per-draw data; gradient of a per-step weighting (g_ll_obs)."""
    from oracle import kalman_torch as kt

    B, p, r = 3, 1, m
    w = W.random_dense(B, n, p, m, r, seed=1900 + n + m)
    rng = np.random.default_rng(n + m)
    yb = np.repeat(w["y"][None], B, axis=0)
    if n == 1:
        yb[:] = 0.3  # standardising a single observation gives NaN
    yb = yb + 0.1 * rng.standard_normal(yb.shape)
    if n > 2:
        yb[:, n // 2] = np.nan
    if n == 1:
        yb = yb[0]  # (B, 1, 1) would read as the reference's (T, p, 1) data layout: shared data instead
    args = (yb,) + tuple(_args(w)[1:])
    ref = kn.kalman_filter(*args, mvn_const="exact")
    out = _np(eng.forward(*args, mvn_const="exact"))
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"n={n} {k}")
    g_ll = rng.standard_normal((B, n))
    res = eng.forward(*args)
    g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_ll_obs=g_ll)
    for b in range(B):
        sl = slice(b, b + 1)
        _, _, gt = kt.logp_and_grads(yb if n == 1 else yb[b], *[x[sl] for x in _args(w)[1:]], g_logp=np.zeros(1),
                                     g_ll_obs=g_ll[sl])
        for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
            a, ref_g = g[k][b].cpu().numpy(), gt[k][0]
            assert np.max(np.abs(a - ref_g)) / (1.0 + np.max(np.abs(ref_g))) < 1e-10, (n, b, k)
    ref0 = kn.kalman_filter(*args)
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], ref0["filtered_states"], ref0["filtered_covs"])
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], ref0["filtered_states"], ref0["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("m", [10, 21, 37, 50], ids=["tile", "block_tile_2x2", "block_tile_3x3", "workgroup"])
def test_smoother_errors_do_not_grow_over_long_series(eng, m):
    """The backward recursion must not amplify rounding errors: 300 steps, every register-tile family.
    (A block-tile smoother that mirrored its off-diagonal blocks but left the diagonal blocks
    unsymmetrised passed at 40 steps and was off by 1e-4 after 129: errors grew 1.45x per step.)"""
    n, B = 300, 2
    w = W.random_dense(B, n, 1, m, m, seed=3100 + m)
    flt = kn.kalman_filter(*_args(w))
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    err = np.abs(sP.cpu().numpy() - rP).reshape(B, n, -1).max(axis=2).max(axis=0)
    assert err.max() < 1e-9, (err[:5], err[-5:])
    assert err[:20].max() < 50 * max(err[-40:-1].max(), 1e-14), "error grows backwards in time"
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("m", [10, 21, 37, 50], ids=["tile", "block_tile_2x2", "block_tile_3x3", "workgroup"])
def test_forward_and_gradient_over_a_long_series(eng, m):
    """400 steps: the recursions of the register-tile families neither drift nor lose symmetry."""
    from oracle import kalman_torch as kt

    n, B = 400, 2
    w = W.random_dense(B, n, 1, m, max(2, m // 3), seed=3300 + m)
    ref = kn.kalman_filter(*_args(w))
    res = eng.forward(*_args(w))
    out = _np(res)
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    pc = out["predicted_covs"][:, 1:]
    assert np.array_equal(pc, np.swapaxes(pc, -1, -2))
    g_logp = np.array([1.0, -0.5])
    _, _, gt = kt.logp_and_grads(*_args(w), g_logp=g_logp)
    g = eng.backward(*_args(w), predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_logp=g_logp)
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
        a, b = g[k].cpu().numpy(), gt[k]
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-9, k


@pytest.mark.parametrize("s", [24, 40, 52, 63], ids=["m25", "m41", "m53_weekly", "m64"])
def test_seasonal_structural_models_on_the_mid_size_kernels(eng, s):
    """Level + trend + s-period dummy seasonal (unit roots, sparse T, r = 3): the model family that
    lands in the 16 < m <= 64 range in practice (weekly data: m = 53)."""
    from oracle import kalman_torch as kt

    B, n = 3, 150
    w = W.daily_seasonal(B=B, n=n, s=s)
    m = s + 1
    y = w["y"].copy()
    y[[7, 77]] = np.nan
    assert eng.forward_kernel_name(B, n, 1, m, 3) in ("kf_fwd_mfma_nb_kernel", "kf_fwd_mfma_wg_kernel")
    ref = kn.kalman_filter(*_args(w, y))
    res = eng.forward(*_args(w, y))
    out = _np(res)
    assert np.max(np.abs(out["logp"] - ref["logp"])) < 1e-8
    for k in TAPES:
        assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    g_logp = np.array([1.0, -0.5, 2.0])
    _, _, gt = kt.logp_and_grads(*_args(w, y), g_logp=g_logp)
    g = eng.backward(*_args(w, y), predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_logp=g_logp)
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
        a, b = g[k].cpu().numpy(), gt[k]
        assert np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b))) < 1e-9, k
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
    assert_allclose(sa.cpu().numpy(), ra, rtol=1e-7, atol=1e-7)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-7, atol=1e-7)


def test_workspace_is_only_requested_where_a_workspace_path_runs(eng):
    """kf_workspace_bytes follows the kernel selection: nothing (or the per-draw flags of the tile
    smoothers) while an on-chip kernel takes the shape, m^2-sized scratch only beyond."""
    import ctypes as C
    from pymc_statespace_amd import _lib

    def ws(B, n, p, m, r, ft="standard", hdiag=False):
        dims = eng._dims(B, n, p, m, r, ft)
        if hdiag:
            dims.flags |= _lib.KF_FLAG_H_DIAGONAL
        nb = C.c_size_t(0)
        assert eng.lib.kf_workspace_bytes(eng._h, C.byref(dims), C.byref(nb)) == 0
        return nb.value

    # (p = 1, m <= 14: the block adjoint hands its last step over through B (m + m^2) doubles; else the tile smoother's flags)
    assert ws(4096, 1000, 1, 10, 4) == 8 * 4096 * (10 + 100) + 64
    assert ws(4096, 1000, 1, 16, 4) == 4 * 4096 + 64
    assert ws(4096, 1000, 1, 40, 4) == 4 * 4096 + 64
    assert ws(4096, 1000, 1, 64, 8) == 4 * 4096 + 64            # workgroup kernels: all on chip (+ flags)
    assert ws(4096, 1000, 3, 60, 8, "standard", True) == 4 * 4096 + 64
    assert ws(4096, 1000, 3, 60, 8, "standard", False) > 4096 * 60 * 60 * 8   # dense H: workspace paths
    assert ws(128, 100, 1, 367, 3) >= 7 * 128 * 367 * 367 * 8   # C5


@pytest.mark.parametrize("m", [29, 45, 57], ids=["block_tile_2x2", "block_tile_3x3", "workgroup"])
def test_mid_size_kernels_at_a_full_size_batch(eng, m):
    """4099 draws (more workgroups than the chip holds at once, odd count): sampled draws against the
    oracle for forward, gradient and smoother; every draw finite; identical draws give identical bits."""
    from oracle import kalman_torch as kt

    B, n, p, r = 4099, 30, 1, m  # full-rank state noise: well-conditioned prediction covariances
    base = W.random_dense(7, n, p, m, r, seed=4100 + m)
    idx = np.arange(B) % 7
    w = {k: (v if k == "y" else v[idx]) for k, v in base.items() if isinstance(v, np.ndarray)}
    res = eng.forward(*_args(w))
    gl = np.ones(B)
    g = eng.backward(*_args(w), predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl)
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    lp = res.logp.cpu().numpy()
    assert np.isfinite(lp).all() and not res.info.cpu().numpy().any()
    ref = kn.kalman_filter(*_args(base))
    assert np.max(np.abs(lp - ref["logp"][idx])) < 1e-8
    # replicas of the same parameters: bitwise equal wherever they ran
    for k in range(7):
        sel = np.where(idx == k)[0]
        assert np.all(lp[sel] == lp[sel[0]])
        assert bool((g["T"][sel] == g["T"][sel[0]]).all()) and bool((sP[sel] == sP[sel[0]]).all())
    _, _, gt = kt.logp_and_grads(*_args(base), g_logp=np.ones(7))
    for k in ("a0", "P0", "T", "Z", "R", "H", "Q"):
        a, b_ = g[k][-7:].cpu().numpy(), gt[k][idx[-7:]]
        assert np.max(np.abs(a - b_)) / (1.0 + np.max(np.abs(b_))) < 1e-9, k
    ra, rP = kn.kalman_smoother(base["T"], base["R"], base["Q"], ref["filtered_states"], ref["filtered_covs"])
    assert_allclose(sa[-7:].cpu().numpy(), ra[idx[-7:]], rtol=1e-7, atol=1e-8)
    assert_allclose(sP[-7:].cpu().numpy(), rP[idx[-7:]], rtol=1e-7, atol=1e-8)


def test_random_shapes_against_the_oracles(eng):
    """scripts/gpu_fuzz.py: random (p, m, r, n, filter, constants, missing data, intercepts) up to
    m = 71 -- whatever kernel AUTO picks must match the CPU oracles (forward tapes, gradient, smoother)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gpu_fuzz

    flagged, worst, kernels = gpu_fuzz.run(n_cases=60, seed=7, eng=eng, verbose=False)
    assert flagged == 0, worst
    assert len(kernels) >= 4  # several kernel families were exercised
