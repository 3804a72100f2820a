"""Round-5 paths through the C-ABI against the oracle.

* kf_smoother_from_filter (csrc/kf_smoother_bf.hip): the smoothed moments of KalmanSmoother.build_graph
  (pymc_statespace/filters/kalman_smoother.py:56-104) from the filter's PREDICTED tapes and the observation side of the model,
  without pinv(P_hat).  Held to (i) the statsmodels golden vectors -- the reference's own external oracle,
  tests/test_kalman_filter.py:226-241, at the reference's tolerances --, (ii) the numpy restatement of the same recursion
  (oracle/kalman_numpy.py::smoother_from_filter, itself pinned to every golden vector: 1e-10) and the restated reference recursion
  on the filtered tapes of the same forward call (rtol = atol = 1e-9), (iii) kf_smoother itself at
  BASELINE sizes (<= 1e-11 of a draw's largest entry), and the edge cases the reference tests: missing observations, the last
  step equal to the filtered moments, intercept d with SingleTimeseriesFilter's sign.
* time-varying R / Q on the block tiles (kf_fwd_b4tv_kernel, r <= 4): reference filters/kalman_filter.py:79-124, 231-252 -- slice t of
  every flagged matrix at step t -- against the numpy oracle and the LDS kernel it replaces.
* block sparsity of T in the m > 64 workspace path (kf_forward_big.hip, big_tmask_kernel): K chunks whose block of T is all zero are
  skipped; bit for bit the product with every chunk (KF_BIG_TMASK=0), and the oracle's numbers.
* kf_filter_value_and_grad on a DENSE H with p > 1 by whitening (kf_api.hip, kf_forward_whiten.hip: kf_unwhiten_grad_kernel): the
  reference differentiates any positive definite H through its scan (filters/kalman_filter.py:262-275, 289-305; core/statespace.py:174);
  against torch autograd of the restated joint recursion at 1e-9 of each gradient's largest entry, against the dense LDS adjoint it
  replaces, and the fallbacks (a partially missing row, an H that is not positive definite).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from conftest import golden_files, load_golden
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


def _p1_golden():
    return [f for f in golden_files() if load_golden(f)["Z"].shape[-2] == 1]


@pytest.mark.parametrize("name", _p1_golden())
def test_smoother_from_filter_matches_statsmodels_golden(eng, name):
    g = load_golden(name)
    res = eng.forward(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"], mvn_const="exact")
    sa, sP = eng.smooth_from_filter(g["y"], g["T"], g["Z"], g["H"], res.predicted_states, res.predicted_covs)
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    if name.startswith("nile_llt"):
        # the reference's own tolerances: atol 1e-7, covariances from index 5 on (tests/test_kalman_filter.py:236-241)
        assert_allclose(sa, g["smoothed_state"], rtol=1e-7, atol=1e-7)
        assert_allclose(sP[:, 5:], g["smoothed_state_cov"][:, 5:], rtol=1e-7, atol=1e-7)
    else:
        # (arma11, H = 0 and a singular P_hat: statsmodels runs this very recursion -- no pinv rank decision to disagree on)
        assert_allclose(sa, g["smoothed_state"], rtol=1e-9, atol=1e-9)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-9, atol=1e-9)
    # last smoothed == last filtered (tests/test_kalman_filter.py:212-222), here to rounding: a_f = a + P Z' v / F
    assert_allclose(sa[:, -1], res.filtered_states[:, -1].cpu().numpy(), rtol=1e-12, atol=1e-12 * np.abs(sa).max())
    assert_allclose(sP[:, -1], res.filtered_covs[:, -1].cpu().numpy(), rtol=0, atol=1e-12 * np.abs(sP).max())


CASES = {
    "structural_m10": lambda: W.structural(B=33, n=120),
    "structural_m10_long": lambda: W.structural(B=5, n=700),
    "dense_p1m5": lambda: W.random_dense(19, 60, 1, 5, 3, seed=3),
    "dense_p1m7": lambda: W.random_dense(11, 50, 1, 7, 7, seed=4),
    "dense_p1m11": lambda: W.random_dense(9, 45, 1, 11, 4, seed=5),
    "dense_p1m12": lambda: W.random_dense(9, 45, 1, 12, 12, seed=6),
    "dense_p1m15": lambda: W.random_dense(7, 40, 1, 15, 6, seed=7),
    "dense_p1m3": lambda: W.random_dense(21, 64, 1, 3, 3, seed=8),
    "nile_local_level": lambda: W.nile_local_level(B=9),
    "arma11": lambda: W.arma11(B=17, n=90),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("missing", [False, True])
def test_smoother_from_filter_matches_reference_recursion(eng, case, missing):
    w = CASES[case]()
    y = w["y"].copy()
    if missing:
        y[[0, 3, 4, 17, y.shape[0] - 1]] = np.nan
    res = eng.forward(*_args(w, y), mvn_const="exact")
    assert not res.info.cpu().numpy().any()
    sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    # (i) the restatement of this very recursion (oracle/kalman_numpy.py::smoother_from_filter, pinned to statsmodels): 1e-10
    da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy())
    assert_allclose(sa.cpu().numpy(), da, rtol=1e-10, atol=1e-10)
    assert_allclose(sP.cpu().numpy(), dP, rtol=1e-10, atol=1e-10 * max(1.0, float(np.abs(res.filtered_covs.cpu().numpy()).max())))
    # (ii) the reference's own recursion on the filtered tapes of the same call
    na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
    na = na.reshape(sa.shape)
    if case == "arma11":
        # H = 0: P_hat is exactly rank deficient and the reference's pinv decides a rank at rounding level -- the restated
        # recursion itself moves by percents in single covariance entries there (the RTS kernels are held to 1e-2,
        # tests/test_gpu_parity.py).  States only; the golden test above pins this case at 1e-9 against statsmodels.
        assert_allclose(sa.cpu().numpy(), na, rtol=1e-5, atol=1e-5)
        return
    assert_allclose(sa.cpu().numpy(), na, rtol=1e-9, atol=1e-9)
    assert_allclose(sP.cpu().numpy(), nP, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("ft", ["standard", "single"])
def test_smoother_from_filter_intercept_d(eng, ft):
    """d enters v = y - Z a - d; SingleTimeseriesFilter flips its sign (kalman_filter.py:335) and the smoother must follow the
    filter that wrote the tapes."""
    w = W.structural(B=6, n=80)
    rng = np.random.default_rng(5)
    d = 0.3 * rng.standard_normal((6, 1))
    res = eng.forward(*_args(w), d=d, filter_type=ft, mvn_const="exact")
    sa, sP = eng.smooth_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d, filter_type=ft)
    ra, rP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    assert_allclose(sa.cpu().numpy(), ra.cpu().numpy(), rtol=1e-9, atol=1e-9)
    assert_allclose(sP.cpu().numpy(), rP.cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_smoother_from_filter_y_per_draw(eng):
    w = W.structural(B=8, n=64)
    rng = np.random.default_rng(9)
    yb = w["y"][None] + 0.1 * rng.standard_normal((8,) + w["y"].shape)
    res = eng.forward(yb, *[w[k] for k in KEYS], mvn_const="exact")
    sa, sP = eng.smooth_from_filter(yb, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    ra, rP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    assert_allclose(sa.cpu().numpy(), ra.cpu().numpy(), rtol=1e-9, atol=1e-9)
    assert_allclose(sP.cpu().numpy(), rP.cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_smoother_from_filter_rejects_what_it_does_not_take(eng):
    """p > 1 and m > 1024 are kf_smoother's: KF_ERR_UNSUPPORTED with a message, nothing launched."""
    import ctypes as C

    import torch

    from pymc_statespace_amd import _lib
    from pymc_statespace_amd.engine import KalmanError

    t = torch.zeros(8, device=eng.device, dtype=torch.float64)
    for p_, m_ in ((5, 10), (1, 1025)):
        dims = _lib.KfDims(3, 20, p_, m_, 2, 0, 0)
        rc = eng.lib.kf_smoother_from_filter(eng._h, C.byref(dims), *([t.data_ptr()] * 9), None, 0, None)
        assert rc == _lib.KF_ERR_UNSUPPORTED
        with pytest.raises(KalmanError, match="p == 1"):
            eng._check(rc, "kf_smoother_from_filter")
        assert not eng.smooth_from_filter_ok(p_, m_)
    # 79 < m <= 1024 runs on the HBM workspace: without it the call is refused (KF_ERR_WORKSPACE), nothing is launched
    dims = _lib.KfDims(3, 20, 1, 100, 2, 0, 0)
    assert eng.lib.kf_smoother_from_filter(eng._h, C.byref(dims), *([t.data_ptr()] * 9), None, 0, None) == _lib.KF_ERR_WORKSPACE
    assert eng.smooth_from_filter_ok(1, 10) and eng.smooth_from_filter_ok(1, 367)
    assert not eng.smooth_from_filter_ok(1, 10, time_varying=("T",))


@pytest.mark.parametrize("name,B,n,ft", [("headline", 4096, 1000, "standard"), ("c3", 8192, 2000, "cholesky"),
                                         ("c2", 4096, 1000, "standard")])
def test_smoother_from_filter_full_size_against_kf_smoother(eng, name, B, n, ft):
    """BASELINE sizes: within 1e-11 of a draw's largest entry of kf_smoother on the filtered tapes of the same call (C2: H = 0
    makes P_hat singular -- the RTS kernels' pinv case, compared at 1e-6 of the states only)."""
    import torch

    w = {"headline": lambda: W.structural(B=B, n=n), "c3": lambda: W.structural(B=B, n=n), "c2": lambda: W.arma11(B=B, n=n)}[name]()
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    res = eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False)
    sa, sP = eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)
    res.predicted_states = res.predicted_covs = None
    ra, rP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
    torch.cuda.synchronize()

    def dev(x, r):
        return float(((x - r).abs().reshape(B, -1).max(dim=1).values / (1e-300 + r.abs().reshape(B, -1).max(dim=1).values)).max())

    ea, eP = dev(sa, ra), dev(sP, rP)
    if name == "c2":
        assert ea < 1e-6, ea
    else:
        assert ea < 1e-11 and eP < 1e-11, (ea, eP)


def _fwd_np(res):
    return dict(logp=res.logp.cpu().numpy(), ll_obs=res.ll_obs.cpu().numpy(), filtered_states=res.filtered_states.cpu().numpy(),
                predicted_states=res.predicted_states.cpu().numpy(), filtered_covs=res.filtered_covs.cpu().numpy(),
                predicted_covs=res.predicted_covs.cpu().numpy(), info=res.info.cpu().numpy())


@pytest.mark.parametrize("ft", ["standard", "univariate"])
@pytest.mark.parametrize("tv", [("Q",), ("R",), ("R", "Q"), ("T", "Q"), ("T", "Z", "R", "H", "Q", "c", "d")])
@pytest.mark.parametrize("m,r", [(1, 1), (3, 2), (6, 3), (10, 4), (7, 1)])
def test_time_varying_R_Q_on_the_block_kernel(eng, monkeypatch, ft, tv, m, r):
    """VERDICT round 4, missing #5: R_t Q_t R_t' per step as four MFMAs on the slices (headline shape with a time-varying Q:
    0.039 of the HBM peak on the LDS kernel).  Against the oracle at 1e-8 / 1e-9 and against the LDS kernel; missing observations."""
    B, n = 9, 70
    w = W.random_dense(B, n, 1, m, r, seed=700 + 10 * m + r)
    rng = np.random.default_rng(m + r)
    mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"],
                c=0.1 * rng.standard_normal((B, m)), d=0.1 * rng.standard_normal((B, 1)))
    for name in tv:
        x = np.repeat(mats[name][:, None], n, axis=1)
        wob = 0.8 + 0.2 * np.cos(0.37 * np.arange(n) + len(name))
        mats[name] = x * wob.reshape((1, n) + (1,) * (x.ndim - 2))
        if name in ("Z", "c", "d", "R"):
            mats[name] = mats[name] + 0.05 * rng.standard_normal(mats[name].shape)
    y = w["y"].copy()
    y[[4, 33, 34]] = np.nan
    args = (y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"], mats["c"], mats["d"])
    assert eng.forward_kernel_name(B, n, 1, m, r, filter_type=ft, time_varying=tv) == "kf_fwd_b4tv_kernel"
    got = _fwd_np(eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv))
    ref = kn.kalman_filter_time_varying(*args, time_varying=tv, filter_type=ft, mvn_const="exact")
    monkeypatch.setenv("KF_FORWARD_B4TV", "0")
    assert eng.forward_kernel_name(B, n, 1, m, r, filter_type=ft, time_varying=tv) == "kf_fwd_lds_kernel"
    lds = _fwd_np(eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv))
    monkeypatch.delenv("KF_FORWARD_B4TV")
    assert not got["info"].any()
    for other, tag in ((ref, "oracle"), (lds, "lds kernel")):
        assert np.max(np.abs(got["logp"] - other["logp"])) < 1e-8, tag
        for k in ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs"):
            assert_allclose(got[k], other[k], rtol=1e-9, atol=1e-9, err_msg=f"{tag} {k}")


def test_time_varying_Q_headline_shape_property(eng):
    """Full headline size with a time-varying Q whose slices are all the static Q: bit for bit... no -- the static kernel averages
    R Q R' over both orders; within 1e-12 of the static call's log-likelihoods and 1e-11 of its tapes."""
    import torch

    w = W.structural(B=4096, n=1000)
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    st = eng.forward(*d, mvn_const="exact", exact_recursion=True)
    Qt = d[7][:, None].expand(4096, 1000, 4, 4).contiguous()
    tv = eng.forward(d[0], d[1], d[2], d[3], d[4], d[5], d[6], Qt, mvn_const="exact", time_varying=("Q",))
    torch.cuda.synchronize()
    assert float(((tv.logp - st.logp).abs() / st.logp.abs()).max()) < 1e-12
    assert float((tv.predicted_covs - st.predicted_covs).abs().max()) < 1e-11
    assert float((tv.filtered_states - st.filtered_states).abs().max()) < 1e-10


@pytest.mark.parametrize("case", ["seasonal_m151", "seasonal_m367", "dense_m130", "seasonal_m151_dense_rows"])
def test_big_path_skips_zero_blocks_of_T_bit_for_bit(eng, monkeypatch, case):
    """The reference's seasonal / companion T (the daily-seasonality notebook, models/SARIMAX.py:59-98) is a shifted identity with
    a few dense rows: most 64 x 32 blocks are zero, and their K chunks add exact zeros.  Skipping them must not change a bit."""
    if case == "seasonal_m151":
        w = W.daily_seasonal(B=5, n=6, s=150)
    elif case == "seasonal_m367":
        w = W.daily_seasonal(B=3, n=4)
    elif case == "dense_m130":
        w = W.random_dense(3, 5, 1, 130, 3, seed=12)
    else:
        w = W.daily_seasonal(B=4, n=5, s=150)
        rng = np.random.default_rng(3)
        w["T"] = w["T"].copy()
        w["T"][:, 70, :] = 0.01 * rng.standard_normal((4, w["m"]))     # a dense row in the second 64-row block
        w["T"][1, :, 140] = 0.01 * rng.standard_normal(w["m"])          # a dense column in one draw only: per-draw maps
    assert w["m"] > 96
    args = _args(w)
    got = _fwd_np(eng.forward(*args, mvn_const="exact"))
    monkeypatch.setenv("KF_BIG_TMASK", "0")
    ref = _fwd_np(eng.forward(*args, mvn_const="exact"))
    monkeypatch.delenv("KF_BIG_TMASK")
    for k in got:
        assert np.array_equal(got[k], ref[k]), k
    orc = kn.kalman_filter(*args, mvn_const="exact")
    assert np.max(np.abs(got["logp"] - orc["logp"])) <= 1e-8 * max(1.0, np.abs(orc["logp"]).max() / 1e4)
    assert_allclose(got["predicted_covs"], orc["predicted_covs"], rtol=1e-9, atol=1e-9)
    # logp only (the P_f-on-the-way-into-LDS product; KF_BIG_SPARSE=0: the MFMA path for a sparse T too): the same log-likelihoods
    monkeypatch.setenv("KF_BIG_SPARSE", "0")
    lo = eng.forward(*args, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
    assert eng.last_forward_kernel() == "big_gemm_kernel"
    assert np.max(np.abs(lo.logp.cpu().numpy() - orc["logp"])) <= 1e-8 * max(1.0, np.abs(orc["logp"]).max() / 1e4)
    monkeypatch.setenv("KF_BIG_TMASK", "0")
    lo0 = eng.forward(*args, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
    monkeypatch.delenv("KF_BIG_TMASK")
    assert np.array_equal(lo.logp.cpu().numpy(), lo0.logp.cpu().numpy())


def test_big_gradient_and_rts_smoother_skip_zero_blocks_of_T_bit_for_bit(eng, monkeypatch):
    """The gradient sweep's two products by T (X1 = Pb T, Sb = X1' T: T the untransposed B operand, block map by columns) and the
    RTS smoother's (W = T P_f, P_hat = W T' + R Q R') walk only the K chunks whose block of T holds a non-zero: the same bits as
    every chunk (KF_BIG_TMASK=0), and the oracles' numbers are held by tests/test_gpu_c5.py."""
    w = W.daily_seasonal(B=3, n=6, s=150)
    rng = np.random.default_rng(4)
    w["T"] = w["T"].copy()
    w["T"][1, :, 100] = 0.01 * rng.standard_normal(w["m"])     # a dense column in one draw: per-draw maps
    args = _args(w)
    res = eng.forward(*args, mvn_const="exact")

    def run():
        g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=np.ones(w["B"]),
                         mvn_const="exact")
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        return {k: v.cpu().numpy() for k, v in g.items()}, sa.cpu().numpy(), sP.cpu().numpy()

    g1, sa1, sP1 = run()
    monkeypatch.setenv("KF_BIG_TMASK", "0")
    g0, sa0, sP0 = run()
    for k in g1:
        assert np.array_equal(g1[k], g0[k]), k
    assert np.array_equal(sa1, sa0) and np.array_equal(sP1, sP0)


@pytest.mark.parametrize("tv", [("T",), ("Z",), ("T", "Z", "H", "c", "d")], ids=lambda t: "".join(t))
@pytest.mark.parametrize("pm", [(1, 3), (1, 10), (1, 15), (3, 10), (5, 16)], ids=lambda x: f"p{x[0]}m{x[1]}")
def test_time_varying_gradient_on_the_tile_adjoint(eng, monkeypatch, tv, pm):
    """T / Z / H / c / d time-varying (R, Q static), m <= 16, scalar observations: kf_bwd_mfma_kernel<.., TV> with per-step slices
    and gradient tapes -- against the LDS adjoint it replaces (KF_BACKWARD_TV_MFMA=0) and torch autograd; a missing step, per-step
    cotangents, 70 draws x 90 steps (more than one 64-step block of y)."""
    from oracle import kalman_torch as kt

    p, m = pm
    B, n, r = 70, 90, 2
    w = W.random_dense(B, n, p, m, r, seed=5 + m + p)
    rng = np.random.default_rng(3)
    mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"], c=0.1 * rng.standard_normal((B, m)), d=0.1 * rng.standard_normal((B, p)))
    for k, name in enumerate(tv):
        x = np.repeat(mats[name][:, None], n, axis=1)
        wave = (0.8 + 0.2 * np.cos(np.arange(n) + k))[None, :]
        mats[name] = x * wave.reshape(wave.shape + (1,) * (x.ndim - 2))
    y = w["y"].copy()
    y[[5, 63, 64]] = np.nan
    gl, gll = rng.standard_normal(B), 0.3 * rng.standard_normal((B, n))
    args = (y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"], mats["c"], mats["d"])
    kw = dict(mvn_const="exact", time_varying=tv, h_diagonal=p > 1)
    res = eng.forward(*args, **kw)
    bkw = dict(predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl, g_ll_obs=gll, want_cd=True, **kw)
    g = {k: v.cpu().numpy() for k, v in eng.backward(*args, **bkw).items()}
    monkeypatch.setenv("KF_BACKWARD_TV_MFMA", "0")
    g0 = {k: v.cpu().numpy() for k, v in eng.backward(*args, **bkw).items()}
    monkeypatch.delenv("KF_BACKWARD_TV_MFMA")
    _, _, gt = kt.logp_and_grads(*args, mvn_const="exact", g_logp=gl, g_ll_obs=gll, time_varying=tv)
    for k in GRADS:
        assert g[k].shape == g0[k].shape == np.asarray(gt[k]).shape, k
        a, b0, bt = g[k], g0[k], np.asarray(gt[k])
        if p > 1 and k in ("P0", "H"):   # (scalar-observation adjoints: symmetric part of g_P0, diagonal of g_H)
            a, b0, bt = (0.5 * (x + np.swapaxes(x, -1, -2)) for x in (a, b0, bt))
            if k == "H":
                a, b0, bt = (np.diagonal(x, axis1=-2, axis2=-1) for x in (a, b0, bt))
        assert np.max(np.abs(a - b0)) / (1.0 + np.max(np.abs(b0))) < 1e-10, (k, "LDS adjoint")
        assert np.max(np.abs(a - bt)) / (1.0 + np.max(np.abs(bt))) < 1e-9, (k, "autograd")


def _sparse_case(case):
    """Models whose T is a few non-zeros per row: what the reference's structural components assemble (seasonal dummies: a row of
    -1 over a shifted identity; trigonometric seasonality: 2 x 2 rotations; ARMA: a companion column over a shifted identity)."""
    rng = np.random.default_rng(sum(map(ord, case)))
    if case == "seasonal_m70":
        return W.daily_seasonal(B=11, n=40, s=69), True
    if case == "seasonal_m367":
        return W.daily_seasonal(B=3, n=12), True
    if case in ("two_seasonals_m150", "five_dense_rows_m150", "one_draw_dense_m150"):
        w = W.daily_seasonal(B=4, n=30, s=80)   # m = 81: level, trend, 79 seasonal states
        m = 150
        T = np.zeros((4, m, m))
        T[:, :81, :81] = w["T"]
        T[:, 81, 81:] = -1.0                    # a second seasonal block (period 70): another dense row
        idx = np.arange(m - 82)
        T[:, 82 + idx, 81 + idx] = 1.0
        Z = np.zeros((4, 1, m)); Z[:, 0, :81] = w["Z"][:, 0]; Z[:, 0, 81] = 1.0
        R = np.zeros((4, m, 4)); R[:, :81, :3] = w["R"]; R[:, 81, 3] = 1.0
        Q = np.zeros((4, 4, 4)); Q[:, :3, :3] = w["Q"]; Q[:, 3, 3] = 0.05 + rng.random(4)
        if case == "five_dense_rows_m150":      # more dense rows than the CSR path keeps aside: the MFMA products
            for row in (100, 110, 120):
                T[:, row, 90:140] += 0.002 * rng.standard_normal((4, 50))
        if case == "one_draw_dense_m150":       # one draw's T does not fit: the whole call takes the MFMA products
            T[2] += 0.001 * rng.standard_normal((m, m))
        P0 = np.broadcast_to(np.eye(m), (4, m, m)).copy()
        out = dict(B=4, n=30, p=1, m=m, r=4, y=w["y"], a0=np.zeros((4, m)), P0=P0, T=T, Z=Z, R=R, H=w["H"], Q=Q)
        return out, case == "two_seasonals_m150"
    if case == "companion_trig_m96":            # ARMA companion (dense first COLUMN: two non-zeros per row) + rotations + random extras
        B, m, n = 5, 96, 50
        T = np.zeros((B, m, m))
        T[:, :40, 0] = 0.5 * rng.standard_normal((B, 40)) / np.arange(1, 41)
        i = np.arange(39)
        T[:, i, i + 1] = 1.0
        for k in range(28):
            lam = 2 * np.pi * (k + 1) / 60.0
            a = 40 + 2 * k
            T[:, a, a], T[:, a, a + 1], T[:, a + 1, a], T[:, a + 1, a + 1] = np.cos(lam), np.sin(lam), -np.sin(lam), np.cos(lam)
        T[:, 95, 95] = 0.7
        T[:, 90, 3] = 0.1                                                   # stray entries: no special structure is assumed
        T[:, 17, 80] = -0.05
        w = W.random_dense(B, n, 1, m, 5, seed=77)
        w["T"] = T
        Z = np.zeros((B, 1, m)); Z[:, 0, 0] = 1.0; Z[:, 0, 40::2] = 1.0
        w["Z"] = Z
        return w, True
    if case == "ten_per_row_m100":               # short rows, but 10 m entries in all: over the CSR's 4 m -> the MFMA products
        w = W.random_dense(3, 20, 1, 100, 3, seed=79)
        keep = np.zeros((100, 100), dtype=bool)
        for k in range(10):
            keep[np.arange(100), (np.arange(100) * 7 + 11 * k) % 100] = True
        w["T"] = w["T"] * keep
        return w, False
    if case == "sparse_m1000":
        B, m, n = 2, 1000, 3
        w = W.random_dense(B, n, 1, m, 2, seed=78)
        T = np.zeros((B, m, m))
        i = np.arange(m - 1)
        T[:, i + 1, i] = 0.9
        T[:, 0, :] = -0.02 * rng.random((B, m))
        T[:, 500, 100:700] = 0.01 * rng.standard_normal((B, 600))
        w["T"] = T
        w["y"] = np.nan_to_num(w["y"], nan=0.2)
        return w, True
    raise KeyError(case)


@pytest.mark.parametrize("case", ["seasonal_m70", "seasonal_m367", "two_seasonals_m150", "five_dense_rows_m150", "one_draw_dense_m150",
                                  "companion_trig_m96", "ten_per_row_m100", "sparse_m1000"])
def test_workspace_forward_sparse_transition(eng, monkeypatch, case):
    """m > 64, p = 1, no filtered covariance tape: a T of a few non-zeros per row runs T P_f T' + R Q R' as a CSR gather
    (bsp_step_kernel: one pass over P per step) instead of two MFMA products.  Against the oracle, against the MFMA path
    (KF_BIG_SPARSE=0), with missing observations, with and without the predicted tapes; a T that does not fit the CSR (more than
    four dense rows, or one dense draw) takes the MFMA products."""
    w, sparse = _sparse_case(case)
    y = w["y"].copy()
    if y.shape[0] > 20:
        y[[0, 7, 19]] = np.nan
    args = (y,) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q"))
    orc = kn.kalman_filter(*args, mvn_const="exact")
    tol = 1e-8 * max(1.0, np.abs(orc["logp"]).max() / 1e4)
    for predicted in (True, False):
        kw = dict(mvn_const="exact", store_filtered=False, store_predicted=predicted)
        res = eng.forward(*args, **kw)
        on_workspace = predicted or w["m"] > 96   # (64 < m <= 96 without covariance tapes: kf_fwd_wgl_kernel, P resident in LDS)
        if on_workspace:
            assert eng.last_forward_kernel() == ("bsp_step_kernel" if sparse else "big_gemm_kernel"), case
        assert not res.info.cpu().numpy().any()
        assert np.max(np.abs(res.logp.cpu().numpy() - orc["logp"])) < tol
        assert_allclose(res.ll_obs.cpu().numpy(), orc["ll_obs"], rtol=1e-9, atol=1e-9)
        if predicted:
            assert_allclose(res.predicted_states.cpu().numpy(), orc["predicted_states"], rtol=1e-9, atol=1e-10)
            pc = res.predicted_covs.cpu().numpy()
            assert_allclose(pc, orc["predicted_covs"], rtol=1e-9, atol=1e-10)
            # symmetric up to rounding (every element is its own short sum; the MFMA path mirrors one triangle)
            assert np.max(np.abs(pc - np.swapaxes(pc, -1, -2))) <= 1e-13 * np.abs(pc).max()
        monkeypatch.setenv("KF_BIG_SPARSE", "0")
        ref = eng.forward(*args, **kw)
        assert eng.last_forward_kernel() == "big_gemm_kernel"   # (also what the previous workspace call left, when this one is not)
        monkeypatch.delenv("KF_BIG_SPARSE")
        assert_allclose(res.logp.cpu().numpy(), ref.logp.cpu().numpy(), rtol=1e-11, atol=1e-9)
    # a call that asks for the filtered covariance tape keeps the products (P_f has to be written)
    eng.forward(*args, mvn_const="exact")
    assert eng.last_forward_kernel() == "big_gemm_kernel"


def test_workspace_forward_sparse_transition_all_missing_and_one_step(eng):
    w = W.daily_seasonal(B=3, n=1, s=100)
    w["y"] = np.full((1, 1), 0.3)
    args = (w["y"],) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q"))
    orc = kn.kalman_filter(*args, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact", store_filtered=False)
    assert eng.last_forward_kernel() == "bsp_step_kernel"
    assert_allclose(res.logp.cpu().numpy(), orc["logp"], rtol=1e-11, atol=1e-12)
    assert_allclose(res.predicted_covs.cpu().numpy(), orc["predicted_covs"], rtol=1e-11, atol=1e-12)
    w = W.daily_seasonal(B=9, n=6, s=100)
    y = np.full_like(w["y"], np.nan)
    args = (y,) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q"))
    orc = kn.kalman_filter(*args, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact", store_filtered=False)
    assert np.array_equal(res.logp.cpu().numpy(), np.zeros(9))
    assert_allclose(res.predicted_covs.cpu().numpy(), orc["predicted_covs"], rtol=1e-11, atol=1e-12)


GRADS = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")


def _rel_grad_err(got, ref):
    worst = {}
    for k in GRADS:
        a = got[k].cpu().numpy() if hasattr(got[k], "cpu") else np.asarray(got[k])
        r = np.asarray(ref[k]).reshape(a.shape)
        if k in ("P0", "H", "Q"):   # symmetric inputs: only the symmetric part of an entrywise gradient is defined by the function
            a = 0.5 * (a + np.swapaxes(a, -1, -2))
            r = 0.5 * (r + np.swapaxes(r, -1, -2))
        B = a.shape[0]
        scale = np.maximum(np.abs(r).reshape(B, -1).max(axis=1), 1e-300)
        worst[k] = float((np.abs(a - r).reshape(B, -1).max(axis=1) / scale).max())
    return worst


@pytest.mark.parametrize("ft", ["standard", "cholesky"])
@pytest.mark.parametrize("shape", [(2, 4, 2), (3, 8, 3), (5, 10, 5), (5, 15, 5), (8, 16, 4), (4, 12, 4)])
def test_dense_H_value_and_grad_by_whitening(eng, monkeypatch, ft, shape):
    """VERDICT round 4, missing #4.  Whole-row missing steps, intercepts, a per-draw cotangent."""
    from oracle import kalman_torch as kt

    p, m, r = shape
    B, n = 12, 70
    w = W.random_dense(B=B, n=n, p=p, m=m, r=r, seed=300 + 7 * p + m, h_diag=False)
    y = w["y"].copy()
    y[[4, 30, 31]] = np.nan
    rng = np.random.default_rng(p + m)
    c, d = 0.05 * rng.standard_normal((B, m)), 0.1 * rng.standard_normal((B, p))
    gl = 1.0 + 0.3 * rng.standard_normal(B)
    args = (y,) + tuple(w[k] for k in KEYS)
    lp, info, g = eng.value_and_grad(*args, c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
    assert not info.cpu().numpy().any()
    rlp, _, rg = kt.logp_and_grads(*args, c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
    assert np.all(np.abs(lp.cpu().numpy() - rlp) <= 1e-8 * np.maximum(1.0, np.abs(rlp) / 1e4))
    err = _rel_grad_err(g, rg)
    assert max(err.values()) < 1e-9, err
    # g_H is the FULL p x p gradient (off-diagonal entries too): not the diagonal a scalar-observation adjoint sees
    gH = g["H"].cpu().numpy()
    assert np.abs(gH - np.swapaxes(gH, 1, 2)).max() <= 1e-12 * np.abs(gH).max()
    assert np.abs(gH[:, 0, 1]).min() > 0.0
    # the path it replaces: the dense LDS adjoint
    monkeypatch.setenv("KF_FORWARD_WHITEN", "0")
    lp0, _, g0 = eng.value_and_grad(*args, c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
    monkeypatch.delenv("KF_FORWARD_WHITEN")
    assert np.all(np.abs(lp.cpu().numpy() - lp0.cpu().numpy()) <= 1e-8 * np.maximum(1.0, np.abs(rlp) / 1e4))
    err0 = _rel_grad_err(g, {k: g0[k].cpu().numpy() for k in GRADS})
    assert max(err0.values()) < 1e-8, err0


def test_dense_H_value_and_grad_falls_back_where_whitening_does_not_apply(eng):
    """A partially missing row / an H that is not positive definite: the dense LDS kernels redo the batch behind the whitened
    call (device flag, no host synchronisation) -- value and gradients are the dense path's."""
    from oracle import kalman_torch as kt

    w = W.random_dense(B=10, n=50, p=3, m=6, r=3, seed=5, h_diag=False)
    y = w["y"].copy()
    y[7, 1] = np.nan                      # one cell of a row
    y[20] = np.nan
    args = (y,) + tuple(w[k] for k in KEYS)
    lp, info, g = eng.value_and_grad(*args, filter_type="standard", mvn_const="exact")
    rlp, _, rg = kt.logp_and_grads(*args, filter_type="standard", mvn_const="exact")
    assert np.all(np.abs(lp.cpu().numpy() - rlp) <= 1e-8 * np.maximum(1.0, np.abs(rlp) / 1e4))
    err = _rel_grad_err(g, rg)
    assert max(err.values()) < 1e-8, err
    # the next call on complete data takes the whitened path again (the flag is per call)
    lp2, _, g2 = eng.value_and_grad(w["y"], *[w[k] for k in KEYS], filter_type="standard", mvn_const="exact")
    rlp2, _, rg2 = kt.logp_and_grads(w["y"], *[w[k] for k in KEYS], filter_type="standard", mvn_const="exact")
    assert max(_rel_grad_err(g2, rg2).values()) < 1e-9
    # H not positive definite in one draw (F may still be: H only enters through Z P Z' + H)
    w2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    w2["H"][4] = 0.0
    a2 = (w["y"],) + tuple(w2[k] for k in KEYS)
    lp3, _, g3 = eng.value_and_grad(*a2, filter_type="standard", mvn_const="exact")
    rlp3, _, rg3 = kt.logp_and_grads(*a2, filter_type="standard", mvn_const="exact")
    assert np.all(np.abs(lp3.cpu().numpy() - rlp3) <= 1e-8 * np.maximum(1.0, np.abs(rlp3) / 1e4))
    assert max(_rel_grad_err(g3, rg3).values()) < 1e-8


def test_dense_H_value_and_grad_c4_shard(eng):
    """The C4 shape (VARMAX (2,1): p = 5, m = 15) with a dense H, 2048 draws x 300 steps: the steady-state adjoint behind the
    multi-observation block kernel takes the whitened model; against the dense LDS adjoint on a sample of draws."""
    import torch

    w = W.varmax(B=2048, n=300, order=(2, 1))
    rng = np.random.default_rng(0)
    A = 0.05 * rng.standard_normal((2048, 5, 5))
    H = w["H"] + A @ np.swapaxes(A, 1, 2)
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R")] + [eng.to_device(H), eng.to_device(w["Q"])]
    lp, info, g = eng.value_and_grad(*d, mvn_const="exact")
    assert not info.cpu().numpy().any()
    idx = torch.arange(0, 2048, 64)
    os_env = __import__("os").environ
    os_env["KF_FORWARD_WHITEN"] = "0"
    try:
        lp0, _, g0 = eng.value_and_grad(d[0], *[x[idx] for x in d[1:]], mvn_const="exact")
    finally:
        del os_env["KF_FORWARD_WHITEN"]
    assert float(((lp[idx] - lp0).abs() / lp0.abs()).max()) < 1e-11
    err = _rel_grad_err({k: g[k][idx] for k in GRADS}, {k: g0[k].cpu().numpy() for k in GRADS})
    assert max(err.values()) < 1e-8, err


def test_ill_conditioned_dense_H_goes_to_the_dense_path(eng):
    """an H with a Cholesky pivot below 1e-8 of its largest diagonal entry (cond ~ 1e10) is not whitened -- L^-1 would
    have norm ~1e5 -- but handed to the dense path by the same device flag as an H that is not positive definite; forward tapes and
    value-and-gradient against the oracle."""
    from oracle import kalman_torch as kt

    w = W.random_dense(B=6, n=40, p=3, m=6, r=3, seed=11, h_diag=False)
    H = w["H"].copy()
    u = np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0)
    for b in (1, 4):   # two draws: H = I + (1e-10 - 1) u u' in the (y1, y2) plane -> eigenvalues 1, 1, 1e-10
        H[b] = np.eye(3) + (1e-10 - 1.0) * np.outer(u, u)
    args = (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])
    res = eng.forward(*args, mvn_const="exact")
    ref = kn.kalman_filter(*args, mvn_const="exact")
    assert np.all(np.abs(res.logp.cpu().numpy() - ref["logp"]) <= 1e-8 * np.maximum(1.0, np.abs(ref["logp"]) / 1e4))
    assert_allclose(res.filtered_states.cpu().numpy(), ref["filtered_states"].reshape(6, 40, 6), rtol=1e-8, atol=1e-8)
    lp, info, g = eng.value_and_grad(*args, mvn_const="exact")
    rlp, _, rg = kt.logp_and_grads(*args, mvn_const="exact")
    assert np.all(np.abs(lp.cpu().numpy() - rlp) <= 1e-8 * np.maximum(1.0, np.abs(rlp) / 1e4))
    err = _rel_grad_err(g, rg)
    assert max(v for k, v in err.items() if k != "H") < 1e-7, err   # (g_H of the near-singular draws: ~1e10-sized entries, 1e-6 relative)
    assert err["H"] < 1e-5, err


@pytest.mark.parametrize("case", ["dense_m16", "dense_m40", "dense_m70", "seasonal_m101", "seasonal_m151_missing", "seasonal_m367",
                                  "companion_m24", "companion_m70"])
def test_smoother_from_filter_workspace_path(eng, monkeypatch, case):
    """15 < m <= 1024 (kf_smoother_bf_big.hip; BASELINE config 5's smoother): the same recursion with the fp64-MFMA products of the
    workspace path -- two dense, two with the block-sparse T -- against its numpy restatement (1e-9 of the draw's largest filtered
    entry) and against kf_smoother on the filtered tapes; with every K chunk (KF_BIG_TMASK=0) bit for bit.
    (Round 6: 16 <= m <= 79 run on chip by default -- kf_smoother_bf_wg.hip, tests/test_gpu_r06.py --; KF_SMOOTHER_BF_WG=0 keeps this path.)"""
    monkeypatch.setenv("KF_SMOOTHER_BF_WG", "0")
    if case.startswith("dense"):
        m = int(case.split("_m")[1])
        w = W.random_dense(4, 12, 1, m, min(m, 5), seed=40 + m)
    elif case.startswith("companion"):   # ARMA companion form: a dense first COLUMN of T -- a row of T' with m entries (CSR up to 32,
        m = int(case.split("_m")[1])     # a "dense row" beyond: the block-sparse products) -- over a shifted identity; rank-one noise
        w = W.random_dense(4, 12, 1, m, 1, seed=60 + m)
        T = np.zeros_like(w["T"])
        T[:, :, 0] = 0.5 * np.random.default_rng(m).standard_normal((4, m)) / np.arange(1, m + 1)
        T[:, np.arange(m - 1), np.arange(1, m)] = 1.0
        w["T"] = T
    else:
        m = int(case.split("_m")[1].split("_")[0])
        w = W.daily_seasonal(B=3, n=9 if m < 300 else 5, s=m - 1)
    assert w["m"] == m
    y = w["y"].copy()
    if "missing" in case:
        y[[2, 8]] = np.nan
    res = eng.forward(*_args(w, y), mvn_const="exact")
    assert not res.info.cpu().numpy().any()
    sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy())
    scale = max(1.0, float(np.abs(res.filtered_covs.cpu().numpy()).max()))
    assert_allclose(sa.cpu().numpy(), da, rtol=1e-9, atol=1e-9 * scale)
    assert_allclose(sP.cpu().numpy(), dP, rtol=1e-9, atol=1e-9 * scale)
    if not case.startswith("companion"):   # (rank-one noise: the RTS recursion's pinv(P_hat) is another function of the tapes there)
        ra, rP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        assert_allclose(sa.cpu().numpy(), ra.cpu().numpy(), rtol=1e-7, atol=1e-7 * scale)
        assert_allclose(sP.cpu().numpy(), rP.cpu().numpy(), rtol=1e-7, atol=1e-7 * scale)
    monkeypatch.setenv("KF_BIG_TMASK", "0")
    sa0, sP0 = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    monkeypatch.delenv("KF_BIG_TMASK")
    assert np.array_equal(sa.cpu().numpy(), sa0.cpu().numpy()) and np.array_equal(sP.cpu().numpy(), sP0.cpu().numpy())
    # T' M T as a CSR gather (a sparse T' without dense rows: the seasonal models, companion forms up to m = 32) against the two
    # block-sparse products (KF_BIG_SPARSE=0), which in turn skip no chunk with KF_BIG_TMASK=0
    monkeypatch.setenv("KF_BIG_SPARSE", "0")
    sa1, sP1 = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    monkeypatch.setenv("KF_BIG_TMASK", "0")
    sa2, sP2 = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    assert np.array_equal(sa1.cpu().numpy(), sa2.cpu().numpy()) and np.array_equal(sP1.cpu().numpy(), sP2.cpu().numpy())
    assert_allclose(sa.cpu().numpy(), sa1.cpu().numpy(), rtol=1e-10, atol=1e-10 * scale)
    assert_allclose(sP.cpu().numpy(), sP1.cpu().numpy(), rtol=1e-10, atol=1e-10 * scale)
    if case.startswith("dense") or case == "companion_m70":
        assert np.array_equal(sP.cpu().numpy(), sP1.cpu().numpy())   # (no CSR form: the same products either way)
