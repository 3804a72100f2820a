"""Pin the CPU oracle (oracle/kalman_numpy.py) to golden vectors.

Mirrors the reference's parity test (reference tests/test_kalman_filter.py:226-241):
every filter x every output vs statsmodels, on the Nile local-linear-trend case with and
without missing rows -- plus the config-shaped and p > 1 cases the reference never pinned.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import golden_files, h_is_diagonal, load_golden
from oracle import kalman_numpy as kn

OUTPUTS = [
    ("filtered_states", "filtered_state"),
    ("predicted_states", "predicted_state"),
    ("filtered_covs", "filtered_state_cov"),
    ("predicted_covs", "predicted_state_cov"),
    ("ll_obs", "llf_obs"),
]


def _run(g, ft, **kw):
    return kn.kalman_filter(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"],
                            filter_type=ft, **kw)


def _ill_conditioned(name):
    # raw Nile flows with P0 = 1e6 I: |logp| ~ 6e5, independent fp64 codes differ ~1e-10 rel
    return name.startswith("nile_llt")


@pytest.mark.parametrize("name", golden_files())
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_filter_matches_statsmodels(name, ft):
    g = load_golden(name)
    p = g["Z"].shape[1]
    if ft == "single" and p != 1:
        with pytest.raises(AssertionError, match="at most 1-dimensional"):
            _run(g, ft)
        return
    if ft == "univariate" and not h_is_diagonal(g["H"]):
        pytest.skip("UnivariateFilter uses diag(H) only (reference kalman_filter.py:493)")
    out = _run(g, ft, mvn_const="exact", cholesky_triangle="correct", partial_missing="delete")
    rtol = 2e-9 if _ill_conditioned(name) else 1e-11
    for ours, theirs in OUTPUTS:
        assert_allclose(out[ours], g[theirs], rtol=rtol, atol=rtol, err_msg=f"{name} {ft} {ours}")
    assert_allclose(out["logp"], g["llf_obs"].sum(-1), rtol=rtol, atol=1e-9)


@pytest.mark.parametrize("name", golden_files())
def test_smoother_matches_statsmodels(name):
    g = load_golden(name)
    out = _run(g, "standard", mvn_const="exact", partial_missing="delete")
    sa, sP = kn.kalman_smoother(g["T"], g["R"], g["Q"], out["filtered_states"], out["filtered_covs"])
    if _ill_conditioned(name):
        # the reference's own test: atol 1e-7 on states, covariances only from index 5 on
        # (reference tests/test_kalman_filter.py:236-241)
        assert_allclose(sa, g["smoothed_state"], rtol=1e-7, atol=1e-7)
        assert_allclose(sP[:, 5:], g["smoothed_state_cov"][:, 5:], rtol=1e-7, atol=1e-7)
    elif name.startswith("arma11"):
        # H = 0: filtered covariances are exactly singular, the pinv gain is rank sensitive
        assert_allclose(sa, g["smoothed_state"], rtol=1e-7, atol=1e-7)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-4, atol=1e-4)
    else:
        assert_allclose(sa, g["smoothed_state"], rtol=1e-10, atol=1e-10)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-10, atol=1e-10)
    # reference tests/test_kalman_filter.py:212-222
    assert_allclose(sa[:, -1], out["filtered_states"][:, -1])
    assert_allclose(sP[:, -1], out["filtered_covs"][:, -1])


def test_known_answers_baseline_md():
    """BASELINE.md section 2 / SURVEY.md section 8(c) known answers."""
    g = load_golden("nile_llt.npz")
    for ft, want in [("standard", -606737.514026642), ("cholesky", -606737.514026642),
                     ("single", -606737.514026642), ("univariate", -606737.5140266517)]:
        out = _run(g, ft)
        assert_allclose(out["logp"][0], want, rtol=1e-13)
    out = _run(g, "standard")
    assert_allclose(out["ll_obs"][0, :4], [-8.45389371, -7.82749490, -4835.65907, -12435.9653],
                    rtol=1e-8)
    assert_allclose(out["filtered_states"][0, -1], [730.6659435031556, -17.64853509699647],
                    rtol=1e-9)
    gm = load_golden("nile_llt_missing.npz")
    out = _run(gm, "standard")
    assert_allclose(out["logp"][0], -517594.31467235665, rtol=1e-13)
    assert out["ll_obs"][0, 3] == 0 and out["ll_obs"][0, 17] == 0
    gl = load_golden("nile_local_level.npz")
    out = _run(gl, "standard")
    assert_allclose(out["logp"][0], -641.5855784594156, rtol=1e-13)
    assert_allclose(out["filtered_states"][0, -1, 0], 798.3702926083578, rtol=1e-12)
    assert_allclose(out["filtered_covs"][0, -1, 0, 0], 4032.1579418087827, rtol=1e-12)


def test_reference_quirk_switches():
    """A7 (single log 2pi) and A8 (wrong-triangle solve) only matter for p > 1."""
    g = load_golden("dense_p3m4r2.npz")
    p = g["Z"].shape[1]
    n = g["y"].shape[0]
    exact = _run(g, "standard", mvn_const="exact")
    ref = _run(g, "standard", mvn_const="reference")
    assert_allclose(ref["logp"] - exact["logp"], 0.5 * (p - 1) * n * kn.LOG_2PI, rtol=1e-12)
    chol_ok = _run(g, "cholesky", mvn_const="exact", cholesky_triangle="correct")
    chol_ref = _run(g, "cholesky", mvn_const="exact", cholesky_triangle="reference")
    assert_allclose(chol_ok["logp"], exact["logp"], rtol=1e-12)
    assert np.max(np.abs(chol_ref["logp"] - exact["logp"])) > 1e-3
    g1 = load_golden("structural_m10.npz")
    a = _run(g1, "cholesky", cholesky_triangle="reference")
    b = _run(g1, "cholesky", cholesky_triangle="correct")
    assert_allclose(a["logp"], b["logp"], rtol=1e-14)


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("pmrn", [(1, 1, 1, 10), (1, 2, 2, 10), (1, 5, 2, 10), (1, 5, 1, 10),
                                  (5, 5, 1, 10)])
def test_reference_shape_contract(ft, pmrn):
    """reference tests/test_kalman_filter.py:66-99,159-189 + get_expected_shape."""
    import workloads as W

    p, m, r, n = pmrn
    w = W.reference_test_inputs(p, m, r, n)
    args = (w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0],
            w["H"][0], w["Q"][0])
    if ft == "single" and p > 1:
        with pytest.raises(AssertionError, match="at most 1-dimensional"):
            kn.reference_outputs(*args, filter_type=ft)
        return
    outs = kn.reference_outputs(*args, filter_type=ft)
    shapes = [(n, m, 1), (n + 1, m, 1), (n, m, 1), (n, m, m), (n + 1, m, m), (n, m, m), (), (n,)]
    for o, s in zip(outs, shapes):
        assert np.shape(o) == s


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("p", [1, 5])
def test_missing_row_gives_no_nan(ft, p):
    """reference tests/test_kalman_filter.py:192-209 (whole-row NaN)."""
    import workloads as W

    if ft == "single" and p > 1:
        pytest.skip("single is p == 1 only")
    w = W.reference_test_inputs(p, 5, 1, 10, missing_rows=(4,))
    outs = kn.reference_outputs(w["y"][..., None], w["a0"][0][:, None], w["P0"][0], w["T"][0],
                                w["Z"][0], w["R"][0], w["H"][0], w["Q"][0], filter_type=ft)
    for o in outs:
        assert not np.any(np.isnan(o))


@pytest.mark.parametrize("name", golden_files())
def test_smoother_from_filter_matches_statsmodels(name):
    """The de Jong / Durbin-Koopman restatement (oracle/kalman_numpy.py::smoother_from_filter, the checker of
    kf_smoother_from_filter) against EVERY golden vector -- p > 1, partially missing rows and the H = 0 ARMA case included --
    at 1e-9, where the RTS restatement needs 1e-2 on arma11 (pinv's rank decision on an exactly singular P_hat)."""
    g = load_golden(name)
    out = _run(g, "standard", mvn_const="exact", partial_missing="delete")
    sa, sP = kn.smoother_from_filter(g["y"], g["T"], g["Z"], g["H"], out["predicted_states"], out["predicted_covs"])
    if _ill_conditioned(name):
        assert_allclose(sa, g["smoothed_state"], rtol=1e-7, atol=1e-7)
        assert_allclose(sP[:, 5:], g["smoothed_state_cov"][:, 5:], rtol=1e-7, atol=1e-7)
    else:
        assert_allclose(sa, g["smoothed_state"], rtol=1e-9, atol=1e-9)
        assert_allclose(sP, g["smoothed_state_cov"], rtol=1e-9, atol=1e-9)


def test_smoother_from_filter_equals_the_rts_restatement_where_it_is_well_conditioned():
    import workloads as W

    w = W.structural(B=5, n=90)
    y = w["y"].copy()
    y[[0, 7, 8, 89]] = np.nan
    d = 0.1 * np.arange(5.0)[:, None]
    for ft in ("standard", "single"):
        out = kn.kalman_filter(y, *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")], d=d, filter_type=ft)
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], out["filtered_states"], out["filtered_covs"])
        sa, sP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], out["predicted_states"], out["predicted_covs"], d=d,
                                         filter_type=ft)
        assert_allclose(sa, ra.reshape(sa.shape), rtol=1e-10, atol=1e-10)
        assert_allclose(sP, rP, rtol=1e-10, atol=1e-10)
