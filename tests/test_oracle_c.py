"""Pin the plain-C restatement (oracle/kf_oracle.c) to the numpy oracle and, through the same
golden files, to statsmodels.  The C oracle is what full-size GPU parity runs and the
``cpu_baseline`` leg of bench.py use."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from conftest import golden_files, h_is_diagonal, load_golden
from oracle import c_oracle as co
from oracle import kalman_numpy as kn

KEYS = ["logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"]


@pytest.mark.parametrize("name", golden_files())
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_c_matches_numpy_on_golden_inputs(name, ft):
    g = load_golden(name)
    p = g["Z"].shape[1]
    if ft == "single" and p != 1:
        pytest.skip("single is p == 1 only")
    if ft == "univariate" and not h_is_diagonal(g["H"]):
        pytest.skip("univariate uses diag(H) only")
    args = (g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"])
    a = co.kalman_filter(*args, filter_type=ft, mvn_const="exact")
    b = kn.kalman_filter(*args, filter_type=ft, mvn_const="exact", cholesky_triangle="correct",
                         partial_missing="delete")
    tol = 1e-8 if name.startswith("nile_llt") else 1e-11
    for k in KEYS:
        assert_allclose(a[k], b[k], rtol=tol, atol=tol, err_msg=f"{name} {ft} {k}")
    assert_allclose(a["ll_obs"], g["llf_obs"], rtol=max(tol, 2e-9), atol=max(tol, 2e-9))


@pytest.mark.parametrize("mvn", ["reference", "exact"])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate"])
def test_c_mvn_const_modes(ft, mvn):
    w = W.random_dense(5, 30, 3, 5, 2, seed=21)
    y = w["y"].copy()
    y[4] = np.nan
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    a = co.kalman_filter(*args, filter_type=ft, mvn_const=mvn)
    b = kn.kalman_filter(*args, filter_type=ft, mvn_const=mvn, cholesky_triangle="correct")
    assert_allclose(a["logp"], b["logp"], rtol=1e-12)


@pytest.mark.parametrize("name", ["structural_m10.npz", "varmax_m10.npz", "dense_p3m4r2.npz",
                                  "nile_local_level.npz"])
def test_c_smoother_matches_numpy(name):
    g = load_golden(name)
    args = (g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"])
    f = co.kalman_filter(*args, mvn_const="exact")
    sa, sP = co.kalman_smoother(g["T"], g["R"], g["Q"], f["filtered_states"], f["filtered_covs"])
    na, nP = kn.kalman_smoother(g["T"], g["R"], g["Q"], f["filtered_states"], f["filtered_covs"])
    assert_allclose(sa, na, rtol=1e-9, atol=1e-9)
    assert_allclose(sP, nP, rtol=1e-9, atol=1e-9)
    assert_allclose(sa, g["smoothed_state"], rtol=1e-8, atol=1e-8)


def test_c_threads_agree():
    w = W.structural(B=16, n=50)
    args = (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    a = co.kalman_filter(*args, nthreads=1)
    b = co.kalman_filter(*args, nthreads=4)
    assert np.array_equal(a["logp"], b["logp"])
