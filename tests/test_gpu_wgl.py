"""kf_forward_wgl.hip: forward filter for 64 < k_states <= 96 with the covariance resident in LDS (one workgroup per draw, scalar
observation, no covariance tapes).  Against the numpy oracle (kalman_filter.py:231-284 restated) and the HBM-workspace path it
replaces for these sizes (KF_FORWARD_WGL=0), every tile count and both kernel shapes (m <= 80: 5 x 5 tiles, m <= 96: 6 x 6)."""
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


@pytest.mark.parametrize("m", [65, 70, 79, 80, 81, 88, 96])
@pytest.mark.parametrize("filter_type", ["standard", "cholesky", "single"])
def test_lds_resident_forward_matches_oracle(eng, m, filter_type, monkeypatch):
    import torch

    from pymc_statespace_amd.engine import FilterResult

    if filter_type != "standard" and m not in (70, 96):
        pytest.skip("the filter variants differ in the intercept's sign and the constant only: two sizes are enough")
    B, n = 11, 150
    w = W.random_dense(B, n, 1, m, 4, seed=400 + m)
    rng = np.random.default_rng(m)
    y = w["y"].copy()
    y[[0, 17, 60, 149]] = np.nan
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    args = (y,) + tuple(w[k] for k in KEYS)
    ref = kn.kalman_filter(*args, c=c, d=d, filter_type=filter_type, mvn_const="exact")
    dev = torch.device("cuda")
    out = FilterResult(logp=torch.empty(B, device=dev, dtype=torch.float64),
                       ll_obs=torch.empty(B, n, device=dev, dtype=torch.float64),
                       filtered_states=torch.empty(B, n, m, device=dev, dtype=torch.float64),
                       predicted_states=torch.empty(B, n + 1, m, device=dev, dtype=torch.float64),
                       filtered_covs=None, predicted_covs=None, info=torch.empty(B, device=dev, dtype=torch.int32))
    res = eng.forward(*args, c=c, d=d, filter_type=filter_type, mvn_const="exact", out=out)
    assert not res.info.cpu().numpy().any()
    assert_allclose(res.logp.cpu().numpy(), ref["logp"], rtol=1e-11)
    assert_allclose(res.ll_obs.cpu().numpy(), ref["ll_obs"], rtol=1e-10, atol=1e-11)
    assert_allclose(res.filtered_states.cpu().numpy(), ref["filtered_states"], rtol=1e-9, atol=1e-10)
    assert_allclose(res.predicted_states.cpu().numpy(), ref["predicted_states"], rtol=1e-9, atol=1e-10)
    # the workspace path on the same inputs (what ran for these sizes before), and the launcher's choice
    monkeypatch.setenv("KF_FORWARD_WGL", "0")
    big = eng.forward(*args, c=c, d=d, filter_type=filter_type, mvn_const="exact", store_filtered=False, store_predicted=False)
    assert_allclose(res.logp.cpu().numpy(), big.logp.cpu().numpy(), rtol=1e-12)


def test_lds_resident_forward_long_series_and_per_draw_data(eng):
    """1000 steps at m = 80 (the size VERDICT round 2 named), data per draw, a non-positive F flagged like the other kernels flag it."""
    B, n, m = 6, 1000, 80
    w = W.random_dense(B, n, 1, m, 3, seed=77)
    rng = np.random.default_rng(3)
    y = np.repeat(w["y"][None], B, axis=0) + 0.1 * rng.standard_normal((B, n, 1))
    args = (y,) + tuple(w[k] for k in KEYS)
    ref = kn.kalman_filter(*args, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact", store_filtered=False, store_predicted=False)
    assert_allclose(res.logp.cpu().numpy(), ref["logp"], rtol=1e-10)
    assert_allclose(res.ll_obs.cpu().numpy(), ref["ll_obs"], rtol=1e-9, atol=1e-10)
    H = w["H"].copy()
    H[2] = -1e6   # F = z' P z + H < 0 at the first step of draw 2
    bad = eng.forward(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"], mvn_const="exact", store_filtered=False,
                      store_predicted=False)
    info = bad.info.cpu().numpy()
    assert info[2] == ((1 << 8) | 1) and not info[[0, 1, 3, 4, 5]].any()


@pytest.mark.parametrize("m,B,n", [(65, 1, 1), (80, 1, 2), (96, 3, 1), (72, 300, 5)])
def test_lds_resident_forward_edge_shapes(eng, m, B, n):
    """One draw, one / two steps, a batch larger than the chip's 256 CUs; a series that is missing from end to end."""
    w = W.random_dense(B, n, 1, m, m, seed=500 + m)   # r = m: a full-rank R Q R'
    w["y"] = np.nan_to_num(w["y"], nan=0.3)   # (a one-step series standardises to NaN)
    args = (w["y"],) + tuple(w[k] for k in KEYS)
    ref = kn.kalman_filter(*args, mvn_const="exact")
    res = eng.forward(*args, mvn_const="exact", store_filtered=False, store_predicted=False)
    assert not res.info.cpu().numpy().any()
    assert_allclose(res.logp.cpu().numpy(), ref["logp"], rtol=1e-11, atol=1e-12)
    assert_allclose(res.ll_obs.cpu().numpy(), ref["ll_obs"], rtol=1e-10, atol=1e-11)
    y = np.full_like(w["y"], np.nan)
    res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact", store_filtered=False, store_predicted=False)
    assert np.array_equal(res.logp.cpu().numpy(), np.zeros(B)) and np.array_equal(res.ll_obs.cpu().numpy(), np.zeros((B, n)))
