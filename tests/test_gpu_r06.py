"""Round-6 paths through the C-ABI against the oracle.

* kf_smoother_from_filter for m <= 4, one LANE per draw (csrc/kf_smoother_bf_lane.hip): KalmanSmoother.build_graph
  (pymc_statespace/filters/kalman_smoother.py:56-104) from the filter's PREDICTED tapes -- every p = 1 statsmodels golden vector
  at 1e-9 (arma11.npz: H = 0, k_posdef = 1, models/SARIMAX.py:23-49 -- the RTS kernels' pinv case, held to 1e-2 before), the numpy
  restatement at 1e-10, all three draws-per-wavefront variants, ragged batches, missing observations, d with both signs, one
  series per draw.
* reachability of that recursion for the callers that hold a filter's result (round-5 VERDICT, missing #3): ``BaseFilter.filter_and_smooth``,
  ``KalmanSmoother.__call__(..., filter_result=...)``, ``KalmanEngine.smooth_result`` and the fused PyTensor node's body all take it
  through ONE rule (engine.smoother_from_filter_applies) -- and all keep the reference's recursion when the state intercept c is not
  zero (the reference's smoother ignores c, kalman_smoother.py:99-104).
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


# ------------------------------------------------------------------------------------------------------------------------------------
# the lane kernel
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("B", [1, 7, 4099, 16390])   # 4 / 4 / 16 / 64 draws per wavefront (B <= 4096 / <= 16384 / beyond), all ragged
def test_lane_smoother_from_filter_matches_restatement(eng, m, B):
    n = 37 if B > 100 else 70   # (not a multiple of the 16-step LDS block)
    w = W.random_dense(B, n, 1, m, min(m, 2), seed=100 + m)
    y = w["y"].copy()
    y[[0, 5, 6, n - 1]] = np.nan
    res = eng.forward(*_args(w, y), mvn_const="exact")
    assert not res.info.cpu().numpy().any()
    sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy())
    scale = max(1.0, float(np.abs(res.filtered_covs.cpu().numpy()).max()))
    assert_allclose(sa.cpu().numpy(), da, rtol=1e-10, atol=1e-10 * scale)
    assert_allclose(sP.cpu().numpy(), dP, rtol=1e-10, atol=1e-10 * scale)
    # a missing LAST observation: smoothed == filtered == predicted there; otherwise last smoothed == last filtered
    assert_allclose(sP.cpu().numpy()[:, -1], res.filtered_covs.cpu().numpy()[:, -1], rtol=0, atol=1e-12 * scale)


def test_lane_and_tile_kernels_agree(eng, monkeypatch):
    """KF_SMOOTHER_BF_LANE=0 keeps round 5's block-tile kernel for m <= 4: the two implementations of one recursion."""
    w = W.arma11(B=300, n=200)
    res = eng.forward(*_args(w), mvn_const="exact")
    sa, sP = eng.smooth_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    monkeypatch.setenv("KF_SMOOTHER_BF_LANE", "0")
    ta, tP = eng.smooth_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    assert_allclose(sa.cpu().numpy(), ta.cpu().numpy(), rtol=1e-11, atol=1e-11)
    assert_allclose(sP.cpu().numpy(), tP.cpu().numpy(), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("ft", ["standard", "single", "univariate", "cholesky"])
def test_lane_smoother_from_filter_intercept_d_and_filter_types(eng, ft):
    w = W.random_dense(13, 50, 1, 2, 2, seed=31)
    d = 0.4 * np.random.default_rng(2).standard_normal((13, 1))
    res = eng.forward(*_args(w), d=d, filter_type=ft, mvn_const="exact")
    sa, sP = eng.smooth_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d, filter_type=ft)
    da, dP = kn.smoother_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(),
                                     res.predicted_covs.cpu().numpy(), d=d, filter_type=ft)
    assert_allclose(sa.cpu().numpy(), da, rtol=1e-10, atol=1e-10)
    assert_allclose(sP.cpu().numpy(), dP, rtol=1e-10, atol=1e-10)
    # and the reference's recursion on the filtered tapes of the same call (d never enters it)
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
    assert_allclose(sa.cpu().numpy(), ra.reshape(da.shape), rtol=1e-8, atol=1e-8)
    assert_allclose(sP.cpu().numpy(), rP, rtol=1e-8, atol=1e-8)


def test_lane_smoother_from_filter_y_per_draw(eng):
    w = W.random_dense(21, 48, 1, 3, 3, seed=8)
    yb = w["y"][None] + 0.1 * np.random.default_rng(9).standard_normal((21,) + w["y"].shape)
    res = eng.forward(yb, *[w[k] for k in KEYS], mvn_const="exact")
    sa, sP = eng.smooth_from_filter(yb, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    for b in (0, 20):
        da, dP = kn.smoother_from_filter(yb[b], w["T"][b:b + 1], w["Z"][b:b + 1], w["H"][b:b + 1],
                                         res.predicted_states[b:b + 1].cpu().numpy(), res.predicted_covs[b:b + 1].cpu().numpy())
        assert_allclose(sa[b:b + 1].cpu().numpy(), da, rtol=1e-10, atol=1e-10)
        assert_allclose(sP[b:b + 1].cpu().numpy(), dP, rtol=1e-10, atol=1e-10)


# ------------------------------------------------------------------------------------------------------------------------------------
# every p = 1 golden smoother at 1e-9 through the eager API a user calls (README quick use)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", _p1_golden())
def test_filter_and_smooth_matches_statsmodels_golden(name):
    from pymc_statespace_amd import KalmanSmoother, StandardFilter

    g = load_golden(name)
    flt = StandardFilter(mvn_const="exact")
    outs = flt.filter_and_smooth(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"])
    assert len(outs) == 8
    fs, ps, fc, pc, logp, ll, ss, sc = outs
    assert ss.shape == fs.shape and sc.shape == fc.shape
    assert_allclose(ll, g["llf_obs"], rtol=1e-9, atol=1e-9)
    if name.startswith("nile_llt"):
        # raw Nile flows (~1e3) behind P0 = 1e6 I: every smoother form loses ~10 digits in the first steps (include/kf.h) -- the numpy
        # restatement is held to the same figures (tests/test_oracle_golden.py) and the reference's own test to 1e-7 with covariances
        # [:5] left out (tests/test_kalman_filter.py:236-241).  Here: 1e-8 on the states, 1e-7 on covariances [5:], and the first five
        # within 1e-9 of the largest FILTERED entry (1e6)
        assert_allclose(ss[..., 0], g["smoothed_state"], rtol=1e-8, atol=1e-8)
        assert_allclose(sc[:, 5:], g["smoothed_state_cov"][:, 5:], rtol=1e-7, atol=1e-7)
        assert_allclose(sc[:, :5], g["smoothed_state_cov"][:, :5], rtol=0, atol=1e-9 * float(np.abs(fc).max()))
    else:
        assert_allclose(ss[..., 0], g["smoothed_state"], rtol=1e-9, atol=1e-9)
        assert_allclose(sc, g["smoothed_state_cov"], rtol=1e-9, atol=1e-9)
    # the same numbers from the smoother class when it is handed the filter's result (bitwise: one kernel, same tapes) ...
    six = flt(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"])
    s2, c2 = KalmanSmoother()(g["T"], g["R"], g["Q"], six[0], six[2], filter_result=six, data=g["y"], Z=g["Z"], H=g["H"])
    assert np.array_equal(s2, ss) and np.array_equal(c2, sc)
    # ... and the reference's recursion when it is not (states agree; arma11's covariances are the pinv case: 1e-2, test_gpu_parity)
    s3, c3 = KalmanSmoother()(g["T"], g["R"], g["Q"], six[0], six[2])
    assert_allclose(s3, ss, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ss).max())))


def test_filter_and_smooth_single_draw_reference_layout():
    """Per-draw (2-D) matrices in, the reference's shapes out: (n,m,1) states, (n,m,m) covariances."""
    from pymc_statespace_amd import SingleTimeseriesFilter

    g = load_golden("arma11.npz")
    b = 1
    outs = SingleTimeseriesFilter(mvn_const="exact").filter_and_smooth(
        g["y"][..., None], g["a0"][b][:, None], g["P0"][b], g["T"][b], g["Z"][b], g["R"][b], g["H"][b], g["Q"][b])
    n, m = g["y"].shape[0], g["T"].shape[-1]
    assert [o.shape for o in outs] == [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,), (n, m, 1), (n, m, m)]
    assert_allclose(outs[6][..., 0], g["smoothed_state"][b], rtol=1e-9, atol=1e-9)
    assert_allclose(outs[7], g["smoothed_state_cov"][b], rtol=1e-9, atol=1e-9)


def test_smooth_result_routes(eng):
    """auto: p == 1, static, zero c -> the predicted tapes; c != 0, p > 1 or time-varying -> the filtered tapes; route='from_filter'
    raises where it cannot run; route='rts' is the reference's recursion bit for bit."""
    import torch

    from pymc_statespace_amd.engine import KalmanError

    w = W.structural(B=9, n=60)
    res = eng.forward(*_args(w), mvn_const="exact")
    a_auto = eng.smooth_result(res, w["y"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    a_bf = eng.smooth_from_filter(w["y"], w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
    a_rts = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    assert torch.equal(a_auto[0], a_bf[0]) and torch.equal(a_auto[1], a_bf[1])
    r = eng.smooth_result(res, w["y"], w["T"], w["Z"], w["R"], w["H"], w["Q"], route="rts")
    assert torch.equal(r[0], a_rts[0]) and torch.equal(r[1], a_rts[1])
    # zero c given explicitly: still the predicted tapes; a non-zero c: the reference's recursion (which ignores c)
    c0 = np.zeros((9, w["m"]))
    z = eng.smooth_result(res, w["y"], w["T"], w["Z"], w["R"], w["H"], w["Q"], c0)
    assert torch.equal(z[0], a_bf[0])
    c1 = c0 + 0.3
    res_c = eng.forward(*_args(w), c=c1, mvn_const="exact")
    got = eng.smooth_result(res_c, w["y"], w["T"], w["Z"], w["R"], w["H"], w["Q"], c1)
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res_c.filtered_states.cpu().numpy(), res_c.filtered_covs.cpu().numpy())
    assert_allclose(got[0].cpu().numpy(), ra.reshape(got[0].shape), rtol=1e-9, atol=1e-9)
    assert_allclose(got[1].cpu().numpy(), rP, rtol=1e-9, atol=1e-9)
    with pytest.raises(KalmanError, match="from_filter"):
        eng.smooth_result(res_c, w["y"], w["T"], w["Z"], w["R"], w["H"], w["Q"], c1, route="from_filter")
    # p > 1: kf_smoother
    w5 = W.varmax(B=5, n=40, order=(2, 0))
    res5 = eng.forward(*_args(w5), mvn_const="exact")
    g5 = eng.smooth_result(res5, w5["y"], w5["T"], w5["Z"], w5["R"], w5["H"], w5["Q"])
    e5 = eng.smooth(w5["T"], w5["R"], w5["Q"], res5.filtered_states, res5.filtered_covs)
    assert torch.equal(g5[0], e5[0]) and torch.equal(g5[1], e5[1])


@pytest.mark.parametrize("m_case", ["m2", "m10"])
def test_fused_smoother_body_with_a_state_intercept(eng, m_case):
    """the fused node's body returned the predicted-tape recursion for c != 0 -- O(c) away from the
    reference's smoother, and only for 5 <= m <= 15.  Now the reference's numbers for every m; c == 0 takes the predicted tapes."""
    from pymc_statespace_amd import op

    w = W.arma11(B=1, n=80) if m_case == "m2" else W.structural(B=1, n=80)
    m = w["m"]
    one = {k: w[k][0] for k in KEYS}
    for cval in (0.0, 0.5):
        c = np.full((m, 1), cval)
        d = np.zeros((1, 1))
        outs = op.perform_filter(w["y"][..., None], one["a0"][:, None], one["P0"], one["T"], one["Z"], one["R"], one["H"], one["Q"], c, d,
                                 mvn_const="exact", engine=eng)
        fs, ps, fP, pP = outs[:4]
        sa, sP = op.perform_smoother_fused(w["y"][..., None], one["T"], one["Z"], one["R"], one["H"], one["Q"], c, d, fs, fP, ps, pP,
                                           engine=eng)
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], fs[None, ..., 0], fP[None])
        if cval == 0.0:
            da, dP = kn.smoother_from_filter(w["y"], w["T"], w["Z"], w["H"], ps[None, ..., 0], pP[None])
            assert_allclose(sa[..., 0], da[0], rtol=1e-10, atol=1e-10)
            assert_allclose(sP, dP[0], rtol=1e-10, atol=1e-10)
        else:
            # the reference's recursion on tapes that contain c: what kalman_smoother.py:87-104 returns (m2: ARMA with H = 0 -- pinv's
            # rank decision at rounding level, and c drives the second state to ~1e5: the RTS kernels' 1e-2 case, test_gpu_parity.py)
            assert_allclose(sa[..., 0], ra.reshape(1, -1, m)[0], rtol=1e-4 if m_case == "m2" else 1e-6, atol=1e-6)
            if m_case == "m10":
                assert_allclose(sP, rP[0], rtol=1e-8, atol=1e-8)
            da, _ = kn.smoother_from_filter(w["y"], w["T"], w["Z"], w["H"], ps[None, ..., 0], pP[None])
            assert np.abs(da[0] - sa[..., 0]).max() > 1e-3   # (the two recursions really differ: c matters)


# ------------------------------------------------------------------------------------------------------------------------------------
# the time-chunked forward pipeline for small batches (csrc/kf_forward_b4c.hip): kalman_filter.py:126-252 for p = 1, m <= 10, static
# matrices, no missing observation -- chunk-start covariances by doubling, B x nc chunk wavefronts, the mean pass
# ------------------------------------------------------------------------------------------------------------------------------------
def _chunk_dev(got, ref):
    import torch

    out = {}
    for k in ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
        g, r = getattr(got, k), getattr(ref, k)
        out[k] = float((g - r).abs().max() / (1e-300 + r.abs().max()))
    out["dlogp"] = float((got.logp - ref.logp).abs().max())
    assert torch.equal(got.info, ref.info)
    return out


CHUNK_CASES = {
    "headline_share_512x1000": lambda: (W.structural(B=512, n=1000), {}),
    "structural_64x300": lambda: (W.structural(B=64, n=300), {}),
    "structural_3x257": lambda: (W.structural(B=3, n=257), {}),          # (the shortest series the pipeline takes: two chunks)
    "dense_m6_c_d": lambda: (W.random_dense(100, 500, 1, 6, 2, seed=5),
                             dict(c=0.2 * np.random.default_rng(3).standard_normal((100, 6)),
                                  d=0.3 * np.random.default_rng(4).standard_normal((100, 1)))),
    "dense_m9_r1": lambda: (W.random_dense(37, 777, 1, 9, 1, seed=6), {}),
    "dense_m5_single": lambda: (W.random_dense(21, 400, 1, 5, 5, seed=7),
                                dict(d=0.3 * np.random.default_rng(8).standard_normal((21, 1)), filter_type="single")),
    "dense_m7_cholesky": lambda: (W.random_dense(21, 600, 1, 7, 3, seed=9), dict(filter_type="cholesky")),
    "dense_m8_univariate": lambda: (W.random_dense(21, 600, 1, 8, 2, seed=10), dict(filter_type="univariate")),
}


@pytest.mark.parametrize("case", sorted(CHUNK_CASES))
def test_chunked_forward_matches_the_recursion_at_every_step(eng, case, monkeypatch):
    """All six reference outputs of the chunked pipeline against the sequential kernel (KF_FLAG_EXACT_RECURSION) on the same inputs:
    |dlogp| < 1e-8 (BASELINE.json's tolerance; measured <= 3e-12), tapes within 1e-11 of their largest entry (measured <= 1e-13)."""
    monkeypatch.delenv("KF_FORWARD_CHUNKED", raising=False)
    w, kw = CHUNK_CASES[case]()
    ref = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True, **kw)
    got = eng.forward(*_args(w), mvn_const="exact", **kw)
    dev = _chunk_dev(got, ref)
    assert dev["dlogp"] < 1e-8, dev
    for k in ("ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
        assert dev[k] < 1e-11, (k, dev)
    # log-likelihood only (no tapes): the same numbers
    lp = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, **kw)
    assert float((lp.logp - got.logp).abs().max()) <= 1e-10 * max(1.0, float(got.logp.abs().max()))


def test_chunked_forward_against_the_numpy_oracle(eng):
    """The pipeline against the restated reference recursion itself (oracle/kalman_numpy.py), small enough for the Python loop."""
    w = W.random_dense(6, 300, 1, 4, 2, seed=21)
    eng.set_forward_kernel("b4")   # (m = 4 at this batch would go to the lane kernel)
    try:
        rng = np.random.default_rng(5)
        c, d = 0.1 * rng.standard_normal((6, 4)), 0.2 * rng.standard_normal((6, 1))
        got = eng.forward(*_args(w), c=c, d=d, mvn_const="exact")
        ref = kn.kalman_filter(w["y"], *[w[k] for k in KEYS], c=c, d=d, mvn_const="exact")
        assert_allclose(got.logp.cpu().numpy(), ref["logp"], rtol=1e-11, atol=1e-8)
        assert_allclose(got.ll_obs.cpu().numpy(), ref["ll_obs"], rtol=1e-9, atol=1e-9)
        assert_allclose(got.filtered_states.cpu().numpy(), ref["filtered_states"].reshape(6, 300, 4), rtol=1e-9, atol=1e-9)
        assert_allclose(got.predicted_states.cpu().numpy(), ref["predicted_states"].reshape(6, 301, 4), rtol=1e-9, atol=1e-9)
        assert_allclose(got.filtered_covs.cpu().numpy(), ref["filtered_covs"], rtol=1e-9, atol=1e-9)
        assert_allclose(got.predicted_covs.cpu().numpy(), ref["predicted_covs"], rtol=1e-9, atol=1e-9)
    finally:
        eng.set_forward_kernel("auto")


def test_chunked_forward_hands_what_it_cannot_vouch_for_to_the_sequential_kernel(eng, monkeypatch):
    """A missing observation makes the covariance path depend on the data (kalman_filter.py:195-213): the pipeline raises its device
    flag and the sequential kernel behind it redoes the batch -- bit for bit what KF_FORWARD_CHUNKED=0 gives; so does a series per
    draw with a NaN in ONE draw, and H = 0 with state noise the observation does not see (S0 = 0)."""
    import torch

    w = W.structural(B=128, n=600)
    y = w["y"].copy()
    y[[5, 300]] = np.nan
    got = eng.forward(*_args(w, y), mvn_const="exact")
    monkeypatch.setenv("KF_FORWARD_CHUNKED", "0")
    ref = eng.forward(*_args(w, y), mvn_const="exact")
    monkeypatch.delenv("KF_FORWARD_CHUNKED")
    for k in ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "info"):
        assert torch.equal(getattr(got, k), getattr(ref, k)), k
    yb = np.repeat(w["y"][None], 128, 0).copy()
    yb[77, 123] = np.nan
    got = eng.forward(yb, *[w[k] for k in KEYS], mvn_const="exact")
    monkeypatch.setenv("KF_FORWARD_CHUNKED", "0")
    ref = eng.forward(yb, *[w[k] for k in KEYS], mvn_const="exact")
    monkeypatch.delenv("KF_FORWARD_CHUNKED")
    for k in ("logp", "ll_obs", "filtered_covs", "predicted_states"):
        assert torch.equal(getattr(got, k), getattr(ref, k)), k
    # S0 = Z R Q R' Z' + H = 0: the one-step element does not exist
    w0 = W.random_dense(9, 300, 1, 5, 1, seed=3)
    w0["H"][:] = 0.0
    w0["R"][:, 0, :] = 0.0       # the observed state (Z = e_0 in random_dense? -- make it so) receives no noise
    w0["Z"][:] = 0.0
    w0["Z"][:, 0, 0] = 1.0
    got = eng.forward(*_args(w0), mvn_const="exact")
    ref = eng.forward(*_args(w0), mvn_const="exact", exact_recursion=True)
    assert_allclose(got.logp.cpu().numpy(), ref.logp.cpu().numpy(), rtol=1e-12)
    assert torch.equal(got.info, ref.info)


def test_chunked_forward_is_capturable_and_leaves_the_big_batches_alone(eng):
    """The pipeline is a memset and four launches on the caller's stream: captured into a hipGraph and replayed it gives the same bits;
    a batch that fills the chip (4096 draws) keeps the sequential kernel (bit for bit KF_FLAG_EXACT_RECURSION at T = 1000)."""
    import torch

    w = W.structural(B=200, n=400)
    dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    eager = eng.forward(*dv, mvn_const="exact")
    torch.cuda.synchronize()
    out = eng.forward(*dv, mvn_const="exact")          # buffers to capture into
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(*dv, mvn_const="exact", out=out)    # (workspace allocated outside the capture)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        out.logp.zero_()
        with torch.cuda.graph(gr, stream=s):
            eng.forward(*dv, mvn_const="exact", out=out)
        gr.replay()
    torch.cuda.synchronize()
    for k in ("logp", "ll_obs", "filtered_states", "predicted_covs"):
        assert torch.equal(getattr(out, k), getattr(eager, k)), k
    wb = W.structural(B=4096, n=300)
    a = eng.forward(*_args(wb), mvn_const="exact")
    b = eng.forward(*_args(wb), mvn_const="exact", exact_recursion=True)
    assert torch.equal(a.logp, b.logp) and torch.equal(a.predicted_covs, b.predicted_covs)


# ------------------------------------------------------------------------------------------------------------------------------------
# C5's sparse path under hipGraph capture: the verdict "T fits the CSR" needs a read-back, which a capturing
# stream cannot do -- round 5 then silently took the 3.6x slower MFMA path
# ------------------------------------------------------------------------------------------------------------------------------------
def test_sparse_workspace_path_is_taken_under_graph_capture(eng):
    """A captured call takes the CSR gather (bsp_step_kernel) when the last un-captured call on the same (T pointer, B, m) did, under
    a device-side guard: replayed on a dense T in the same buffer every draw comes back NaN with info 0x7fffff02 -- loud, never a
    wrong number; a first call that is captured has no verdict yet and keeps the MFMA path (results equal to rounding)."""
    import torch

    w = W.daily_seasonal(B=8, n=24, s=110)         # m = 111 (> 96: past the LDS-resident kernel): the workspace path, 2 non-zeros per row
    dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(mvn_const="exact", store_filtered=False, store_predicted=False)
    eng.release_workspace()
    out = eng.forward(*dv, **kw)                    # un-captured: read-back, verdict cached
    assert eng.last_forward_kernel() == "bsp_step_kernel"
    ref_logp = out.logp.clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(*dv, out=out, **kw)             # (workspace of this stream allocated outside the capture)
        s.synchronize()
        out.logp.zero_()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            eng.forward(*dv, out=out, **kw)
        assert eng.last_forward_kernel() == "bsp_step_kernel"     # the captured call took the sparse chain
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.logp, ref_logp) and not out.info.any()
        # the same graph on a T that no longer fits (dense rows everywhere): the guard speaks up
        T_keep = dv[3].clone()
        dv[3].copy_(torch.rand_like(dv[3]) * 0.01)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.isnan(out.logp).all() and (out.info == 0x7fffff02).all()
        dv[3].copy_(T_keep)
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.logp, ref_logp) and not out.info.any()


# ------------------------------------------------------------------------------------------------------------------------------------
# kf_smoother_from_filter on 16 x 16 fp64 MFMA tiles, one workgroup per draw (csrc/kf_smoother_bf_wg.hip, 16 <= m <= 79)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [16, 17, 31, 32, 40, 47, 48, 63, 64, 79])   # 2 / 2 / 2 / 3 / 3 / 3 / 4 / 4 / 5 / 5 wavefronts per draw
@pytest.mark.parametrize("rank", ["one", "full"])
def test_tile_smoother_from_filter_matches_restatement(eng, m, rank):
    """kalman_smoother.py:56-104 on the filter's own tapes for the SARIMA-sized models (k_posdef = 1: the RTS kernels' pinv case, seconds)
    and full-rank noise alike: the numpy restatement pinned to statsmodels at 1e-10 of the draw's largest filtered entry; missing
    observations (first, last, middle), an intercept d; and the reference's own recursion where that is well posed (full rank)."""
    r = 1 if rank == "one" else m
    w = W.random_dense(4, 37, 1, m, r, seed=300 + m + r)
    y = w["y"].copy()
    y[[0, 11, 36]] = np.nan
    d = 0.1 * np.random.default_rng(m).standard_normal((4, 1))
    res = eng.forward(y, *[w[k] for k in KEYS], d=d, mvn_const="exact")
    assert not res.info.cpu().numpy().any()
    sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d)
    da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy(), d=d)
    scale = max(1.0, float(np.abs(res.filtered_covs.cpu().numpy()).max()))
    assert_allclose(sa.cpu().numpy(), da, rtol=1e-10, atol=1e-10 * scale)
    assert_allclose(sP.cpu().numpy(), dP, rtol=1e-10, atol=1e-10 * scale)
    if rank == "full":
        ra, rP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        assert_allclose(sa.cpu().numpy(), ra.cpu().numpy(), rtol=1e-7, atol=1e-7 * scale)
        assert_allclose(sP.cpu().numpy(), rP.cpu().numpy(), rtol=1e-7, atol=1e-7 * scale)


def test_tile_smoother_from_filter_variants(eng, monkeypatch):
    """SingleTimeseriesFilter's sign of d, one series per draw, and the workspace path it replaced (KF_SMOOTHER_BF_WG=0) on the same tapes."""
    w = W.random_dense(5, 50, 1, 24, 2, seed=77)
    rng = np.random.default_rng(6)
    d = 0.2 * rng.standard_normal((5, 1))
    yb = w["y"][None] + 0.1 * rng.standard_normal((5,) + w["y"].shape)
    res = eng.forward(yb, *[w[k] for k in KEYS], d=d, filter_type="single", mvn_const="exact")
    sa, sP = eng.smooth_from_filter(yb, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d, filter_type="single")
    for b in (0, 4):
        da, dP = kn.smoother_from_filter(yb[b], w["T"][b:b + 1], w["Z"][b:b + 1], w["H"][b:b + 1], res.predicted_states[b:b + 1].cpu().numpy(),
                                         res.predicted_covs[b:b + 1].cpu().numpy(), d=d[b:b + 1], filter_type="single")
        assert_allclose(sa[b:b + 1].cpu().numpy(), da, rtol=1e-10, atol=1e-10)
        assert_allclose(sP[b:b + 1].cpu().numpy(), dP, rtol=1e-10, atol=1e-10)
    monkeypatch.setenv("KF_SMOOTHER_BF_WG", "0")
    wa, wP = eng.smooth_from_filter(yb, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d, filter_type="single")
    assert_allclose(sa.cpu().numpy(), wa.cpu().numpy(), rtol=1e-10, atol=1e-10)
    assert_allclose(sP.cpu().numpy(), wP.cpu().numpy(), rtol=1e-10, atol=1e-10)


# ---- gradients through the filter's tape outputs (kf_filter_backward_tapes; VERDICT round 5, missing #5) ------------------------------
# The reference differentiates a loss on ANY of the scan's outputs for free (README.md:6; filters/kalman_filter.py:166-212 returns the
# six of them); checker: torch.autograd through the torch restatement of the same scan (oracle/kalman_torch.py, its tapes pinned
# to the numpy oracle in tests/test_oracle_torch.py).  Tolerance: 1e-8 of the largest entry of each gradient.
TAPE_CASES = {
    "dense_p1m3": (lambda: W.random_dense(5, 20, 1, 3, 2, seed=21), "standard"),
    "dense_p2m4_fullH": (lambda: W.random_dense(4, 15, 2, 4, 2, seed=22, h_diag=False), "standard"),
    "dense_p3m6_univariate": (lambda: W.random_dense(3, 12, 3, 6, 3, seed=23), "univariate"),
    "structural_m10": (lambda: W.structural(B=3, n=25), "standard"),
    "arma11_cholesky": (lambda: W.arma11(B=6, n=30), "cholesky"),
}


def _tape_cotangents(rng, B, n, m, which):
    g = {}
    if "fa" in which:
        g["filtered_states"] = rng.standard_normal((B, n, m))
    if "pa" in which:
        g["predicted_states"] = rng.standard_normal((B, n + 1, m))
    if "fP" in which:
        x = rng.standard_normal((B, n, m, m))
        g["filtered_covs"] = 0.5 * (x + np.swapaxes(x, -1, -2))
    if "pP" in which:
        x = rng.standard_normal((B, n + 1, m, m))
        g["predicted_covs"] = 0.5 * (x + np.swapaxes(x, -1, -2))
    return g


def _backward_tapes(eng, w, ft, gt, y=None, c=None, d=None, g_logp=None, g_ll=None, tv=(), T=None):
    args = [w["y"] if y is None else y, w["a0"], w["P0"], w["T"] if T is None else T, w["Z"], w["R"], w["H"], w["Q"]]
    res = eng.forward(*args, c=c, d=d, filter_type=ft, mvn_const="exact", time_varying=tv)
    return eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                        g_logp=g_logp, g_ll_obs=g_ll, g_filtered_states=gt.get("filtered_states"),
                        g_predicted_states=gt.get("predicted_states"), g_filtered_covs=gt.get("filtered_covs"),
                        g_predicted_covs=gt.get("predicted_covs"), filter_type=ft, mvn_const="exact", want_cd=True,
                        time_varying=tv)


def _assert_grads(g, ref, names=("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d"), tol=1e-8, sym_P0=False):
    for k in names:
        got = g[k].cpu().numpy()
        if k == "P0" and sym_P0:   # (the Univariate restatement reads P z where the kernels' shared adjoint reads z' P: the two agree
            # on symmetric P0 -- all a covariance can be -- and differ in the antisymmetric part of the entrywise gradient)
            got = 0.5 * (got + np.swapaxes(got, -1, -2))
            ref = dict(ref, P0=0.5 * (ref["P0"] + np.swapaxes(ref["P0"], -1, -2)))
        scale = 1.0 + np.max(np.abs(ref[k]))
        assert np.max(np.abs(got - ref[k])) / scale < tol, (k, np.max(np.abs(got - ref[k])) / scale)


@pytest.mark.parametrize("case", sorted(TAPE_CASES))
@pytest.mark.parametrize("which", ["fa", "pa", "fP", "pP", "fa+pa+fP+pP"])
def test_backward_tapes_matches_autograd(eng, case, which):
    from oracle import kalman_torch as kt

    make, ft = TAPE_CASES[case]
    w = make()
    rng = np.random.default_rng(5)
    gt = _tape_cotangents(rng, w["B"], w["n"], w["m"], which.split("+"))
    g = _backward_tapes(eng, w, ft, gt)
    oft = "standard" if ft == "cholesky" else ft
    _, _, ref = kt.logp_and_grads(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], filter_type=oft,
                                  mvn_const="exact", g_tapes=gt)
    _assert_grads(g, ref, sym_P0=(ft == "univariate"))


def test_backward_tapes_with_logp_ll_c_d_and_missing(eng):
    """All six cotangents at once, c and d given, observations missing (whole steps and single entries)."""
    from oracle import kalman_torch as kt

    w = W.random_dense(4, 18, 2, 5, 2, seed=31)
    rng = np.random.default_rng(6)
    y = w["y"].copy()
    y[3] = np.nan
    y[7, 1] = np.nan
    y[-1] = np.nan
    c = 0.2 * rng.standard_normal((4, 5))
    d = 0.3 * rng.standard_normal((4, 2))
    g_logp = rng.standard_normal(4)
    g_ll = rng.standard_normal((4, 18))
    gt = _tape_cotangents(rng, 4, 18, 5, ("fa", "pa", "fP", "pP"))
    g = _backward_tapes(eng, w, "standard", gt, y=y, c=c, d=d, g_logp=g_logp, g_ll=g_ll)
    _, _, ref = kt.logp_and_grads(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], c=c, d=d, mvn_const="exact",
                                  g_logp=g_logp, g_ll_obs=g_ll, g_tapes=gt)
    _assert_grads(g, ref)


def test_backward_tapes_time_varying_T(eng):
    from oracle import kalman_torch as kt

    w = W.random_dense(3, 14, 1, 4, 2, seed=32)
    rng = np.random.default_rng(7)
    Tt = w["T"][:, None] * (1.0 + 0.05 * rng.standard_normal((3, 14, 1, 1)))
    gt = _tape_cotangents(rng, 3, 14, 4, ("fa", "pP"))
    g = _backward_tapes(eng, w, "standard", gt, tv=("T",), T=Tt)
    _, _, ref = kt.logp_and_grads(w["y"], w["a0"], w["P0"], Tt, w["Z"], w["R"], w["H"], w["Q"], mvn_const="exact",
                                  time_varying=("T",), g_tapes=gt)
    assert g["T"].shape == (3, 14, 4, 4)
    _assert_grads(g, ref)


def test_backward_tapes_only_the_symmetric_part_of_a_covariance_cotangent_acts(eng):
    """include/kf.h: the filter's covariances are symmetric by construction, so G and (G + G') / 2 give the same gradients --
    except g_P0 from predicted slot 0, which is P0 as given and takes its cotangent unchanged."""
    w = W.random_dense(3, 10, 1, 4, 2, seed=33)
    rng = np.random.default_rng(8)
    G = rng.standard_normal((3, 10, 4, 4))
    a = _backward_tapes(eng, w, "standard", {"filtered_covs": G})
    b = _backward_tapes(eng, w, "standard", {"filtered_covs": 0.5 * (G + np.swapaxes(G, -1, -2))})
    for k in a:
        assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=1e-13, atol=1e-13, err_msg=k)
    Gp = np.zeros((3, 11, 4, 4))
    Gp[:, 0] = rng.standard_normal((3, 4, 4))
    c = _backward_tapes(eng, w, "standard", {"predicted_covs": Gp})
    assert np.allclose(c["P0"].cpu().numpy(), Gp[:, 0], rtol=0, atol=1e-14)
    assert float(c["T"].abs().max()) == 0.0


def test_backward_tapes_without_tape_cotangents_is_kf_filter_backward(eng):
    w = W.structural(B=8, n=40)
    args = [w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"]]
    res = eng.forward(*args, mvn_const="exact")
    g_logp = np.linspace(0.5, 1.5, 8)
    a = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g_logp, mvn_const="exact")
    zero = np.zeros((8, 40, 10))
    b = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g_logp,
                     g_filtered_states=zero, mvn_const="exact")
    for k in a:   # (different kernels: the block adjoint against the LDS-resident one)
        ref = a[k].cpu().numpy()
        assert np.max(np.abs(b[k].cpu().numpy() - ref)) <= 1e-9 * (1.0 + np.max(np.abs(ref))), k


def test_backward_tapes_beyond_lds_is_refused(eng):
    w = W.random_dense(2, 6, 1, 60, 2, seed=34)
    args = [w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"]]
    res = eng.forward(*args, mvn_const="exact")
    with pytest.raises(Exception, match="LDS"):
        eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_filtered_states=np.ones((2, 6, 60)), mvn_const="exact")


@pytest.mark.parametrize("case", ["structural_m10", "dense_p3m6_univariate", "arma11_cholesky"])
def test_backward_tapes_tile_adjoint_against_the_lds_adjoint(eng, case):
    """The two kernels that carry tape cotangents (kf_backward_mfma.hip's TP instantiation: static, m <= 16, scalar observations; the
    LDS-resident adjoint: everything else, forced here with the handle's kernel selector) on the same call."""
    make, ft = TAPE_CASES[case]
    w = make()
    rng = np.random.default_rng(9)
    gt = _tape_cotangents(rng, w["B"], w["n"], w["m"], ("fa", "pa", "fP", "pP"))
    g_logp = rng.standard_normal(w["B"])
    a = _backward_tapes(eng, w, ft, gt, g_logp=g_logp)
    eng.set_forward_kernel("lds")
    try:
        b = _backward_tapes(eng, w, ft, gt, g_logp=g_logp)
    finally:
        eng.set_forward_kernel("auto")
    for k in a:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        if k == "P0" and w["p"] > 1:   # (p > 1 on the tile adjoint: the symmetric part, include/kf.h)
            y = 0.5 * (y + np.swapaxes(y, -1, -2))
            x = 0.5 * (x + np.swapaxes(x, -1, -2))
        if k == "H" and w["p"] > 1:
            y = y * np.eye(w["p"])
        assert np.max(np.abs(x - y)) <= 1e-9 * (1.0 + np.max(np.abs(y))), k


# ---- path switches per handle (kf_set_option; VERDICT round 5 #8: per-handle options instead of getenv) ---------------------------------
def test_set_option_wins_over_the_environment_and_is_per_handle(eng, monkeypatch):
    from pymc_statespace_amd import KalmanEngine

    other = KalmanEngine()
    tv = ("T",)
    assert eng.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_b4tv_kernel"
    eng.set_option("KF_FORWARD_B4TV", "0")
    try:
        assert eng.get_option("KF_FORWARD_B4TV") == "0"
        assert eng.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_lds_kernel"
        assert other.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_b4tv_kernel"   # another handle: untouched
        monkeypatch.setenv("KF_FORWARD_B4TV", "1")
        assert eng.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_lds_kernel"       # the handle wins
        eng.set_option("KF_FORWARD_B4TV", None)
        assert eng.get_option("KF_FORWARD_B4TV") == "1"                                               # back to the environment
        monkeypatch.setenv("KF_FORWARD_B4TV", "0")
        assert eng.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_lds_kernel"
        eng.set_option("KF_FORWARD_B4TV", "")                                                         # "" = as if unset
        assert eng.get_option("KF_FORWARD_B4TV") is None
        assert eng.forward_kernel_name(64, 50, 1, 6, 2, time_varying=tv) == "kf_fwd_b4tv_kernel"
    finally:
        eng.set_option("KF_FORWARD_B4TV", None)
    with pytest.raises(Exception, match="not a switch"):
        eng.set_option("KF_NO_SUCH_SWITCH", "1")
    with pytest.raises(Exception, match="longer"):
        eng.set_option("KF_STEADY_STATE", "x" * 40)


def test_set_option_selects_the_path_of_a_call(eng):
    """The chunked pipeline switched off on the handle: the call is the sequential kernel's, bit for bit what
    KF_FLAG_EXACT_RECURSION gives -- and on again: the pipeline's numbers (within 1e-11)."""
    import torch

    w = W.structural(B=128, n=600)
    a = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True)
    eng.set_option("KF_FORWARD_CHUNKED", "0")
    try:
        b = eng.forward(*_args(w), mvn_const="exact")
    finally:
        eng.set_option("KF_FORWARD_CHUNKED", None)
    c = eng.forward(*_args(w), mvn_const="exact")
    assert torch.equal(a.filtered_covs, b.filtered_covs) and torch.equal(a.logp, b.logp)
    assert not torch.equal(a.filtered_covs, c.filtered_covs)
    assert float((a.logp - c.logp).abs().max()) < 1e-8


@pytest.mark.parametrize("case", ["structural_m10", "dense_p3m6_univariate", "dense_p1m3"])
def test_backward_tapes_large_batch_variant_is_the_small_batch_kernel(eng, case):
    """Beyond three draws per SIMD the tile adjoint with tape cotangents runs its four-wavefront instantiation (cotangents loaded where
    they are used, csrc/kf_backward_mfma.hip W4): same numbers as the three-wavefront one, draw for draw."""
    make, ft = TAPE_CASES[case]
    w = make()
    rng = np.random.default_rng(12)
    B0, reps = w["B"], 3300 // w["B"] + 1
    gt = _tape_cotangents(rng, B0, w["n"], w["m"], ("fa", "pa", "fP", "pP"))
    g_logp = rng.standard_normal(B0)
    small = _backward_tapes(eng, w, ft, gt, g_logp=g_logp)
    big_w = dict(w, B=B0 * reps, **{k: np.tile(w[k], (reps,) + (1,) * (w[k].ndim - 1)) for k in KEYS})
    big_gt = {k: np.tile(v, (reps,) + (1,) * (v.ndim - 1)) for k, v in gt.items()}
    big = _backward_tapes(eng, big_w, ft, big_gt, g_logp=np.tile(g_logp, reps))
    assert big_w["B"] > 3072
    for k in small:
        a, b = small[k].cpu().numpy(), big[k].cpu().numpy()
        for rblk in (0, reps // 2, reps - 1):
            got = b[rblk * B0:(rblk + 1) * B0]
            assert np.max(np.abs(got - a)) <= 1e-12 * (1.0 + np.max(np.abs(a))), (k, rblk)


def test_backward_tapes_time_varying_Q_and_R(eng):
    """Time-varying state noise with tape cotangents: each step's R_t Q_t R_t' enters predicted slot t + 1 only, whose adjoint now
    carries that slot's cotangent (the LDS-resident adjoint's per-step g_R / g_Q)."""
    from oracle import kalman_torch as kt

    w = W.random_dense(3, 12, 1, 4, 2, seed=41)
    rng = np.random.default_rng(14)
    Qt = w["Q"][:, None] * (1.0 + 0.1 * rng.random((3, 12, 1, 1)))
    Rt = w["R"][:, None] * (1.0 + 0.1 * rng.standard_normal((3, 12, 1, 1)))
    gt = _tape_cotangents(rng, 3, 12, 4, ("pa", "fP", "pP"))
    args = [w["y"], w["a0"], w["P0"], w["T"], w["Z"], Rt, w["H"], Qt]
    tv = ("R", "Q")
    res = eng.forward(*args, mvn_const="exact", time_varying=tv)
    g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_predicted_states=gt["predicted_states"], g_filtered_covs=gt["filtered_covs"],
                     g_predicted_covs=gt["predicted_covs"], mvn_const="exact", want_cd=True, time_varying=tv)
    _, _, ref = kt.logp_and_grads(*args, mvn_const="exact", time_varying=tv, g_tapes=gt)
    assert g["Q"].shape == (3, 12, 2, 2) and g["R"].shape == (3, 12, 4, 2)
    _assert_grads(g, ref)


def test_set_option_forward_kernel_selector_and_its_removal(eng):
    name = lambda: eng.forward_kernel_name(64, 50, 1, 10, 2)  # noqa: E731
    assert name() == "kf_fwd_b4_kernel"
    eng.set_option("KF_FORWARD_KERNEL", "mfma")
    try:
        assert name() == "kf_fwd_mfma_kernel"
    finally:
        eng.set_option("KF_FORWARD_KERNEL", None)
    assert name() == "kf_fwd_b4_kernel"


# ---- the lane-per-draw forward kernel's chunked first 256 steps (csrc/kf_forward_lane.hip, KF_LANE_PRECHUNK) ---------------------------------
# BaseFilter.build_graph's scan (filters/kalman_filter.py:126-252) at p = 1, m <= 2: the first 256 steps as sixteen 16-step chunks on a
# draw slot's sixteen lanes -- chunk-start covariances from four squarings of the one-step Riccati element, means chained over the
# lanes.  Tolerances: the numpy oracle at 1e-9 (every output), the same kernel stepping from step 0 at 1e-11.
@pytest.mark.parametrize("case", ["arma11", "local_level_m1", "dense_m2_r2", "dense_m2_c_d_single"])
def test_lane_prechunk_against_the_oracle_and_the_sequential_steps(eng, case):
    rng = np.random.default_rng(3)
    kw = {}
    if case == "arma11":
        w = W.arma11(B=9, n=530)
    elif case == "local_level_m1":
        w = W.random_dense(6, 600, 1, 1, 1, seed=51)
    elif case == "dense_m2_r2":
        w = W.random_dense(7, 515, 1, 2, 2, seed=52)
    else:
        w = W.random_dense(5, 640, 1, 2, 1, seed=53)
        kw = dict(c=0.2 * rng.standard_normal((5, 2)), d=0.3 * rng.standard_normal((5, 1)), filter_type="single")
    assert eng.forward_kernel_name(w["B"], w["n"], 1, w["m"], w["r"]) == "kf_fwd_lane_kernel"
    got = eng.forward(*_args(w), mvn_const="exact", **kw)
    eng.set_option("KF_LANE_PRECHUNK", "0")
    try:
        seq = eng.forward(*_args(w), mvn_const="exact", **kw)
    finally:
        eng.set_option("KF_LANE_PRECHUNK", None)
    ref = kn.kalman_filter(*_args(w), mvn_const="exact", **kw)
    m = w["m"]
    for k in ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
        a, s_, r_ = getattr(got, k).cpu().numpy(), getattr(seq, k).cpu().numpy(), np.asarray(ref[k]).reshape(getattr(got, k).shape)
        scale = 1.0 + np.max(np.abs(r_))
        assert np.max(np.abs(a - r_)) <= 1e-9 * scale, (k, "oracle", np.max(np.abs(a - r_)) / scale)
        assert np.max(np.abs(a - s_)) <= 1e-11 * scale, (k, "sequential", np.max(np.abs(a - s_)) / scale)
    if case.startswith("dense_m2"):   # (the chunked path ran: slowly converging covariances differ in their last bits; an ARMA(1,1)
        # covariance has reached its fixed point after 16 steps and the doubling lands on the same bits)
        assert not np.array_equal(got.filtered_covs.cpu().numpy(), seq.filtered_covs.cpu().numpy())


def test_lane_prechunk_leaves_the_sequential_path_alone_when_in_doubt(eng):
    """A missing observation in the first 256 steps, a series shorter than 512 steps, KF_FLAG_EXACT_RECURSION: bit for bit the kernel
    stepping from step 0."""
    import torch

    w = W.arma11(B=13, n=600)
    y = w["y"].copy()
    y[77] = np.nan
    short = W.arma11(B=13, n=511)
    for args, kw in (((y,) + _args(w)[1:], {}), (_args(short), {}), (_args(w), dict(exact_recursion=True))):
        got = eng.forward(*args, mvn_const="exact", **kw)
        eng.set_option("KF_LANE_PRECHUNK", "0")
        try:
            seq = eng.forward(*args, mvn_const="exact", **kw)
        finally:
            eng.set_option("KF_LANE_PRECHUNK", None)
        for k in ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
            assert torch.equal(torch.nan_to_num(getattr(got, k), nan=-7.0), torch.nan_to_num(getattr(seq, k), nan=-7.0)), k
