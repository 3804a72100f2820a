"""Round-3 forward paths of the block kernel (p == 1, m <= 14), through the C-ABI, against the numpy / C oracle
(reference pymc_statespace/filters/kalman_filter.py:126-252, 255-351, 444-505):

* the PAIR kernel (csrc/kf_forward_b4p.hip: recursion wavefront + tape-writer wavefront per draw) -- what batches of
  <= 1024 draws with covariance tapes run on, and the tail of the split pipeline;
* the SPLIT pipeline (csrc/kf_forward_b4.hip SS = 2 + csrc/kf_forward_b4_steady.hip): recursion launches over compacted
  draw lists, converged draws handed to the constant-covariance fill and the time-parallel mean scan on other streams --
  the default for series of >= 1536 steps, requested below that with exact_recursion=False (KF_FLAG_STEADY_STATE);
* the convergence test that guards every steady-state phase: entrywise on two consecutive checks (a
  diffuse block must not hide a small block that still moves).

Tolerances: |dlogp| < 1e-8 * max(1, |logp| / 1e4); tapes rtol = atol = 1e-9 (as tests/test_gpu_parity.py).
"""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from oracle import c_oracle as co
from oracle import kalman_numpy as kn

pytestmark = pytest.mark.gpu
TAPES = ["filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs"]
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


@pytest.fixture(scope="module")
def eng():
    import torch

    from pymc_statespace_amd import KalmanEngine

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return KalmanEngine()


def _args(w, y=None):
    return (w["y"] if y is None else y,) + tuple(w[k] for k in KEYS)


def _check(res, ref, tag, draws=None):
    sl = slice(None) if draws is None else draws
    lp = res.logp.cpu().numpy()[sl]
    tol = 1e-8 * np.maximum(1.0, np.abs(ref["logp"]) / 1e4)
    assert np.all(np.abs(lp - ref["logp"]) <= tol), f"{tag}: max |dlogp| = {np.abs(lp - ref['logp']).max():.2e}"
    for k in TAPES:
        v = getattr(res, k)
        if v is not None:
            assert_allclose(v.cpu().numpy()[sl], ref[k], rtol=1e-9, atol=1e-9, err_msg=f"{tag} {k}")


# ----------------------------------------------------------------------------------------------
# pair kernel
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 2, 5, 6, 9, 10, 13, 14])
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_pair_kernel_matches_oracle(eng, m, ft):
    """Every block count (NB = 1 .. 4), missing rows, intercepts, every filter's constants; identical to the fused kernel."""
    B, n = 21, 173
    w = W.random_dense(B, n, 1, m, min(m, 3), seed=900 + m)
    y = w["y"].copy()
    y[[2, 40, 41, 172]] = np.nan
    rng = np.random.default_rng(m)
    c, d = 0.05 * rng.standard_normal((B, m)), 0.1 * rng.standard_normal((B, 1))
    ref = kn.kalman_filter(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact")
    r = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact")
    assert not r.info.cpu().numpy().any()
    _check(r, ref, f"pair m={m} {ft}")
    os.environ["KF_FORWARD_B4P"] = "0"
    try:
        f = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", exact_recursion=True)
    finally:
        del os.environ["KF_FORWARD_B4P"]
    for k in TAPES + ["logp"]:  # (same arithmetic: bit-identical at the north-star shape; rounding-level otherwise)
        assert_allclose(getattr(r, k).cpu().numpy(), getattr(f, k).cpu().numpy(), rtol=1e-12, atol=1e-13, err_msg=k)


def test_pair_kernel_partial_outputs_per_draw_data_and_flags(eng):
    w = W.structural(B=33, n=200)
    full = eng.forward(*_args(w), mvn_const="exact")
    for sel in ((False, True, False), (False, False, True), (True, True, False)):
        r = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=sel[0], store_filtered=sel[1], store_predicted=sel[2])
        assert np.array_equal(r.logp.cpu().numpy(), full.logp.cpu().numpy())
        for k in TAPES:
            v = getattr(r, k)
            if v is not None:
                assert np.array_equal(v.cpu().numpy(), getattr(full, k).cpu().numpy()), f"{sel} {k}"
    # y per draw, a draw whose H makes F negative, a NaN parameter
    yb = np.repeat(w["y"][None], 33, 0).copy()
    yb[3, 150] = np.nan
    H = w["H"].copy()
    H[4] = -50.0
    T = w["T"].copy()
    T[6, 0, 0] = np.nan
    r = eng.forward(yb, w["a0"], w["P0"], T, w["Z"], w["R"], H, w["Q"], mvn_const="exact")
    info = r.info.cpu().numpy()
    assert info[4] != 0 and info[6] != 0 and (info[np.r_[0:4, 5, 7:33]] == 0).all()  # (a NaN F is flagged like a negative one)
    lp = r.logp.cpu().numpy()
    assert np.isnan(lp[[4, 6]]).all()
    good = np.r_[0:4, 5, 7:33]
    ref = kn.kalman_filter(yb[good], *[w[k][good] for k in KEYS], mvn_const="exact")
    _check(r, ref, "pair per-draw y", draws=good)


# ----------------------------------------------------------------------------------------------
# split pipeline
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 2, 3, 5, 6, 7, 9, 10])
@pytest.mark.parametrize("n", [256, 300, 1000])
def test_split_pipeline_matches_oracle_and_exact_recursion(eng, m, n):
    """exact_recursion=False -> KF_FLAG_STEADY_STATE: chunks of the recursion, compaction, hand-over to the fill / scan
    kernels (and to the pair kernel once <= 1024 draws are left: at once for a batch this small)."""
    B = 37
    w = W.random_dense(B, n, 1, m, min(m, 3), seed=100 + m)
    ref = co.kalman_filter(*_args(w), filter_type="standard", mvn_const="exact")
    s = eng.forward(*_args(w), mvn_const="exact", exact_recursion=False)
    x = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True)
    assert not s.info.cpu().numpy().any()
    _check(s, ref, f"split m={m} n={n}")
    assert_allclose(s.logp.cpu().numpy(), x.logp.cpu().numpy(), rtol=1e-12)
    assert_allclose(s.predicted_covs.cpu().numpy(), x.predicted_covs.cpu().numpy(), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_split_pipeline_filters_intercepts_and_late_missing_rows(eng, ft):
    """A late NaN keeps every draw in the full recursion until it has passed; intercepts; the fused chunks with compaction
    (1500 draws: more than the pair kernel takes over) and then the pair kernel's tail."""
    B, n = 1500, 640
    w = W.structural(B=B, n=n)
    y = w["y"].copy()
    y[[5, 130, 131, 300]] = np.nan
    rng = np.random.default_rng(3)
    c, d = 0.05 * rng.standard_normal((B, w["m"])), 0.1 * rng.standard_normal((B, 1))
    k = 96
    ref = kn.kalman_filter(y, *[w[x][:k] for x in KEYS], c=c[:k], d=d[:k], filter_type=ft, mvn_const="exact")
    s = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", exact_recursion=False)
    x = eng.forward(*_args(w, y), c=c, d=d, filter_type=ft, mvn_const="exact", exact_recursion=True)
    assert not s.info.cpu().numpy().any()
    _check(s, ref, f"split {ft}", draws=slice(0, k))
    assert_allclose(s.logp.cpu().numpy(), x.logp.cpu().numpy(), rtol=1e-12)
    for kk in TAPES:
        assert_allclose(getattr(s, kk).cpu().numpy(), getattr(x, kk).cpu().numpy(), rtol=1e-10, atol=1e-11, err_msg=kk)
    # the phase did switch for some draw: the covariance tapes become exact constants
    fP = s.filtered_covs.cpu().numpy()
    assert np.any(np.all(fP[:, -1] == fP[:, -40], axis=(1, 2)))


def test_split_pipeline_partial_outputs_and_per_draw_data(eng):
    w = W.structural(B=1300, n=512)
    full = eng.forward(*_args(w), mvn_const="exact", exact_recursion=False)
    for sel in ((False, True, False), (False, False, True), (True, False, False), (False, False, False)):
        r = eng.forward(*_args(w), mvn_const="exact", store_ll_obs=sel[0], store_filtered=sel[1], store_predicted=sel[2],
                        exact_recursion=False)
        assert_allclose(r.logp.cpu().numpy(), full.logp.cpu().numpy(), rtol=1e-12)
        for k in TAPES:
            v = getattr(r, k)
            if v is not None:
                assert_allclose(v.cpu().numpy(), getattr(full, k).cpu().numpy(), rtol=1e-11, atol=1e-12, err_msg=f"{sel} {k}")
    B = 70
    w = W.structural(B=B, n=400)
    yb = np.repeat(w["y"][None], B, 0).copy()
    yb[3, 350] = np.nan          # that draw enters the steady phase only behind step 350
    yb[10] *= 1.5
    H = w["H"].copy()
    H[4] = -50.0                 # F < 0: flagged at the first step, logp NaN
    r = eng.forward(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"], mvn_const="exact", exact_recursion=False)
    x = eng.forward(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"], mvn_const="exact", exact_recursion=True)
    assert np.array_equal(r.info.cpu().numpy(), x.info.cpu().numpy())
    good = np.r_[0:4, 5:B]
    ref = kn.kalman_filter(yb[good], *[w[k][good] if k != "H" else H[good] for k in KEYS], mvn_const="exact")
    _check(r, ref, "split per-draw y", draws=good)


def test_split_pipeline_writes_inside_its_tapes_only(eng):
    """Guard words around every tape: the fill kernel's head / tail elements and line-aligned windows, the scan kernel's rows."""
    import torch

    from pymc_statespace_amd.engine import FilterResult

    for m, n in ((10, 450), (9, 300), (6, 700)):
        B = 11
        w = W.structural(B=B, n=n) if m == 10 else W.random_dense(B, n, 1, m, 3, seed=m)
        ref = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True)

        def guarded(*shape):
            flat = torch.full((int(np.prod(shape)) + 6,), float("nan"), device="cuda", dtype=torch.float64)
            return flat, flat[3:3 + int(np.prod(shape))].view(*shape)

        bufs = {k: guarded(*s) for k, s in dict(ll_obs=(B, n), filtered_states=(B, n, m), predicted_states=(B, n + 1, m),
                                                filtered_covs=(B, n, m, m), predicted_covs=(B, n + 1, m, m)).items()}
        out = FilterResult(logp=torch.empty(B, device="cuda", dtype=torch.float64),
                           info=torch.empty(B, device="cuda", dtype=torch.int32), **{k: v[1] for k, v in bufs.items()})
        eng.forward(*_args(w), mvn_const="exact", out=out, exact_recursion=False)
        for k, (flat, view) in bufs.items():
            assert_allclose(view.cpu().numpy(), getattr(ref, k).cpu().numpy(), rtol=1e-9, atol=1e-10, err_msg=k)
            gd = flat.cpu().numpy()
            assert np.isnan(gd[:3]).all() and np.isnan(gd[-3:]).all(), f"m={m} {k}: wrote outside the tape"


def test_long_series_take_the_split_pipeline_by_default(eng):
    """T >= 1536: the library's default for the block kernel (config C3 is T = 2000); C3-shaped, Cholesky filter."""
    w = W.structural(B=1100, n=1600)
    k = 48
    ref = co.kalman_filter(w["y"], *[w[x][:k] for x in KEYS], filter_type="cholesky", mvn_const="exact")
    d = eng.forward(*_args(w), filter_type="cholesky", mvn_const="exact")
    x = eng.forward(*_args(w), filter_type="cholesky", mvn_const="exact", exact_recursion=True)
    _check(d, ref, "default T=1600", draws=slice(0, k))
    fd = d.filtered_covs.cpu().numpy()
    assert np.any(np.all(fd[:, -1] == fd[:, -200], axis=(1, 2)))  # the phase did switch: exact constants on the tape
    assert_allclose(d.logp.cpu().numpy(), x.logp.cpu().numpy(), rtol=1e-12)
    # T = 1000 (the north-star length): the default IS the recursion at every step
    w = W.structural(B=1100, n=1000)
    d = eng.forward(*_args(w), mvn_const="exact")
    x = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True)
    for kk in TAPES + ["logp"]:
        assert np.array_equal(getattr(d, kk).cpu().numpy(), getattr(x, kk).cpu().numpy()), kk


# ----------------------------------------------------------------------------------------------
# the convergence test
# ----------------------------------------------------------------------------------------------
def _diffuse_beside_slow(B, n, rho, seed=0):
    """Local level with a diffuse start (P0 = 1e7) and large variance beside an AR(1) component with a small variance whose
    covariance converges slowly (rho close to 1): under a test scaled by the LARGEST entry alone the small block is
    still moving at ~1e-6 relative when the large one has stopped."""
    rng = np.random.default_rng(seed)
    m, r = 2, 2
    T = np.zeros((B, m, m)); T[:, 0, 0] = 1.0; T[:, 1, 1] = rho
    Z = np.ones((B, 1, m))
    R = np.repeat(np.eye(m)[None], B, 0)
    Q = np.zeros((B, r, r)); Q[:, 0, 0] = 1e4 * (1 + rng.random(B)); Q[:, 1, 1] = 1e-4 * (1 + rng.random(B))
    H = np.full((B, 1, 1), 1e4)
    a0 = np.zeros((B, m))
    P0 = np.zeros((B, m, m)); P0[:, 0, 0] = 1e7; P0[:, 1, 1] = 1e-4 / (1 - rho ** 2) * 50
    y = np.cumsum(100 * rng.standard_normal(n))
    return dict(y=y[:, None], a0=a0, P0=P0, T=T, Z=Z, R=R, H=H, Q=Q, B=B, n=n, m=m)


@pytest.mark.parametrize("kernel", ["lane", "b4"])
@pytest.mark.parametrize("rho", [0.99, 0.999])
def test_steady_state_test_is_entrywise(eng, kernel, rho):
    """A diffuse 1e7 state beside a slowly converging small component: the steady-state phase must not freeze the small
    block early -- its covariance and the log-likelihood stay within the parity tolerances of the step-by-step recursion."""
    w = _diffuse_beside_slow(40, 1200, rho)
    eng.set_forward_kernel(kernel)
    try:
        s = eng.forward(*_args(w), mvn_const="exact", exact_recursion=False)
        x = eng.forward(*_args(w), mvn_const="exact", exact_recursion=True)
    finally:
        eng.set_forward_kernel("auto")
    ps, px = s.predicted_covs.cpu().numpy(), x.predicted_covs.cpu().numpy()
    # entrywise RELATIVE agreement of the small block (an absolute 1e-9 would hide it under the 1e4-sized entries): the frozen
    # covariance is within tol / (1 - rho) = 1e-12 / (1 - rho) of the fixpoint, entry by entry
    tol = 3e-12 / (1.0 - rho)
    assert_allclose(ps[:, :, 1, 1], px[:, :, 1, 1], rtol=tol)
    assert_allclose(ps, px, rtol=tol, atol=1e-12)
    assert_allclose(s.logp.cpu().numpy(), x.logp.cpu().numpy(), rtol=1e-11)
    assert_allclose(s.filtered_states.cpu().numpy(), x.filtered_states.cpu().numpy(), rtol=1e-8, atol=1e-8)


# ----------------------------------------------------------------------------------------------
# steady-state adjoint (csrc/kf_backward_steady.hip) behind kf_filter_value_and_grad
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ft", ["standard", "univariate", "cholesky"])
@pytest.mark.parametrize("shape", ["varmax_m15_p5", "varmax_m10_p5", "varmax_m9_p3", "missing"])
def test_value_and_grad_with_the_steady_state_adjoint(eng, monkeypatch, ft, shape):
    """VARMAX shapes (the multi-observation block kernel's steady-state phase): logp and its nine gradients from
    kf_filter_value_and_grad -- the steps behind each draw's convergence point through the steady-state adjoint (a vector
    recursion + one Riccati adjoint per draw), the steps before it through the exact adjoint -- against torch autograd of
    the recursion at every step (the reference's gradient IS autodiff of the scan, core/statespace.py:174) and against
    the exact adjoint at every step (KF_VG_STEADY=0).  Gradients relative to the largest entry, 1e-10."""
    from oracle import kalman_torch as kt

    w = {"varmax_m15_p5": lambda: W.varmax(B=20, n=300), "varmax_m10_p5": lambda: W.varmax(B=20, n=300, order=(2, 0)),
         "varmax_m9_p3": lambda: W.varmax(B=20, n=250, k=3, order=(2, 1)), "missing": lambda: W.varmax(B=12, n=260)}[shape]()
    y = w["y"].copy()
    if shape == "missing":
        y[[3, 100]] = np.nan
        y[150, 2] = np.nan
    B, m, p = w["B"], w["m"], w["p"]
    rng = np.random.default_rng(5)
    c, d = 0.05 * rng.standard_normal((B, m)), 0.1 * rng.standard_normal((B, p))
    gl = 1.0 + 0.3 * rng.standard_normal(B)
    args = (y,) + tuple(w[k] for k in KEYS)
    kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact", h_diagonal=True)
    lp, info, g = eng.value_and_grad(*args, **kw)
    monkeypatch.setenv("KF_VG_STEADY", "0")
    lpx, _, gx = eng.value_and_grad(*args, **kw)
    monkeypatch.delenv("KF_VG_STEADY")
    assert not info.cpu().numpy().any()
    k = 5
    rlp, _, rg = kt.logp_and_grads(y, *[w[x][:k] for x in KEYS], c=c[:k], d=d[:k], filter_type=ft, mvn_const="exact", g_logp=gl[:k])
    assert np.max(np.abs(lp[:k].cpu().numpy() - rlp)) < 1e-8
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    for nm in ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d"):
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        assert np.max(np.abs(a_ - x_)) / (1.0 + np.max(np.abs(x_))) < 1e-10, f"{nm} vs the exact adjoint"
        ref = rg[nm] * np.eye(p) if nm == "H" else rg[nm]           # (a diagonal H: its diagonal only, as every scalar-observation adjoint)
        a_ = a_[:k].reshape(ref.shape)
        if nm in ("P0", "Q"):                                         # symmetric inputs: the symmetric parts agree
            a_, ref = a_ + np.swapaxes(a_, -1, -2), ref + np.swapaxes(ref, -1, -2)
        assert np.max(np.abs(a_ - ref)) / (1.0 + np.max(np.abs(ref))) < 1e-9, f"{nm} vs autograd"


def test_split_pipeline_and_steady_adjoint_replay_from_a_hip_graph(eng):
    """The split pipeline forks to the handle's own streams and joins back with events; the steady-state value-and-gradient
    is five launches on the caller's stream: both must capture into a hipGraph (stream and graph semantics)
    and replay on new data."""
    import torch

    w = W.structural(B=1300, n=512)
    dev = [eng.to_device(x) for x in _args(w)]
    wv = W.varmax(B=64, n=200)
    dvv = [eng.to_device(x) for x in _args(wv)]

    def step():
        res = eng.forward(*dev, mvn_const="exact", exact_recursion=False)
        small = eng.forward(*[d if i == 0 else d[:40] for i, d in enumerate(dev)], mvn_const="exact")   # pair kernel
        vg = eng.value_and_grad(*dvv, mvn_const="exact", h_diagonal=True)
        return res, small, vg

    step()  # lazy initialisation (streams, events, workspaces) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res_c, small_c, vg_c = step()
    rng = np.random.default_rng(1)
    dev[0].copy_(torch.as_tensor(w["y"] + 0.3 * rng.standard_normal(w["y"].shape), device=dev[0].device))
    dvv[0].copy_(torch.as_tensor(wv["y"] + 0.3 * rng.standard_normal(wv["y"].shape), device=dev[0].device))
    graph.replay()
    torch.cuda.synchronize()
    res_e, small_e, vg_e = step()
    torch.cuda.synchronize()
    for k in TAPES + ["logp"]:
        assert torch.equal(getattr(res_c, k), getattr(res_e, k)), k
        assert torch.equal(getattr(small_c, k), getattr(small_e, k)), k
    assert torch.equal(vg_c[0], vg_e[0])
    for k in vg_e[2]:
        assert torch.equal(vg_c[2][k], vg_e[2][k]), k


@pytest.mark.parametrize("m", [5, 8, 10, 12, 15])
def test_block_smoother_reuses_the_gain_of_a_converged_filter(eng, m):
    """kf_smoother_b4.hip: where the filtered covariance on the tape is bit for bit the next step's (a steady-state forward's constant
    tapes) the step reuses the last X_e = P_hat^-1 T P_f; a missing stretch in the middle of the series moves the covariance again
    and the full path takes over.  Against the pinv oracle (kalman_smoother.py:56-104) and against every step in full
    (KF_FLAG_EXACT_RECURSION), on the tapes of both forward modes."""
    import torch

    w = W.random_dense(7, 600, 1, m, m, seed=260 + m)
    y = w["y"].copy()
    y[100:109] = np.nan
    y[200] = np.nan
    for exact_fwd in (True, False):
        res = eng.forward(y, *[w[k] for k in KEYS], exact_recursion=exact_fwd, store_ll_obs=False, store_predicted=False)
        fa, fP = res.filtered_states, res.filtered_covs
        # the tape does converge: otherwise this test exercises nothing
        d = (fP[:, 590] - fP[:, 589]).abs().flatten(1).max(1).values / fP[:, 590].abs().flatten(1).max(1).values
        assert int((d <= 1e-15).sum()) >= 4   # (a slow draw or two among the seven keep moving: the full path beside the reuse)
        if not exact_fwd and m <= 10:
            assert int((d == 0.0).sum()) >= 4  # the steady-state forward's tapes (p = 1: m <= 10) are constant to the bit: the reuse runs
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], fa.cpu().numpy(), fP.cpu().numpy())
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], fa, fP)
        ea, eP = eng.smooth(w["T"], w["R"], w["Q"], fa, fP, exact_recursion=True)
        sc_a, sc_P = 1.0 + np.abs(ra).max(), 1.0 + np.abs(rP).max()
        assert np.abs(sa.cpu().numpy() - ra).max() / sc_a < 1e-11 and np.abs(sP.cpu().numpy() - rP).max() / sc_P < 1e-11
        assert float((sa - ea).abs().max()) / sc_a < 1e-12 and float((sP - eP).abs().max()) / sc_P < 1e-12
        assert torch.equal(sa[:, -1], fa[:, -1]) and torch.equal(sP[:, -1], fP[:, -1])


def test_block_smoother_reuse_test_is_entrywise(eng):
    """A diffuse, large-variance level beside five slowly converging small AR(1) components (the forward kernels' ADVICE case, one
    size up): the smoother's gain reuse must not treat the tape as converged while the small block still moves -- every entry of the
    smoothed covariance stays within 1e-9 of its own scale sqrt(P_ii P_jj) of the every-step result, the small block's diagonal
    within 1e-9 relative."""
    import torch

    B, n, m = 12, 1500, 6
    rng = np.random.default_rng(9)
    rho = 0.999 - 0.002 * np.arange(m - 1)
    T = np.zeros((B, m, m)); T[:, 0, 0] = 1.0
    for i in range(1, m):
        T[:, i, i] = rho[i - 1]
    Z = np.ones((B, 1, m))
    R = np.repeat(np.eye(m)[None], B, 0)
    Q = np.zeros((B, m, m)); Q[:, 0, 0] = 1e4 * (1 + rng.random(B))
    P0 = np.zeros((B, m, m)); P0[:, 0, 0] = 1e7
    for i in range(1, m):
        Q[:, i, i] = 1e-4 * (1 + rng.random(B))
        P0[:, i, i] = 1e-4 / (1 - rho[i - 1] ** 2) * 50
    H = np.full((B, 1, 1), 1e4)
    y = np.cumsum(100 * rng.standard_normal(n))[:, None]
    res = eng.forward(y, np.zeros((B, m)), P0, T, Z, R, H, Q, mvn_const="exact", exact_recursion=True, store_ll_obs=False,
                      store_predicted=False)
    fa, fP = res.filtered_states, res.filtered_covs
    sa, sP = eng.smooth(T, R, Q, fa, fP)
    ea, eP = eng.smooth(T, R, Q, fa, fP, exact_recursion=True)
    d = torch.diagonal(eP, dim1=-2, dim2=-1)
    scale = torch.sqrt(d.unsqueeze(-1) * d.unsqueeze(-2))
    assert float(((sP - eP).abs() / scale).max()) < 1e-9
    ds = torch.diagonal(sP, dim1=-2, dim2=-1)
    assert float(((ds - d).abs() / d)[..., 1:].max()) < 1e-9
    assert float((sa - ea).abs().max()) < 1e-7 * float(ea.abs().max())
