"""Round-4 paths through the C-ABI against the oracle:

* kf_filter_value_and_grad for the scalar-observation block kernel (p == 1, m <= 10) with the STEADY-STATE ADJOINT behind each
  draw's convergence point (csrc/kf_forward_b4.hip "VGS" instantiation -> kf_bwd_steady_kernel -> kf_dare_bwd_kernel ->
  kf_bwd_b4_kernel with per-draw step counts).  The reference's gradient is autodiff of the scan at every step
  (core/statespace.py:174; kalman_filter.py:215-223, 255-284): asserted at 1e-10 of the largest entry against the exact adjoint
  at every step (KF_VG_STEADY=0) and at 1e-9 against torch autograd of the restated recursion, including late missing
  observations (the phase may start only behind the last of them) and draws whose covariance never converges.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W

pytestmark = pytest.mark.gpu
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
GRADS = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")


@pytest.fixture(scope="module")
def eng():
    import torch

    from pymc_statespace_amd import KalmanEngine

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return KalmanEngine()


def _never_converging(w, idx):
    """Give the draws `idx` a unit-root level with a large Q / tiny H ratio drift: a random-walk-plus-drift state whose
    predicted covariance keeps growing (P0 far from the fixpoint and T with an eigenvalue of exactly one, a second one above
    0.9999: the Riccati iteration moves by ~1e-6 of itself per step for thousands of steps)."""
    w = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    for b in idx:
        w["T"][b, 0, 0] = 1.0
        w["T"][b, 0, 1] = 1.0
        w["T"][b, 1, 1] = 1.0
        w["P0"][b] = 1e3 * np.eye(w["m"])
        w["Q"][b] *= 1e-8
    return w


def _compare(eng, monkeypatch, w, y, ft, c=None, d=None, k=4, need_steady=True, tol_autograd=1e-9):
    from oracle import kalman_torch as kt

    B, m, p = w["B"], w["m"], w["p"]
    rng = np.random.default_rng(11)
    gl = 1.0 + 0.3 * rng.standard_normal(B)
    args = (y,) + tuple(w[x] for x in KEYS)
    kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
    # exact_recursion=False = KF_FLAG_STEADY_STATE: the steady-state adjoint whatever the batch size (AUTO takes it from two rounds
    # of the chip's wavefront slots on, include/kf.h)
    lp, info, g = eng.value_and_grad(*args, exact_recursion=False, **kw)
    monkeypatch.setenv("KF_VG_STEADY", "0")
    lpx, infox, gx = eng.value_and_grad(*args, exact_recursion=False, **kw)
    monkeypatch.delenv("KF_VG_STEADY")
    lpe, _, ge = eng.value_and_grad(*args, exact_recursion=True, **kw)
    assert not info.cpu().numpy().any() and not infox.cpu().numpy().any()
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    assert_allclose(lpe.cpu().numpy(), lpx.cpu().numpy(), rtol=0, atol=0)   # the flag and the switch select the same path
    differs = False
    for nm in GRADS:
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        assert np.all(np.isfinite(a_)), nm
        differs = differs or not np.array_equal(a_, x_)
        scale = 1.0 + np.max(np.abs(x_.reshape(B, -1)), axis=1)
        err = np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / scale
        assert err.max() < 1e-10, f"{nm} vs the exact adjoint: {err.max():.2e} (draw {err.argmax()})"
        assert np.array_equal(ge[nm].cpu().numpy(), x_), nm
    if need_steady:
        assert differs, "the steady-state adjoint did not run (bitwise equal to the exact adjoint)"
    idx = list(range(k)) if np.isscalar(k) else list(k)   # the draws autograd is run on
    cs = None if c is None else c[idx]
    ds = None if d is None else d[idx]
    rlp, _, rg = kt.logp_and_grads(y, *[w[x][idx] for x in KEYS], c=cs, d=ds, filter_type=ft, mvn_const="exact", g_logp=gl[idx])
    assert np.max(np.abs(lp.cpu().numpy()[idx] - rlp) / np.maximum(1.0, np.abs(rlp) / 1e4)) < 1e-8
    for nm in GRADS:
        a_ = g[nm].cpu().numpy()[idx].reshape(rg[nm].shape)
        ref = rg[nm]
        if nm in ("P0", "Q"):   # symmetric inputs: the symmetric parts agree
            a_, ref = a_ + np.swapaxes(a_, -1, -2), ref + np.swapaxes(ref, -1, -2)
        assert np.max(np.abs(a_ - ref)) / (1.0 + np.max(np.abs(ref))) < tol_autograd, f"{nm} vs autograd"


@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
def test_value_and_grad_block_kernel_with_the_steady_state_adjoint(eng, monkeypatch, ft):
    w = W.structural(B=24, n=400)
    rng = np.random.default_rng(3)
    c, d = 0.05 * rng.standard_normal((w["B"], w["m"])), 0.1 * rng.standard_normal((w["B"], 1))
    _compare(eng, monkeypatch, w, w["y"], ft, c=c, d=d)


@pytest.mark.parametrize("m", [1, 2, 3, 5, 6, 7, 9, 10])
def test_value_and_grad_steady_state_adjoint_every_block_count(eng, monkeypatch, m):
    """m + 2 <= 4 / 8 / 12: one, two and three 4 x 4 blocks per tile row; dense random stable models."""
    w = W.random_dense(B=16, n=300, p=1, m=m, r=max(1, m // 2), seed=40 + m)
    if m <= 4:
        eng.set_forward_kernel("b4")   # (AUTO hands m <= 4 to the lane kernel: force the block kernels)
    try:
        _compare(eng, monkeypatch, w, w["y"], "standard")
    finally:
        eng.set_forward_kernel("auto")


def test_value_and_grad_steady_state_adjoint_late_nans_and_slow_draws(eng, monkeypatch):
    """Missing observations late in the series (the phase starts behind the last one, or never: fewer than 16 steps left) and
    draws whose covariance never converges (the whole sweep is the exact adjoint's; the steady-state adjoint and the Riccati
    adjoint see zero steps and must leave exact zeros, no NaN from 0 * inf)."""
    w = _never_converging(W.structural(B=20, n=900), [1, 7, 19])
    y = w["y"].copy()
    y[[5, 130, 131, 250]] = np.nan
    # (autograd on well-conditioned draws and ONE diffuse draw's log-likelihood path; the diffuse draws' P0 = 1e3 I gradients carry
    # the recursion's own cancellation -- they are held to the exact adjoint, 1e-10, not to a differently ordered recursion)
    # 900 steps: the gradients of the INITIAL moments pass through the whole series -- the hand-written adjoint (rank-one update
    # form) and autograd of the Joseph-form restatement drift apart by ~1e-8 of the largest entry there (the exact adjoint does
    # too: it is held to the steady-state adjoint at 1e-10 above); the 1e-9 of the 300 - 400-step tests holds for every other input
    _compare(eng, monkeypatch, w, y, "standard", k=(0, 2, 3), tol_autograd=1e-7)
    y2 = w["y"].copy()
    y2[[10, 891]] = np.nan          # nothing can converge behind step 891 (16 steps must remain): every draw takes the exact path
    _compare(eng, monkeypatch, w, y2, "standard", k=2, need_steady=False, tol_autograd=1e-7)


def test_value_and_grad_steady_state_adjoint_large_batch_and_default_cotangent(eng, monkeypatch):
    """More draws than wavefront slots of the chip (queued workgroups), g_logp = NULL (ones), per-draw series."""
    w = W.random_dense(B=9000, n=200, p=1, m=9, r=3, seed=77)   # (stable dense models: every draw converges within ~100 steps)
    args = (w["y"],) + tuple(w[x] for x in KEYS)
    lp, info, g = eng.value_and_grad(*args, mvn_const="exact")        # AUTO: >= two rounds of wavefront slots -> steady-state adjoint
    lpx, _, gx = eng.value_and_grad(*args, mvn_const="exact", exact_recursion=True)
    assert any(not np.array_equal(g[nm].cpu().numpy(), gx[nm].cpu().numpy()) for nm in GRADS), "AUTO did not take the steady-state adjoint"
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    for nm in GRADS:
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        scale = 1.0 + np.max(np.abs(x_.reshape(9000, -1)), axis=1)
        assert (np.max(np.abs(a_ - x_).reshape(9000, -1), axis=1) / scale).max() < 1e-10, nm
    wy = W.structural(B=12, n=260)
    rng = np.random.default_rng(2)
    ys = wy["y"][None] + 0.2 * rng.standard_normal((12,) + wy["y"].shape)
    _compare(eng, monkeypatch, wy, ys[0], "standard")     # shared series
    lp, _, g = eng.value_and_grad(ys, *[wy[x] for x in KEYS], mvn_const="exact", exact_recursion=False)
    for b in (0, 5, 11):
        lpb, _, gb = eng.value_and_grad(ys[b], *[wy[x][b:b + 1] for x in KEYS], mvn_const="exact", exact_recursion=False)
        assert_allclose(lp[b].item(), lpb[0].item(), rtol=1e-13)
        for nm in GRADS:
            assert_allclose(g[nm][b].cpu().numpy(), gb[nm][0].cpu().numpy(), rtol=1e-9, atol=1e-11, err_msg=nm)


@pytest.mark.parametrize("m,n", [(1, 210), (3, 277), (6, 400), (10, 333), (10, 1000)])
def test_time_parallel_steady_state_adjoint_matches_the_sequential_one(eng, monkeypatch, m, n):
    """kf_bwd_steady_scan_kernel (64 steps per wavefront pass, prefix scan of the affine lambda recursion) against
    kf_bwd_steady_kernel (step by step; KF_VG_STEADY_SCAN=0): the same sums in a different order -- rounding only.  Series lengths
    that leave full, partial and single-step last blocks; a missing observation early on (the phase starts behind it)."""
    w = W.structural(B=40, n=n) if m == 10 else W.random_dense(B=40, n=n, p=1, m=m, r=max(1, m // 2), seed=90 + m)
    y = w["y"].copy()
    y[7] = np.nan
    rng = np.random.default_rng(m)
    gl = 1.0 + 0.3 * rng.standard_normal(w["B"])
    c, d = 0.05 * rng.standard_normal((w["B"], w["m"])), 0.1 * rng.standard_normal((w["B"], 1))
    args = (y,) + tuple(w[x] for x in KEYS)
    if m <= 4:
        eng.set_forward_kernel("b4")
    try:
        for ft in ("standard", "single"):
            kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact", exact_recursion=False)
            lp, info, g = eng.value_and_grad(*args, **kw)
            monkeypatch.setenv("KF_VG_STEADY_SCAN", "0")
            lps, infos, gs = eng.value_and_grad(*args, **kw)
            monkeypatch.delenv("KF_VG_STEADY_SCAN")
            assert not info.cpu().numpy().any()
            assert np.array_equal(info.cpu().numpy(), infos.cpu().numpy())
            assert_allclose(lp.cpu().numpy(), lps.cpu().numpy(), rtol=1e-13)   # (the forward means of those steps come from the scan too)
            differs = False
            for nm in GRADS:
                a_, x_ = g[nm].cpu().numpy(), gs[nm].cpu().numpy()
                differs = differs or not np.array_equal(a_, x_)
                scale = 1.0 + np.max(np.abs(x_.reshape(w["B"], -1)), axis=1)
                err = np.max(np.abs(a_ - x_).reshape(w["B"], -1), axis=1) / scale
                assert err.max() < 1e-12, f"{ft} {nm}: {err.max():.2e} (draw {err.argmax()})"
            assert differs, "the switch selected the same kernel twice"
    finally:
        eng.set_forward_kernel("auto")


def test_forward_placement_hint_never_changes_a_result(eng, monkeypatch):
    """The forward launch of the steady-state value-and-gradient path is ordered by the convergence points the previous call left
    in the workspace (kf_vg_order_kernel).  That decides where a draw runs, nothing else: first call (no header), repeated call
    (exact hint), a call after DIFFERENT parameters (stale hint), a workspace overwritten with garbage that keeps a valid header,
    and the switch off -- bitwise the same outputs."""
    w = _never_converging(W.structural(B=300, n=500), [3, 100, 299])
    w2 = W.structural(B=300, n=500, seed=4321)
    args = lambda x: (x["y"],) + tuple(x[k] for k in KEYS)   # noqa: E731
    kw = dict(mvn_const="exact", exact_recursion=False)

    def run(x):
        lp, info, g = eng.value_and_grad(*args(x), **kw)
        return [lp.cpu().numpy(), info.cpu().numpy()] + [g[nm].cpu().numpy() for nm in GRADS]

    eng.release_workspace()
    first = run(w)                      # fresh workspace: no header -> identity order
    again = run(w)                      # the hint is this call's own convergence points
    run(w2)
    stale = run(w)                      # the hint belongs to other parameters
    ws = eng._ws[next(iter(eng._ws))]   # the cached workspace of this entry point / stream
    import torch
    rnd = torch.randint(0, 255, (min(ws.numel(), 1 << 16),), dtype=torch.uint8, device=ws.device)
    ws[:1 << 16] = rnd                  # garbage over ones | ts | header | order: the header check fails -> identity
    garbage = run(w)
    run(w)
    ws32 = ws[: (ws.numel() // 4) * 4].view(torch.int32)
    pos = (ws32[: 1 << 14] == 0x6B665F74).nonzero()   # the header's magic word: [ones | ts | header | ...] at the head of the workspace
    assert pos.numel() >= 1
    h0 = int(pos[0])
    ws32[h0 - 300: h0] = torch.randint(-(2 ** 31), 2 ** 31 - 1, (300,), dtype=torch.int64, device=ws.device).to(torch.int32)
    # valid header, random "convergence points" (negative, huge) in the slots ahead of it: still a permutation
    scrambled = run(w)
    monkeypatch.setenv("KF_VG_SCHED_HINT", "0")
    off = run(w)
    monkeypatch.delenv("KF_VG_SCHED_HINT")
    for name, other in (("repeat", again), ("stale", stale), ("garbage", garbage), ("scrambled", scrambled), ("off", off)):
        for a_, b_ in zip(first, other):
            assert np.array_equal(a_, b_, equal_nan=True), name


def test_forcing_the_block_kernel_keeps_the_c4_gradient_right(eng):
    """with the forward kernel forced to 'b4' the steady-state value-and-gradient of a C4-like shape used to fall
    through to the LDS adjoint, which ignores the per-draw step counts.  Per-draw step counts now always reach the adjoint that
    honours them."""
    w = W.varmax(B=16, n=300)
    args = (w["y"],) + tuple(w[x] for x in KEYS)
    kw = dict(filter_type="standard", mvn_const="exact", h_diagonal=True)
    try:
        eng.set_forward_kernel("b4")
        lp, _, g = eng.value_and_grad(*args, **kw)
        lpx, _, gx = eng.value_and_grad(*args, exact_recursion=True, **kw)
    finally:
        eng.set_forward_kernel("auto")
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    for nm in GRADS:
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        # (with 'b4' forced the every-step gradient comes from the LDS adjoint, which returns the ENTRYWISE gradient of a dense P0 / H;
        # the scalar-observation adjoints return the symmetric part of g_P0 and the diagonal of g_H -- include/kf.h -- so compare those)
        if nm == "P0":
            a_, x_ = a_ + np.swapaxes(a_, -1, -2), x_ + np.swapaxes(x_, -1, -2)
        if nm == "H":
            a_, x_ = np.einsum("bii->bi", a_), np.einsum("bii->bi", x_)
        assert np.max(np.abs(a_ - x_)) / (1.0 + np.max(np.abs(x_))) < 1e-10, nm


# ----------------------------------------------------------------------------------------------
# dense H with p > 1 by whitening (csrc/kf_forward_whiten.hip)
# ----------------------------------------------------------------------------------------------
def _fwd_np(res):
    out = {"logp": res.logp.cpu().numpy(), "info": res.info.cpu().numpy()}
    for k in ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs"):
        out[k] = getattr(res, k).cpu().numpy()
    return out


@pytest.mark.parametrize("ft", ["standard", "cholesky"])
@pytest.mark.parametrize("mvn", ["exact", "reference"])
@pytest.mark.parametrize("shape", [(2, 4, 2), (3, 9, 3), (5, 15, 5), (5, 10, 5), (8, 12, 4), (4, 20, 6), (3, 33, 5), (3, 60, 5)])
def test_dense_H_runs_on_the_scalar_observation_kernels_by_whitening(eng, monkeypatch, ft, mvn, shape):
    """Reference Standard / Cholesky updates with any positive definite H (kalman_filter.py:262-275, 289-305): the whitened path
    against the numpy oracle (1e-8 on logp, 1e-9 on tapes and ll_obs) and against the dense LDS kernel it replaces, with whole
    rows missing and intercepts; the kernel the call takes is a diagonal-H one."""
    from oracle import kalman_numpy as kn

    p, m, r = shape
    w = W.random_dense(B=24, n=90, p=p, m=m, r=r, seed=100 + p + m, h_diag=False)
    y = w["y"].copy()
    y[[4, 30, 31]] = np.nan
    rng = np.random.default_rng(p * m)
    c, d = 0.05 * rng.standard_normal((24, m)), 0.1 * rng.standard_normal((24, p))
    args = (y,) + tuple(w[k] for k in KEYS)
    assert "lds" not in eng.forward_kernel_name(24, 90, p, m, r, filter_type=ft)
    got = _fwd_np(eng.forward(*args, c, d, filter_type=ft, mvn_const=mvn))
    monkeypatch.setenv("KF_FORWARD_WHITEN", "0")
    dense_name = eng.forward_kernel_name(24, 90, p, m, r, filter_type=ft)   # (m > 48: the HBM-workspace path in front of the LDS kernel)
    assert "lds" in dense_name or "big" in dense_name
    dense = _fwd_np(eng.forward(*args, c, d, filter_type=ft, mvn_const=mvn))
    monkeypatch.delenv("KF_FORWARD_WHITEN")
    # (cholesky_triangle="correct": the build never reproduces the reference's wrong-triangle solve for p > 1)
    ref = kn.kalman_filter(*args, c=c, d=d, filter_type=ft, mvn_const=mvn, cholesky_triangle="correct")
    assert not got["info"].any()
    for other, tag in ((ref, "oracle"), (dense, "dense kernel")):
        tol = 1e-8 * np.maximum(1.0, np.abs(other["logp"]) / 1e4)
        assert np.all(np.abs(got["logp"] - other["logp"]) <= tol), tag
        for k in ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs"):
            assert_allclose(got[k], other[k], rtol=1e-9, atol=1e-9, err_msg=f"{tag} {k}")
    assert got["ll_obs"][0, 4] == 0.0 and got["ll_obs"][0, 30] == 0.0   # whole-row NaN: no update, ll 0 (:195-213, :268)


def test_dense_H_falls_back_where_whitening_does_not_apply(eng):
    """A partially missing row (L^-1 would spread the NaN: the row-deletion semantics need the original observations) and an H
    that is not positive definite raise the device flag: the dense kernel redoes the batch behind the whitened one, with no host
    synchronisation in between -- results are the dense path's, per-draw data and one-draw batches included."""
    from oracle import kalman_numpy as kn

    w = W.random_dense(B=10, n=60, p=3, m=6, r=3, seed=5, h_diag=False)
    y = w["y"].copy()
    y[7, 1] = np.nan                      # one cell of a row
    y[20] = np.nan
    args = (y,) + tuple(w[k] for k in KEYS)
    got = _fwd_np(eng.forward(*args, filter_type="standard", mvn_const="exact"))
    ref = kn.kalman_filter(*args, filter_type="standard", mvn_const="exact", partial_missing="delete")
    assert np.all(np.abs(got["logp"] - ref["logp"]) <= 1e-8 * np.maximum(1.0, np.abs(ref["logp"]) / 1e4))
    for k in ("filtered_states", "predicted_covs", "ll_obs"):
        assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-9, err_msg=k)
    # per-draw series, one of them with a partially missing row
    ys = np.repeat(w["y"][None], 10, 0)
    ys[3, 11, 2] = np.nan
    got = _fwd_np(eng.forward(ys, *[w[k] for k in KEYS], filter_type="cholesky", mvn_const="exact"))
    for b in (0, 3, 9):
        rb = kn.kalman_filter(ys[b], *[w[k][b:b + 1] for k in KEYS], filter_type="cholesky", mvn_const="exact",
                              cholesky_triangle="correct", partial_missing="delete")
        assert abs(got["logp"][b] - rb["logp"][0]) <= 1e-8 * max(1.0, abs(rb["logp"][0]) / 1e4)
        assert_allclose(got["predicted_states"][b], rb["predicted_states"][0], rtol=1e-9, atol=1e-9)
    # H not positive definite in one draw: the dense path's answer (F may still be PD: H only enters through F = Z P Z' + H)
    w2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    w2["H"][4] = 0.0
    got = _fwd_np(eng.forward(w["y"], *[w2[k] for k in KEYS], filter_type="standard", mvn_const="exact"))
    ref = kn.kalman_filter(w["y"], *[w2[k] for k in KEYS], filter_type="standard", mvn_const="exact")
    assert np.all(np.abs(got["logp"] - ref["logp"]) <= 1e-8 * np.maximum(1.0, np.abs(ref["logp"]) / 1e4))
    assert_allclose(got["filtered_covs"], ref["filtered_covs"], rtol=1e-9, atol=1e-9)


# ----------------------------------------------------------------------------------------------
# time-varying T / Z / H / c / d on the block tiles (csrc/kf_forward_b4.hip: kf_fwd_b4tv_kernel)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("tv", [("T",), ("Z",), ("H", "d"), ("c",), ("T", "Z", "H", "c", "d")])
@pytest.mark.parametrize("m", [1, 2, 3, 6, 7, 10])
def test_time_varying_block_kernel_matches_oracle(eng, monkeypatch, ft, tv, m):
    """Reference: 3-D system matrices, time on axis 0, slice t used by the update AND the predict of step t
    (filters/utilities.py:1-20, kalman_filter.py:79-124, 231-252).  p = 1, m <= 10, R / Q static: the block kernel with per-step
    slices, against the numpy oracle (1e-8 on logp, 1e-9 on the tapes) and against the LDS kernel it replaces; missing observations,
    static intercepts beside time-varying ones, partial tape requests."""
    from oracle import kalman_numpy as kn

    B, n, r = 9, 70, max(1, m // 2)
    w = W.random_dense(B, n, 1, m, r, seed=900 + m)
    rng = np.random.default_rng(m)
    mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"],
                c=0.1 * rng.standard_normal((B, m)), d=0.1 * rng.standard_normal((B, 1)))
    for name in tv:
        x = np.repeat(mats[name][:, None], n, axis=1)
        wob = 0.8 + 0.2 * np.cos(0.37 * np.arange(n) + len(name))
        mats[name] = x * wob.reshape((1, n) + (1,) * (x.ndim - 2))
        if name in ("Z", "c", "d"):
            mats[name] = mats[name] + 0.05 * rng.standard_normal(mats[name].shape)
    y = w["y"].copy()
    y[[4, 33, 34]] = np.nan
    args = (y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"], mats["c"], mats["d"])
    assert eng.forward_kernel_name(B, n, 1, m, r, filter_type=ft, time_varying=tv) == "kf_fwd_b4tv_kernel"
    res = eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv)
    got = _fwd_np(res)
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
    # logp only / predicted tapes only: the same numbers
    lo = eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv, store_ll_obs=False, store_filtered=False,
                     store_predicted=False)
    assert np.array_equal(lo.logp.cpu().numpy(), got["logp"])
    po = eng.forward(*args, filter_type=ft, mvn_const="exact", time_varying=tv, store_filtered=False)
    assert np.array_equal(po.predicted_covs.cpu().numpy(), got["predicted_covs"])


def test_time_varying_block_kernel_edges(eng):
    """One step, one draw, a batch beyond the chip's wavefront slots, a series that is not a multiple of the 32-step block, per-draw
    data, F <= 0 flagged; R / Q time-varying with r > 4 keeps the LDS kernel."""
    from oracle import kalman_numpy as kn

    for B, n, m in ((1, 1, 4), (1, 33, 10), (5000, 40, 5), (3, 97, 9)):
        w = W.random_dense(B, n, 1, m, 2, seed=B + n)
        Tt = np.repeat(w["T"][:, None], n, axis=1) * (0.9 + 0.1 * np.cos(np.arange(n)))[None, :, None, None]
        args = (w["y"], w["a0"], w["P0"], Tt, w["Z"], w["R"], w["H"], w["Q"])
        res = eng.forward(*args, mvn_const="exact", time_varying=("T",))
        k = min(B, 6)
        ref = kn.kalman_filter_time_varying(w["y"], *[a[:k] for a in args[1:]], time_varying=("T",), mvn_const="exact")
        assert np.max(np.abs(res.logp.cpu().numpy()[:k] - ref["logp"])) < 1e-8
        assert_allclose(res.predicted_covs.cpu().numpy()[:k], ref["predicted_covs"], rtol=1e-9, atol=1e-9)
        assert_allclose(res.filtered_states.cpu().numpy()[:k], ref["filtered_states"], rtol=1e-9, atol=1e-9)
    w = W.random_dense(4, 50, 1, 6, 2, seed=3)
    ys = np.repeat(w["y"][None], 4, 0) + 0.1 * np.random.default_rng(0).standard_normal((4, 50, 1))
    Zt = np.repeat(w["Z"][:, None], 50, axis=1) * (1.0 + 0.1 * np.sin(np.arange(50)))[None, :, None, None]
    res = eng.forward(ys, w["a0"], w["P0"], w["T"], Zt, w["R"], w["H"], w["Q"], mvn_const="exact", time_varying=("Z",))
    for b in range(4):
        ref = kn.kalman_filter_time_varying(ys[b], *[w[k][b:b + 1] for k in ("a0", "P0", "T")], Zt[b:b + 1],
                                            *[w[k][b:b + 1] for k in ("R", "H", "Q")], time_varying=("Z",), mvn_const="exact")
        assert abs(res.logp[b].item() - ref["logp"][0]) < 1e-8
    Hneg = w["H"].copy()
    Hneg[2] = -50.0
    res = eng.forward(w["y"], w["a0"], w["P0"], np.repeat(w["T"][:, None], 50, axis=1), w["Z"], w["R"], Hneg, w["Q"],
                      mvn_const="exact", time_varying=("T",))
    info = res.info.cpu().numpy()
    assert info[2] != 0 and not info[[0, 1, 3]].any()
    # (round 5: R / Q time-varying ride the block kernel too while r <= 4; beyond that the LDS kernel)
    assert eng.forward_kernel_name(4, 50, 1, 6, 2, time_varying=("T", "Q")) == "kf_fwd_b4tv_kernel"
    assert eng.forward_kernel_name(4, 50, 1, 6, 5, time_varying=("T", "Q")) == "kf_fwd_lds_kernel"


# ---- RTS smoother: the time-parallel kernels behind the filter's convergence point (kf_smoother_b4_steady.hip) -------------------------
def _smooth_both(eng, monkeypatch, w, res, **kw):
    args = (w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    monkeypatch.setenv("KF_SMOOTHER_SPLIT", "1")
    sa, sP = eng.smooth(*args, **kw)
    monkeypatch.setenv("KF_SMOOTHER_SPLIT", "0")
    xa, xP = eng.smooth(*args, **kw)
    monkeypatch.delenv("KF_SMOOTHER_SPLIT")
    return sa.cpu().numpy(), sP.cpu().numpy(), xa.cpu().numpy(), xP.cpu().numpy()


def _rel(a_, x_):
    B = a_.shape[0]
    return (np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / (1.0 + np.max(np.abs(x_).reshape(B, -1), axis=1))).max()


@pytest.mark.parametrize("fwd", ["every_step", "steady_state"])
@pytest.mark.parametrize("m", [5, 6, 8, 9, 10])
def test_smoother_behind_the_convergence_point_matches_every_step(eng, monkeypatch, m, fwd):
    """Mean as a prefix scan, covariance recursed to its fixpoint and filled, the steps before the filter's convergence point in
    kf_smooth_b4_kernel from the hand-over moments: against every step in that kernel (1e-11 of a draw's largest entry: the
    tape tolerance 1e-14 times the conditioning of P_hat) and against the oracle on the same tapes (1e-9, the suite's bound)."""
    from oracle import kalman_numpy as kn

    w = W.structural(B=48, n=700) if m == 10 else W.random_dense(B=48, n=300 + 37 * m, p=1, m=m, r=max(1, m // 2), seed=130 + m)
    y = w["y"].copy()
    y[11] = np.nan   # an early missing observation: inside the part kf_smooth_b4_kernel keeps
    res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False,
                      exact_recursion=(fwd == "every_step"))
    sa, sP, xa, xP = _smooth_both(eng, monkeypatch, w, res)
    assert np.all(np.isfinite(sa)) and np.all(np.isfinite(sP))
    assert not (np.array_equal(sa, xa) and np.array_equal(sP, xP)), "the switch selected the same path twice"
    assert _rel(sa, xa) < 1e-11 and _rel(sP, xP) < 1e-11, (_rel(sa, xa), _rel(sP, xP))
    na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
    assert_allclose(sa, na, rtol=1e-9, atol=1e-9)
    assert_allclose(sP, nP, rtol=1e-9, atol=1e-9)


def test_smoother_behind_the_convergence_point_edges(eng, monkeypatch):
    """A missing observation late in the series (the filtered covariance of that step differs: the range ends above it), a missing
    LAST observation (nothing qualifies: the whole series stays in kf_smooth_b4_kernel), draws whose filter never converges, a
    series that ends inside the first 64-step block, KF_FLAG_EXACT_RECURSION (never split)."""
    from oracle import kalman_numpy as kn

    w = _never_converging(W.structural(B=40, n=520), [0, 9, 39])
    for nan_at in ([300], [519], [255, 256, 450], []):
        y = w["y"].copy()
        y[nan_at] = np.nan
        res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False)
        sa, sP, xa, xP = _smooth_both(eng, monkeypatch, w, res)
        assert _rel(sa, xa) < 1e-11 and _rel(sP, xP) < 1e-11, (nan_at, _rel(sa, xa), _rel(sP, xP))
        if nan_at == [519]:
            assert np.array_equal(sa, xa) and np.array_equal(sP, xP)   # P_f[n-1] is a one-off: every step in the sequential kernel
        na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
        ok = [b for b in range(40) if b not in (0, 9, 39)]   # (the diffuse draws, P0 = 1e3 I with Q ~ 1e-9: cond(P_hat) ~ 1e8, both paths
        assert_allclose(sa[ok], na[ok], rtol=1e-9, atol=1e-9)   #  sit 1e-7 from the oracle's solve there -- and 1e-11 from each other)
        assert_allclose(sP[ok], nP[ok], rtol=1e-9, atol=1e-9)
        assert_allclose(sa, na, rtol=1e-5, atol=1e-5)
    ws = W.structural(B=12, n=150)
    res = eng.forward(*[ws[k] for k in ("y",) + KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False)
    sa, sP, xa, xP = _smooth_both(eng, monkeypatch, ws, res)
    assert _rel(sa, xa) < 1e-11 and _rel(sP, xP) < 1e-11
    sa, sP, xa, xP = _smooth_both(eng, monkeypatch, ws, res, exact_recursion=True)
    assert np.array_equal(sa, xa) and np.array_equal(sP, xP)


def test_smoother_split_is_the_default_for_long_series_and_keeps_the_pinv_fallback(eng, monkeypatch):
    """>= 1536 steps: AUTO takes the split (bitwise the forced run).  A draw whose P_hat is singular fails the Cholesky pivot test
    and is left to kf_smooth_b4_kernel and the pseudo-inverse kernel behind it, next to draws that take the split."""
    from oracle import kalman_numpy as kn

    w = W.structural(B=24, n=1600)
    res = eng.forward(*[w[k] for k in ("y",) + KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False)
    args = (w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    da, dP = eng.smooth(*args)
    monkeypatch.setenv("KF_SMOOTHER_SPLIT", "1")
    fa, fP = eng.smooth(*args)
    monkeypatch.setenv("KF_SMOOTHER_SPLIT", "0")
    xa, xP = eng.smooth(*args)
    monkeypatch.delenv("KF_SMOOTHER_SPLIT")
    assert np.array_equal(da.cpu().numpy(), fa.cpu().numpy()) and np.array_equal(dP.cpu().numpy(), fP.cpu().numpy())
    assert not np.array_equal(dP.cpu().numpy(), xP.cpu().numpy())
    assert _rel(da.cpu().numpy(), xa.cpu().numpy()) < 1e-11 and _rel(dP.cpu().numpy(), xP.cpu().numpy()) < 1e-11
    # singular P_hat: two perfectly correlated states inside an m = 6 model
    ws = W.random_dense(B=6, n=200, p=1, m=6, r=3, seed=5)
    for b in (1, 4):
        ws["T"][b, 1, :] = ws["T"][b, 0, :]
        ws["R"][b, 1, :] = ws["R"][b, 0, :]
        ws["P0"][b] = 0.0
        ws["P0"][b, 2:, 2:] = np.eye(4)
    res = eng.forward(*[ws[k] for k in ("y",) + KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False)
    sa, sP, xa, xP = _smooth_both(eng, monkeypatch, ws, res)
    na, nP = kn.kalman_smoother(ws["T"], ws["R"], ws["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
    assert_allclose(sa, na, rtol=1e-7, atol=1e-7)
    assert_allclose(sP, nP, rtol=1e-7, atol=1e-7)
    assert_allclose(sa, xa, rtol=1e-7, atol=1e-7)


def test_block_smoother_hands_numerically_singular_draws_to_the_pinv_kernel(eng, monkeypatch):
    """Rank-one state noise behind a long series: cond(P_hat) = 1.4e12 on draw 4 of this batch (found by scripts/r04/fuzz_r04b.py).
    The Newton-Schulz kernel's residual test passed there while the error of its inverse along the near-null directions made the
    smoothed recursion unstable -- 1e30 where the reference's pinv gives 1e0.  The kernel now estimates the conditioning
    (||X|| ||P_hat||) and hands such a draw to the elimination kernel (and that one, at pivots below 1e-9, to the pseudo-inverse
    kernel): the result is that kernel's, finite and as close to the oracle as cond x eps allows; the well-conditioned draws of the
    batch are untouched."""
    from oracle import kalman_numpy as kn

    w = W.random_dense(8, 200, 1, 10, 1, seed=618827647)
    res = eng.forward(*[w[k] for k in ("y",) + KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False)
    args = (w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
    for split in ("0", "1"):
        monkeypatch.setenv("KF_SMOOTHER_SPLIT", split)
        sa, sP = [x.cpu().numpy() for x in eng.smooth(*args)]
        monkeypatch.delenv("KF_SMOOTHER_SPLIT")
        na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
        assert np.all(np.isfinite(sa)) and np.all(np.isfinite(sP))
        err = np.max(np.abs(sP - nP).reshape(8, -1), axis=1) / (1.0 + np.max(np.abs(nP).reshape(8, -1), axis=1))
        assert err[4] < 5e-2, err            # (the elimination kernel and the LDS kernel sit at 1e-2 on this draw as well)
        assert np.all(err[[0, 1, 3, 6, 7]] < 1e-8), err
    monkeypatch.setenv("KF_SMOOTHER_B4", "0")
    ea, eP = [x.cpu().numpy() for x in eng.smooth(*args)]
    monkeypatch.delenv("KF_SMOOTHER_B4")
    assert_allclose(sP[4], eP[4], rtol=1e-9, atol=1e-9)   # the flagged draw IS the elimination kernel's (then, if that flags too, pinv)


# ---- logp + gradient behind the lane kernel (m <= 4) with the steady-state adjoint -------------------------------------------------------
@pytest.mark.parametrize("ft", ["standard", "cholesky", "univariate", "single"])
@pytest.mark.parametrize("m", [1, 2, 3])
def test_value_and_grad_behind_the_lane_kernel_with_the_steady_state_adjoint(eng, monkeypatch, m, ft):
    """AUTO keeps m <= 3 on the lane kernel; its value-and-gradient call now takes the steps behind each draw's convergence point
    with kf_bwd_steady_scan_kernel (a_t and the converged covariance from the compact tapes) + the Riccati adjoint and sweeps only
    the steps before it (kf_bwd_b4_kernel on the compact tapes, per-draw step counts).  Against the full sweep (1e-10), torch
    autograd on the first draws (1e-9); intercepts, a missing observation, per-draw cotangents; more draws than one wavefront
    packs and a batch that is not a multiple of it."""
    from oracle import kalman_torch as kt

    nb = 2100 if m == 3 else 37   # (AUTO hands m = 3 to the block kernels up to 2048 draws)
    w = W.arma11(B=nb, n=300) if m == 2 else W.random_dense(B=nb, n=300, p=1, m=m, r=1, seed=60 + m)
    y = w["y"].copy()
    y[9] = np.nan
    rng = np.random.default_rng(m)
    B = w["B"]
    c, d = 0.05 * rng.standard_normal((B, m)), 0.1 * rng.standard_normal((B, 1))
    gl = 1.0 + 0.3 * rng.standard_normal(B)
    args = (y,) + tuple(w[x] for x in KEYS)
    kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
    assert eng.forward_kernel_name(B, 300, 1, m, w["r"], filter_type=ft) == "kf_fwd_lane_kernel"
    lp, info, g = eng.value_and_grad(*args, **kw)
    lpx, infox, gx = eng.value_and_grad(*args, exact_recursion=True, **kw)
    assert not info.cpu().numpy().any() and not infox.cpu().numpy().any()
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    differs = False
    for nm in GRADS:
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        differs = differs or not np.array_equal(a_, x_)
        scale = 1.0 + np.max(np.abs(x_.reshape(B, -1)), axis=1)
        assert (np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / scale).max() < 1e-10, nm
    assert differs, "the steady-state adjoint did not run"
    k = 3
    rlp, _, rg = kt.logp_and_grads(y, *[w[x][:k] for x in KEYS], c=c[:k], d=d[:k], filter_type=ft, mvn_const="exact", g_logp=gl[:k])
    for nm in GRADS:
        a_, ref = g[nm].cpu().numpy()[:k].reshape(rg[nm].shape), rg[nm]
        if nm in ("P0", "Q"):
            a_, ref = a_ + np.swapaxes(a_, -1, -2), ref + np.swapaxes(ref, -1, -2)
        assert np.max(np.abs(a_ - ref)) / (1.0 + np.max(np.abs(ref))) < 1e-9, f"{nm} vs autograd"


def test_value_and_grad_behind_the_lane_kernel_edges(eng, monkeypatch):
    """Draws that never converge next to draws that do (ts = T: the whole sweep is the exact one), a missing observation late in the
    series, the switch (KF_VG_STEADY=0) and the batch rule (beyond two rounds of wavefront slots AUTO keeps the packed full sweep:
    bitwise the flag's result)."""
    w = W.arma11(B=50, n=400)
    w["T"][[3, 17]] = np.array([[1.0, 1.0], [0.0, 1.0]])     # a unit root: the predicted covariance keeps growing
    w["P0"][[3, 17]] = 1e3 * np.eye(2)
    y = w["y"].copy()
    y[[20, 300]] = np.nan     # (the phase can start two 16-step blocks behind the last missing observation: step 320 of 400)
    args = (y,) + tuple(w[x] for x in KEYS)
    lp, info, g = eng.value_and_grad(*args, mvn_const="exact")
    lpx, _, gx = eng.value_and_grad(*args, mvn_const="exact", exact_recursion=True)
    monkeypatch.setenv("KF_VG_STEADY", "0")
    lps, _, gs = eng.value_and_grad(*args, mvn_const="exact")
    monkeypatch.delenv("KF_VG_STEADY")
    assert_allclose(lp.cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    for nm in GRADS:
        a_, x_ = g[nm].cpu().numpy(), gx[nm].cpu().numpy()
        assert np.all(np.isfinite(a_)), nm
        scale = 1.0 + np.max(np.abs(x_.reshape(50, -1)), axis=1)
        assert (np.max(np.abs(a_ - x_).reshape(50, -1), axis=1) / scale).max() < 1e-10, nm
        assert (np.max(np.abs(gs[nm].cpu().numpy() - x_).reshape(50, -1), axis=1) / scale).max() < 1e-10, nm   # the switch: the full sweep
    assert any(not np.array_equal(g[nm].cpu().numpy(), gs[nm].cpu().numpy()) for nm in GRADS), "the steady-state adjoint did not run"
    big = W.arma11(B=9000, n=64)
    argb = (big["y"],) + tuple(big[x] for x in KEYS)
    _, _, gb = eng.value_and_grad(*argb, mvn_const="exact")
    monkeypatch.setenv("KF_VG_STEADY", "0")
    _, _, gbs = eng.value_and_grad(*argb, mvn_const="exact")
    monkeypatch.delenv("KF_VG_STEADY")
    assert all(np.array_equal(gb[nm].cpu().numpy(), gbs[nm].cpu().numpy()) for nm in GRADS)   # AUTO: not beyond 32 x CUs draws


@pytest.mark.parametrize("m", [5, 6, 8])
def test_value_and_grad_takes_the_block_kernels_where_the_forward_alone_would_not(eng, m):
    """16384+ draws of a 5 <= m <= 8 model: AUTO runs the FORWARD alone on the packed row kernels, the value-and-gradient call on
    the block kernels (tile tape, 2.3 - 2.5 times faster there: scripts/r04/vg_rowshapes.py) -- the same bits as with them forced,
    and the row-kernel path's numbers to rounding."""
    B = 20000
    w = W.random_dense(B, 70, 1, m, 2, seed=m)
    assert eng.forward_kernel_name(B, 70, 1, m, 2).startswith("kf_fwd_row")
    args = (w["y"],) + tuple(w[x] for x in KEYS)
    lp, _, g = eng.value_and_grad(*args, mvn_const="exact")
    try:
        eng.set_forward_kernel("b4")
        lpb, _, gb = eng.value_and_grad(*args, mvn_const="exact")
        eng.set_forward_kernel("row")
        lpr, _, gr = eng.value_and_grad(*args, mvn_const="exact")
    finally:
        eng.set_forward_kernel("auto")
    assert np.array_equal(lp.cpu().numpy(), lpb.cpu().numpy())
    for nm in GRADS:
        assert np.array_equal(g[nm].cpu().numpy(), gb[nm].cpu().numpy()), nm
        a_, x_ = g[nm].cpu().numpy(), gr[nm].cpu().numpy()
        scale = 1.0 + np.max(np.abs(x_.reshape(B, -1)), axis=1)
        assert (np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / scale).max() < 1e-10, nm


def test_value_and_grad_takes_the_multi_observation_block_kernel_at_m8_large_batches(eng):
    """p = 3, m = 8 beyond 2048 draws: the forward alone runs on the packed row kernel, the value-and-gradient call on
    kf_fwd_b4q_kernel + the steady-state adjoint (the same bits as with the block kernels forced)."""
    B = 5000
    w = W.random_dense(B, 120, 3, 8, 2, seed=38)
    assert "rowsplit" in eng.forward_kernel_name(B, 120, 3, 8, 2, h_diagonal=True)
    args = (w["y"],) + tuple(w[x] for x in KEYS)
    kw = dict(mvn_const="exact", h_diagonal=True)
    lp, _, g = eng.value_and_grad(*args, **kw)
    try:
        eng.set_forward_kernel("b4")
        lpb, _, gb = eng.value_and_grad(*args, **kw)
        eng.set_forward_kernel("row")
        lpr, _, gr = eng.value_and_grad(*args, **kw)
    finally:
        eng.set_forward_kernel("auto")
    assert np.array_equal(lp.cpu().numpy(), lpb.cpu().numpy())
    for nm in GRADS:
        assert np.array_equal(g[nm].cpu().numpy(), gb[nm].cpu().numpy()), nm
        a_, x_ = g[nm].cpu().numpy(), gr[nm].cpu().numpy()
        if nm in ("P0", "H"):   # (the block path returns the symmetric / diagonal parts)
            continue
        scale = 1.0 + np.max(np.abs(x_.reshape(B, -1)), axis=1)
        assert (np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / scale).max() < 1e-9, nm


@pytest.mark.parametrize("m", [2, 10])
def test_steady_state_paths_replay_from_a_hip_graph(eng, monkeypatch, m):
    """The round's multi-kernel paths -- value-and-gradient with the order kernels, the time-parallel adjoint, the Riccati adjoint
    and the exact sweep (block kernels at m = 10, lane kernel at m = 2); the smoother with bisection, verification, scan, fill and
    the resumed sweep -- are plain launches on the caller's stream: captured once, replayed on new data, bitwise an eager run."""
    import torch

    B, n = 40, 260
    w = W.structural(B=B, n=n) if m == 10 else W.arma11(B=B, n=n)
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    monkeypatch.setenv("KF_SMOOTHER_SPLIT", "1")

    def step():
        vg = eng.value_and_grad(*dev, mvn_const="exact", exact_recursion=False)
        res = eng.forward(*dev, mvn_const="exact", store_predicted=False, store_ll_obs=False)
        sm = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs) if m == 10 else None
        return vg, sm

    step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        vg_c, sm_c = step()
    y2 = w["y"] + 0.3 * np.random.default_rng(1).standard_normal(w["y"].shape)
    dev[0].copy_(torch.as_tensor(y2, device=dev[0].device))
    graph.replay()
    graph.replay()   # (twice: the second forward launch is placed by the first replay's convergence points)
    torch.cuda.synchronize()
    vg_e, sm_e = step()
    torch.cuda.synchronize()
    assert torch.equal(vg_c[0], vg_e[0]) and torch.equal(vg_c[1], vg_e[1])
    for k in vg_e[2]:
        assert torch.equal(vg_c[2][k], vg_e[2][k]), k
    if sm_e is not None:
        assert torch.equal(sm_c[0], sm_e[0]) and torch.equal(sm_c[1], sm_e[1])
    lpx, _, gx = eng.value_and_grad(*dev, mvn_const="exact", exact_recursion=True)
    assert_allclose(vg_c[0].cpu().numpy(), lpx.cpu().numpy(), rtol=1e-12)
    for k in gx:
        a_, x_ = vg_c[2][k].cpu().numpy(), gx[k].cpu().numpy()
        assert np.max(np.abs(a_ - x_)) / (1.0 + np.max(np.abs(x_))) < 1e-9, k
