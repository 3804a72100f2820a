"""CPU oracle: numpy restatement of the reference Kalman filter / RTS smoother.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``pymc_statespace_amd``)
may import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

What it restates (all paths relative to /root/reference/pymc_statespace):

* ``filters/kalman_filter.py:195-213``  handle_missing_values
* ``filters/kalman_filter.py:215-223``  predict (incl. the 0.5*(P+P.T) symmetrisation)
* ``filters/kalman_filter.py:231-252``  kalman_step  (update THEN predict)
* ``filters/kalman_filter.py:255-284``  StandardFilter.update  (Joseph form, single log(2pi))
* ``filters/kalman_filter.py:287-318``  CholeskyFilter.update
* ``filters/kalman_filter.py:321-351``  SingleTimeseriesFilter.update
* ``filters/kalman_filter.py:444-505``  UnivariateFilter
* ``filters/kalman_filter.py:165-193``  _postprocess_scan_results
* ``filters/kalman_smoother.py:56-104``  KalmanSmoother  (``kalman_smoother``; and ``smoother_from_filter``: the same moments by the
  de Jong / Durbin-Koopman recursion on the filter's predicted tapes -- the checker of ``kf_smoother_from_filter``)

The reference evaluates ONE parameter draw per graph call; here every system
matrix may carry a leading batch axis ``B`` (the batch-of-draws axis the MI355X
build adds) and each draw is evaluated independently with the reference's
operation order.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this module
against (i) statsmodels 0.12.2 golden vectors (``tests/golden/*.npz``, made by
``tests/golden/make_golden.py``) -- the same external oracle the reference's own
parity test uses (tests/test_kalman_filter.py:226-241) -- and (ii) the known
answers recorded in BASELINE.md.  The reference itself cannot be imported here
(pytensor / pymc are not installed), so it is pinned through those vectors.

Three reference quirks change results for p > 1 and are explicit switches
(SURVEY.md section 8a rows A5/A7/A8):

``mvn_const``          "reference": StandardFilter adds log(2pi) once per step
                       (kalman_filter.py:281), CholeskyFilter adds it
                       ``y.shape[0]`` = p times (:314), UnivariateFilter once per
                       non-zero inner term (:503).  "exact": n_observed * log(2pi).
``cholesky_triangle``  "reference": ``SolveTriangular(lower=True)`` applied to
                       ``F_chol.T`` (:299, :307) reads only the lower triangle,
                       i.e. divides by diag(L).  "correct": true L^-T solve.
                       Identical for p == 1.
``partial_missing``    "reference": only the all-missing step is made safe
                       (``F + I*flag``); a partially missing step leaves a zero
                       row in F (singular).  "delete": proper row deletion
                       (masked rows/cols of H zeroed, unit diagonal inserted, the
                       deleted entries of v zeroed -- d is not masked by :195-213).
"""
from __future__ import annotations

import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))  # MVN_CONST, kalman_filter.py:16

FILTER_TYPES = ("standard", "cholesky", "univariate", "single")


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def _T(x):
    return np.swapaxes(x, -1, -2)


def _as_batch(x, ndim_core, B=None):
    """Give ``x`` a leading batch axis; returns (array, had_batch)."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == ndim_core:
        return x[None], False
    if x.ndim == ndim_core + 1:
        return x, True
    raise ValueError(f"expected ndim {ndim_core} or {ndim_core + 1}, got {x.ndim}")


def canonicalize(data, a0, P0, T, Z, R, H, Q, c=None, d=None):
    """Bring reference-shaped inputs to batched core shapes.

    Reference layouts: data (T,p,1) | (T,p); a0 (m,1) | (m,);
    P0 (m,m); T (m,m); Z (p,m); R (m,r); H (p,p); Q (r,r); c (m,1); d (p,1).
    Every system matrix may have one extra leading axis B.  ``data`` may be
    shared (T,p[,1]) or per-draw (B,T,p).
    Returns dict of arrays with a leading B axis (broadcast views) and ``batched``.
    """
    Z = np.asarray(Z, dtype=np.float64)
    T_ = np.asarray(T, dtype=np.float64)
    m = T_.shape[-1]
    p = Z.shape[-2]

    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3 and data.shape[-1] == 1 and data.shape[-2] == p:
        data = data[..., 0]  # (T,p,1) -> (T,p)
    if data.ndim == 2:
        data_b, data_batched = data[None], False
    elif data.ndim == 3:
        data_b, data_batched = data, True
    else:
        raise ValueError("data must be (T,p), (T,p,1) or (B,T,p)")

    a0 = np.asarray(a0, dtype=np.float64)
    if a0.shape[-1] == 1 and a0.ndim >= 2 and a0.shape[-2] == m:
        a0 = a0[..., 0]
    a0_b, f0 = _as_batch(a0, 1)
    P0_b, f1 = _as_batch(P0, 2)
    T_b, f2 = _as_batch(T_, 2)
    Z_b, f3 = _as_batch(Z, 2)
    R_b, f4 = _as_batch(R, 2)
    H_b, f5 = _as_batch(H, 2)
    Q_b, f6 = _as_batch(Q, 2)
    r = R_b.shape[-1]

    if c is None:
        c_b, f7 = np.zeros((1, m)), False  # initialize_intercepts :36-51 (zeros)
    else:
        c = np.asarray(c, dtype=np.float64)
        if c.shape[-1] == 1 and c.ndim >= 2 and c.shape[-2] == m:
            c = c[..., 0]
        c_b, f7 = _as_batch(c, 1)
    if d is None:
        d_b, f8 = np.zeros((1, p)), False
    else:
        d = np.asarray(d, dtype=np.float64)
        if d.shape[-1] == 1 and d.ndim >= 2 and d.shape[-2] == p:
            d = d[..., 0]
        d_b, f8 = _as_batch(d, 1)

    arrs = [data_b, a0_b, P0_b, T_b, Z_b, R_b, H_b, Q_b, c_b, d_b]
    Bs = {x.shape[0] for x in arrs if x.shape[0] != 1}
    if len(Bs) > 1:
        raise ValueError(f"inconsistent batch sizes {Bs}")
    B = Bs.pop() if Bs else 1
    batched = any([data_batched, f0, f1, f2, f3, f4, f5, f6, f7, f8])

    def bc(x):
        return np.broadcast_to(x, (B,) + x.shape[1:])

    out = dict(
        data=bc(data_b), a0=bc(a0_b), P0=bc(P0_b), T=bc(T_b), Z=bc(Z_b), R=bc(R_b),
        H=bc(H_b), Q=bc(Q_b), c=bc(c_b), d=bc(d_b),
        B=B, n=data_b.shape[1], p=p, m=m, r=r, batched=batched,
    )
    assert out["a0"].shape == (B, m) and out["P0"].shape == (B, m, m)
    assert out["T"].shape == (B, m, m) and out["Z"].shape == (B, p, m)
    assert out["R"].shape == (B, m, r) and out["H"].shape == (B, p, p)
    assert out["Q"].shape == (B, r, r) and out["data"].shape[2] == p
    return out


# --------------------------------------------------------------------------- #
# one time step, batched over draws
# --------------------------------------------------------------------------- #
def handle_missing_values(y, Z, H, partial_missing="reference"):
    """kalman_filter.py:195-213.  y (B,p), Z (B,p,m), H (B,p,p)."""
    nan_mask = np.isnan(y)  # (B,p)
    all_nan = nan_mask.all(axis=-1)  # (B,)
    W = (~nan_mask).astype(np.float64)  # diagonal of the masking matrix
    Z_masked = W[..., None] * Z  # W.dot(Z): zero the missing rows
    H_masked = W[..., None] * H  # W.dot(H): zero the missing ROWS only
    if partial_missing == "delete":
        H_masked = H_masked * W[..., None, :]  # also the columns
    y_masked = np.where(nan_mask, 0.0, y)
    return y_masked, Z_masked, H_masked, all_nan, nan_mask


def predict(a, P, c, T, RQR):
    """kalman_filter.py:215-223.  ``RQR`` = R Q R^T (static, hoisted)."""
    a_hat = np.einsum("bij,bj->bi", T, a) + c
    P_hat = T @ P @ _T(T) + RQR
    P_hat = 0.5 * (P_hat + _T(P_hat))
    return a_hat, P_hat


def _n_const(kind, mvn_const, p, nan_mask):
    """How many log(2pi) the step adds, per draw."""
    if mvn_const == "exact":
        return (~nan_mask).sum(axis=-1).astype(np.float64)
    if kind in ("standard", "single"):
        return np.ones(nan_mask.shape[0])  # :281, :349
    if kind == "cholesky":
        return np.full(nan_mask.shape[0], float(p))  # :309, :314
    raise ValueError(kind)


def update_standard(a, P, y, d, Z, H, all_nan, nan_mask, mvn_const, partial_missing):
    """StandardFilter.update, kalman_filter.py:255-284."""
    B, p = y.shape
    m = a.shape[-1]
    eye_p = np.eye(p)
    eye_m = np.eye(m)
    flag = all_nan.astype(np.float64)[:, None, None]

    v = y - np.einsum("bij,bj->bi", Z, a) - d
    if partial_missing == "delete":
        v = np.where(nan_mask, 0.0, v)  # a deleted row carries no innovation (d is not masked upstream)
    PZT = P @ _T(Z)
    F = Z @ PZT + H
    F_solve = F + eye_p * flag
    if partial_missing == "delete":
        # unit diagonal on deleted rows so the p x p system stays regular
        F_solve = F_solve + (nan_mask & ~all_nan[:, None])[..., None] * eye_p
    F_inv = np.linalg.solve(F_solve, np.broadcast_to(eye_p, F_solve.shape))
    K = PZT @ F_inv
    I_KZ = eye_m - K @ Z
    a_filtered = a + np.einsum("bij,bj->bi", K, v)
    P_filtered = I_KZ @ P @ _T(I_KZ) + K @ H @ _T(K)  # Joseph form

    inner = np.einsum("bi,bij,bj->b", v, F_inv, v)
    F_det = F if partial_missing != "delete" else F_solve - eye_p * flag
    with np.errstate(divide="ignore", invalid="ignore"):
        logdet = np.log(np.linalg.det(F_det))
    nconst = _n_const("standard", mvn_const, p, nan_mask)
    ll = np.where(all_nan, 0.0, -0.5 * (nconst * LOG_2PI + logdet + inner))
    return a_filtered, P_filtered, ll


def _solve_lower(L, X):
    """SolveTriangular(lower=True)(L, X): uses ONLY the lower triangle of L."""
    Ltri = np.tril(L)
    return np.linalg.solve(Ltri, X)


def update_cholesky(a, P, y, d, Z, H, all_nan, nan_mask, mvn_const, partial_missing,
                    cholesky_triangle):
    """CholeskyFilter.update, kalman_filter.py:287-318."""
    B, p = y.shape
    m = a.shape[-1]
    eye_p = np.eye(p)
    eye_m = np.eye(m)
    flag = all_nan.astype(np.float64)[:, None, None]

    v = y - np.einsum("bij,bj->bi", Z, a) - d
    if partial_missing == "delete":
        v = np.where(nan_mask, 0.0, v)  # a deleted row carries no innovation (d is not masked upstream)
    PZT = P @ _T(Z)
    F = Z @ PZT + H + eye_p * flag
    if partial_missing == "delete":
        F = F + (nan_mask & ~all_nan[:, None])[..., None] * eye_p
    F_chol = np.linalg.cholesky(F)

    if cholesky_triangle == "reference":
        # solve_lower_triangular(F_chol.T, .) reads the lower triangle of an
        # upper-triangular matrix = its diagonal (kalman_filter.py:18, :299, :307)
        def solve_LT(X):
            return _solve_lower(_T(F_chol), X)
    else:
        def solve_LT(X):
            return np.linalg.solve(_T(F_chol), X)

    K = _T(solve_LT(_solve_lower(F_chol, _T(PZT)))) * (1.0 - flag)
    I_KZ = eye_m - K @ Z
    a_filtered = a + np.einsum("bij,bj->bi", K, v)
    P_filtered = I_KZ @ P @ _T(I_KZ) + K @ H @ _T(K)

    inner_term = solve_LT(_solve_lower(F_chol, v[..., None]))[..., 0]
    nconst = _n_const("cholesky", mvn_const, p, nan_mask)
    logdiag = np.log(np.diagonal(F_chol, axis1=-2, axis2=-1)).sum(axis=-1)
    ll = np.where(
        all_nan, 0.0, -0.5 * (nconst * LOG_2PI + np.einsum("bi,bi->b", v, inner_term)) - logdiag
    )
    return a_filtered, P_filtered, ll


def update_single(a, P, y, d, Z, H, all_nan, nan_mask, mvn_const):
    """SingleTimeseriesFilter.update, kalman_filter.py:321-351 (p == 1).

    Quirk kept: ``y_hat = Z a - d`` (sign of d flipped, :335)."""
    B, p = y.shape
    if p != 1:
        # assert_data_is_1d, kalman_filter.py:19, :329
        raise AssertionError("UnivariateTimeSeries filter requires data be at most 1-dimensional")
    m = a.shape[-1]
    eye_m = np.eye(m)
    flag = all_nan.astype(np.float64)

    y_hat = np.einsum("bij,bj->bi", Z, a) - d
    v = y - y_hat  # (B,1)
    PZT = P @ _T(Z)  # (B,m,1)
    F = (Z @ PZT + H)[:, 0, 0] + flag  # (B,)
    K = PZT / F[:, None, None]
    I_KZ = eye_m - K @ Z
    a_filtered = a + K[..., 0] * v
    P_filtered = I_KZ @ P @ _T(I_KZ) + K @ H @ _T(K)
    nconst = _n_const("single", mvn_const, p, nan_mask)
    with np.errstate(divide="ignore", invalid="ignore"):
        ll = np.where(all_nan, 0.0, -0.5 * (nconst * LOG_2PI + np.log(F) + v[:, 0] ** 2 / F))
    return a_filtered, P_filtered, ll


def update_univariate(a, P, y, d, Z, H, nan_mask, mvn_const):
    """UnivariateFilter, kalman_filter.py:460-505: p sequential scalar updates.

    Uses diag(H) only (:493).  ``y, Z, H`` are already masked."""
    B, p = y.shape
    ll_inner = np.zeros((B, p))
    for i in range(p):
        Z_row = Z[:, i, :]  # (B,m)
        v = y[:, i] - np.einsum("bj,bj->b", Z_row, a) - d[:, i]
        PZT = np.einsum("bij,bj->bi", P, Z_row)  # (B,m)
        F = np.einsum("bj,bj->b", Z_row, PZT) + H[:, i, i]
        F_zero_flag = ((F == 0) | nan_mask[:, i]).astype(np.float64)
        F = F + 1e-8 * F_zero_flag
        K = PZT / F[:, None] * (1.0 - F_zero_flag)[:, None]
        a = a + K * (v * (1.0 - F_zero_flag))[:, None]
        P = P - np.einsum("bi,bj->bij", K, K) * (F * (1.0 - F_zero_flag))[:, None, None]
        ll_inner[:, i] = (np.log(F) + v ** 2 / F) * (1.0 - F_zero_flag)
    if mvn_const == "exact":
        nconst = (~nan_mask).sum(axis=-1).astype(np.float64)
    else:
        nconst = (ll_inner != 0).sum(axis=-1).astype(np.float64)  # :503
    ll = -0.5 * (nconst * LOG_2PI + ll_inner.sum(axis=-1))
    return a, P, ll


# --------------------------------------------------------------------------- #
# whole filter
# --------------------------------------------------------------------------- #
def _split_time_varying(x, name, time_varying, core_ndim):
    """Time-varying matrices carry a time axis in front of the core axes (reference: a 3-D
    matrix = time on axis 0, filters/utilities.py:1-20); with a draw axis it is (B,n,...).
    Returns an array shaped (B|1, n|1, ...core)."""
    x = np.asarray(x, dtype=np.float64)
    if name in time_varying:
        if x.ndim == core_ndim + 1:
            return x[None]
        if x.ndim == core_ndim + 2:
            return x
        raise ValueError(f"time-varying {name}: bad ndim {x.ndim}")
    if x.ndim == core_ndim:
        return x[None, None]
    if x.ndim == core_ndim + 1:
        return x[:, None]
    raise ValueError(f"{name}: bad ndim {x.ndim}")


def kalman_filter_time_varying(data, a0, P0, T, Z, R, H, Q, c=None, d=None, time_varying=(),
                               filter_type="standard", mvn_const="reference"):
    """BaseFilter.build_graph with sequences (kalman_filter.py:126-163, unpack_args :92-124):
    step t uses slice t of every time-varying matrix for its update AND its predict.
    Whole-row missing data only.  Same outputs as ``kalman_filter``."""
    tv = set(time_varying)
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3 and data.shape[-1] == 1:
        data = data[..., 0]
    P0 = np.asarray(P0, dtype=np.float64)
    batched = P0.ndim == 3
    P0b = P0 if batched else P0[None]
    a0 = np.asarray(a0, dtype=np.float64)
    if a0.ndim >= 2 and a0.shape[-1] == 1 and a0.shape[-2] == P0b.shape[-1]:
        a0 = a0[..., 0]
    a0b = a0 if a0.ndim == 2 else a0[None]
    B, m = P0b.shape[0], P0b.shape[1]
    n = data.shape[-2]
    Tm = _split_time_varying(T, "T", tv, 2)
    Zm = _split_time_varying(Z, "Z", tv, 2)
    Rm = _split_time_varying(R, "R", tv, 2)
    Hm = _split_time_varying(H, "H", tv, 2)
    Qm = _split_time_varying(Q, "Q", tv, 2)
    p = Zm.shape[-2]
    def _vec(x, name, dim):
        """(.., dim, 1) columns -> (.., dim); a batched time-varying vector given as (B, n, dim) is taken as it is (at dim == 1 its
        trailing axis is not a column axis: round-4 fuzz, n = m = 1)."""
        x = np.asarray(x, dtype=np.float64)
        if name in tv and batched and x.shape == (B, n, dim):
            return x
        if x.shape[-1] == 1 and x.ndim > 1 and x.shape[-2] == dim:
            return x.reshape(x.shape[:-1])
        return x

    cm = np.zeros((1, 1, m)) if c is None else _split_time_varying(_vec(c, "c", m), "c", tv, 1)
    dm = np.zeros((1, 1, p)) if d is None else _split_time_varying(_vec(d, "d", p), "d", tv, 1)

    def at(x, t):
        xt = x[:, t if x.shape[1] > 1 else 0]
        return np.broadcast_to(xt, (B,) + xt.shape[1:])

    a = np.broadcast_to(a0b, (B, m)).copy()
    P = np.broadcast_to(P0b, (B, m, m)).copy()
    fa = np.empty((B, n, m)); pa = np.empty((B, n + 1, m))
    fP = np.empty((B, n, m, m)); pP = np.empty((B, n + 1, m, m))
    ll_obs = np.empty((B, n))
    pa[:, 0], pP[:, 0] = a, P
    for t in range(n):
        y = np.broadcast_to(data[t] if data.ndim == 2 else data[:, t], (B, p))
        Tt, Zt, Rt, Ht, Qt, ct, dt = (at(x, t) for x in (Tm, Zm, Rm, Hm, Qm, cm, dm))
        y_m, Z_m, H_m, all_nan, nan_mask = handle_missing_values(y, Zt, Ht, "delete")
        if filter_type == "standard":
            a_f, P_f, ll = update_standard(a, P, y_m, dt, Z_m, H_m, all_nan, nan_mask, mvn_const, "delete")
        elif filter_type == "cholesky":
            a_f, P_f, ll = update_cholesky(a, P, y_m, dt, Z_m, H_m, all_nan, nan_mask, mvn_const,
                                           "delete", "correct")
        elif filter_type == "single":
            a_f, P_f, ll = update_single(a, P, y_m, dt, Z_m, H_m, all_nan, nan_mask, mvn_const)
        else:
            a_f, P_f, ll = update_univariate(a, P, y_m, dt, Z_m, H_m, nan_mask, mvn_const)
        a, P = predict(a_f, P_f, ct, Tt, Rt @ Qt @ _T(Rt))
        fa[:, t], fP[:, t], ll_obs[:, t] = a_f, P_f, ll
        pa[:, t + 1], pP[:, t + 1] = a, P
    return dict(filtered_states=fa, predicted_states=pa, filtered_covs=fP, predicted_covs=pP,
                logp=ll_obs.sum(axis=-1), ll_obs=ll_obs, batched=batched)


def kalman_filter(data, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                  mvn_const="reference", cholesky_triangle="reference",
                  partial_missing="reference"):
    """BaseFilter.build_graph + _postprocess_scan_results, batched.

    Returns dict with
      filtered_states (B,n,m)   predicted_states (B,n+1,m)
      filtered_covs (B,n,m,m)   predicted_covs (B,n+1,m,m)
      logp (B,)                 ll_obs (B,n)
    """
    if filter_type not in FILTER_TYPES:
        raise NotImplementedError(filter_type)
    s = canonicalize(data, a0, P0, T, Z, R, H, Q, c, d)
    B, n, p, m = s["B"], s["n"], s["p"], s["m"]
    Tm, Zm, Rm, Hm, Qm, cv, dv = s["T"], s["Z"], s["R"], s["H"], s["Q"], s["c"], s["d"]
    if filter_type == "single" and p != 1:
        raise AssertionError("UnivariateTimeSeries filter requires data be at most 1-dimensional")

    RQR = Rm @ Qm @ _T(Rm)
    a = np.array(s["a0"])
    P = np.array(s["P0"])

    fa = np.empty((B, n, m))
    pa = np.empty((B, n + 1, m))
    fP = np.empty((B, n, m, m))
    pP = np.empty((B, n + 1, m, m))
    ll_obs = np.empty((B, n))
    pa[:, 0] = a
    pP[:, 0] = P

    for t in range(n):
        y = s["data"][:, t, :]
        y_m, Z_m, H_m, all_nan, nan_mask = handle_missing_values(y, Zm, Hm, partial_missing)
        if filter_type == "standard":
            a_f, P_f, ll = update_standard(a, P, y_m, dv, Z_m, H_m, all_nan, nan_mask,
                                           mvn_const, partial_missing)
        elif filter_type == "cholesky":
            a_f, P_f, ll = update_cholesky(a, P, y_m, dv, Z_m, H_m, all_nan, nan_mask,
                                           mvn_const, partial_missing, cholesky_triangle)
        elif filter_type == "single":
            a_f, P_f, ll = update_single(a, P, y_m, dv, Z_m, H_m, all_nan, nan_mask, mvn_const)
        else:
            a_f, P_f, ll = update_univariate(a, P, y_m, dv, Z_m, H_m, nan_mask, mvn_const)
        a, P = predict(a_f, P_f, cv, Tm, RQR)
        fa[:, t], fP[:, t], ll_obs[:, t] = a_f, P_f, ll
        pa[:, t + 1], pP[:, t + 1] = a, P

    return dict(filtered_states=fa, predicted_states=pa, filtered_covs=fP, predicted_covs=pP,
                logp=ll_obs.sum(axis=-1), ll_obs=ll_obs, batched=s["batched"])


def kalman_smoother(T, R, Q, filtered_states, filtered_covs, time_varying=(), pairing="filter"):
    """KalmanSmoother.build_graph, kalman_smoother.py:56-104, batched.

    filtered_states (B,n,m) | (n,m) | (n,m,1); filtered_covs (B,n,m,m) | (n,m,m).
    predict here has NO intercept and NO symmetrisation (:99-104); the gain uses
    ``pinv`` (numpy default rcond=1e-15, SVD based), :92.

    ``time_varying`` (subset of T, R, Q): that matrix is (B,n,.,.) and the step that smooths t predicts from t with
    slice t -- the pairing of the forward filter (kalman_filter.py:231-252).  QUIRK, not reproduced: the reference's
    own scan (kalman_smoother.py:62-77) hands the n-long matrix sequences reversed next to the (n-1)-long reversed
    filtered tapes, so its step for t receives slice t + 1; with that pairing the smoother is not the adjoint of the
    filter that produced the tapes (and nothing in the reference's tests exercises it numerically: they are shape
    tests, tests/test_kalman_filter.py:143-156).  ``pairing="reference"`` reproduces the reference's pairing (slice t + 1 at the
    step that smooths t), so that parity against upstream can be shown for a model ported with time-varying T / R / Q;
    the default ``"filter"`` is the pairing described above.
    """
    if pairing not in ("filter", "reference"):
        raise ValueError("pairing must be 'filter' or 'reference'")
    tv = set(time_varying)
    if tv:
        return _kalman_smoother_tv(T, R, Q, filtered_states, filtered_covs, tv, 1 if pairing == "reference" else 0)
    T_b, _ = _as_batch(T, 2)
    R_b, _ = _as_batch(R, 2)
    Q_b, _ = _as_batch(Q, 2)
    fa = np.asarray(filtered_states, dtype=np.float64)
    fP = np.asarray(filtered_covs, dtype=np.float64)
    m = T_b.shape[-1]
    if fa.shape[-1] == 1 and fa.ndim >= 2 and fa.shape[-2] == m:
        fa = fa[..., 0]
    batched = fP.ndim == 4
    if not batched:
        fa, fP = fa[None], fP[None]
    B, n = fa.shape[0], fa.shape[1]
    T_b = np.broadcast_to(T_b, (B, m, m))
    RQR = np.broadcast_to(R_b @ Q_b @ _T(R_b), (B, m, m))

    sa = np.empty_like(fa)
    sP = np.empty_like(fP)
    sa[:, -1], sP[:, -1] = fa[:, -1], fP[:, -1]
    a_s, P_s = fa[:, -1], fP[:, -1]
    for t in range(n - 2, -1, -1):
        a, P = fa[:, t], fP[:, t]
        a_hat = np.einsum("bij,bj->bi", T_b, a)
        P_hat = T_b @ P @ _T(T_b) + RQR
        gain = _T(np.linalg.pinv(P_hat) @ T_b @ P)
        a_s = a + np.einsum("bij,bj->bi", gain, a_s - a_hat)
        P_s = P + gain @ (P_s - P_hat) @ _T(gain)
        sa[:, t], sP[:, t] = a_s, P_s
    if not batched:
        return sa[0], sP[0]
    return sa, sP


def _kalman_smoother_tv(T, R, Q, filtered_states, filtered_covs, tv, shift=0):
    fa = np.asarray(filtered_states, dtype=np.float64)
    fP = np.asarray(filtered_covs, dtype=np.float64)
    B, n, m = fa.shape
    T, R, Q = (np.asarray(x, dtype=np.float64) for x in (T, R, Q))

    def at(x, name, t):
        return x[:, t] if name in tv else x

    sa, sP = np.empty_like(fa), np.empty_like(fP)
    sa[:, -1], sP[:, -1] = fa[:, -1], fP[:, -1]
    a_s, P_s = fa[:, -1], fP[:, -1]
    for t in range(n - 2, -1, -1):
        Tt, Rt, Qt = at(T, "T", t + shift), at(R, "R", t + shift), at(Q, "Q", t + shift)
        a, P = fa[:, t], fP[:, t]
        a_hat = np.einsum("bij,bj->bi", Tt, a)
        P_hat = Tt @ P @ _T(Tt) + Rt @ Qt @ _T(Rt)
        gain = _T(np.linalg.pinv(P_hat) @ Tt @ P)
        a_s = a + np.einsum("bij,bj->bi", gain, a_s - a_hat)
        P_s = P + gain @ (P_s - P_hat) @ _T(gain)
        sa[:, t], sP[:, t] = a_s, P_s
    return sa, sP


def smoother_from_filter(data, T, Z, H, predicted_states, predicted_covs, d=None, filter_type="standard"):
    """The smoothed moments from the FILTER's predicted tapes, by the backward recursion of de Jong (1989) / Durbin & Koopman
    (2012, section 4.4) -- the recursion statsmodels itself runs, i.e. the external oracle of the reference's only numeric test
    (tests/test_kalman_filter.py:226-241); what ``kf_smoother_from_filter`` (csrc/kf_smoother_bf.hip) computes on the GPU:

        v_t = y_t - Z a_t - d      F_t = Z P_t Z' + H      K_t = P_t Z' F_t^-1      L_t = T (I - K_t Z)
        r_{t-1} = Z' F_t^-1 v_t + L_t' r_t       N_{t-1} = Z' F_t^-1 Z + L_t' N_t L_t       (r_{n-1} = 0, N_{n-1} = 0)
        a_s[t] = a_t + P_t r_{t-1}               P_s[t] = P_t - P_t N_{t-1} P_t

    with (a_t, P_t) = predicted_states[:, t], predicted_covs[:, t] (the (B, n+1, ..) tapes of ``kalman_filter``; slot n unused).
    Missing entries of y_t are deleted from that step (rows of Z, rows / columns of H: the semantics statsmodels pins); a wholly
    missing step has L_t = T and no Z terms.  Same E / Var of the state given all observations as ``kalman_smoother`` (the
    reference's RTS form, kalman_smoother.py:56-104), without pinv(P_hat): the two agree to rounding where P_hat is well
    conditioned, and this form stays bounded where the RTS recursion does not (a numerically singular P_hat: rank-deficient state
    noise behind a long series -- scripts/r05/fuzz_r05.py found draws whose RTS covariances reach 1e22 .. 1e63 in float64).
    Pinned against every statsmodels golden vector in tests/test_oracle_golden.py.  Any p; ``filter_type="single"`` flips the
    sign of d (kalman_filter.py:335).  Returns (smoothed_states (B,n,m), smoothed_covs (B,n,m,m))."""
    y = np.asarray(data, dtype=np.float64)
    T_b, _ = _as_batch(T, 2)
    Z_b, _ = _as_batch(Z, 2)
    H_b, _ = _as_batch(H, 2)
    pa = np.asarray(predicted_states, dtype=np.float64)
    pP = np.asarray(predicted_covs, dtype=np.float64)
    m, p = T_b.shape[-1], Z_b.shape[-2]
    if pa.shape[-1] == 1 and pa.ndim >= 2 and pa.shape[-2] == m:
        pa = pa[..., 0]
    if pP.ndim == 3:
        pa, pP = pa[None], pP[None]
    B = pP.shape[0]
    n = pP.shape[1] - 1
    T_b = np.broadcast_to(T_b, (B, m, m))
    Z_b = np.broadcast_to(Z_b, (B, p, m))
    H_b = np.broadcast_to(H_b, (B, p, p))
    dv = np.zeros((B, p)) if d is None else np.broadcast_to(np.asarray(d, dtype=np.float64).reshape(-1, p), (B, p))
    dsign = 1.0 if filter_type == "single" else -1.0
    per_draw = y.ndim == 3
    sa = np.empty((B, n, m))
    sP = np.empty((B, n, m, m))
    for b in range(B):
        Tm, r, N = T_b[b], np.zeros(m), np.zeros((m, m))
        for t in range(n - 1, -1, -1):
            yt = (y[b, t] if per_draw else y[t]).reshape(p)
            ob = ~np.isnan(yt)
            a, P = pa[b, t], pP[b, t]
            if ob.any():
                Zo = Z_b[b][ob]
                F = Zo @ P @ Zo.T + H_b[b][np.ix_(ob, ob)]
                Fi = np.linalg.inv(F)
                v = yt[ob] - Zo @ a + dsign * dv[b][ob]
                K = P @ Zo.T @ Fi
                L = Tm @ (np.eye(m) - K @ Zo)
                r = Zo.T @ (Fi @ v) + L.T @ r
                N = Zo.T @ Fi @ Zo + L.T @ N @ L
            else:
                r = Tm.T @ r
                N = Tm.T @ N @ Tm
            sa[b, t] = a + P @ r
            sP[b, t] = P - P @ N @ P
    return sa, sP


def reference_outputs(data, a0, P0, T, Z, R, H, Q, filter_type="standard", **kw):
    """Single-draw convenience in the reference test order and shapes
    (tests/utilities/test_helpers.py:40-49, :79-90):
    [filtered_states (n,m,1), predicted_states (n+1,m,1), smoothed_states (n,m,1),
     filtered_covs, predicted_covs, smoothed_covs, log_likelihood (), ll_obs (n,)]."""
    out = kalman_filter(data, a0, P0, T, Z, R, H, Q, filter_type=filter_type, **kw)
    sa, sP = kalman_smoother(T, R, Q, out["filtered_states"][0], out["filtered_covs"][0])
    return [
        out["filtered_states"][0][..., None],
        out["predicted_states"][0][..., None],
        sa[..., None],
        out["filtered_covs"][0],
        out["predicted_covs"][0],
        sP,
        out["logp"][0],
        out["ll_obs"][0],
    ]
