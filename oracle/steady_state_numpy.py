"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's SteadyStateFilter and
simulate_statespace.  Never imported by the product path.

Follows, line by line:
  * SteadyStateFilter.build_graph / update / kalman_step
    (reference pymc_statespace/filters/kalman_filter.py:354-441): ``P_steady =
    solve_discrete_are(T.T, Z.T, R Q R', H)`` (:388, the SciPy routine of
    utils/pytensor_scipy.py:11-60), ``F_inv = (Z P_steady Z' + H)^-1`` frozen (:389-390), the
    scan starts from ``P_steady`` (:395) while ``_postprocess_scan_results`` prepends the
    caller's ``P0`` (:397); ``v = y - Z a`` without ``d`` (:400); Joseph form (:409); ``det(F)``
    of the CURRENT F in the likelihood (:415).
  * simulate_statespace (reference pymc_statespace/utils/simulation.py:29-64) with the normals
    passed in instead of ``np.random.randn``.

Parity status: unpinned (no reference test pins the steady-state filter's numbers --
tests/test_kalman_filter.py only shape-checks it -- and the simulation is random); the two
restatements are checked against SciPy's ``solve_discrete_are`` and against a direct
simulation in tests/test_steady_state.py.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_discrete_are

LOG_2PI = float(np.log(2.0 * np.pi))


def steady_state_filter(y, a0, P0, T, Z, R, H, Q, c=None):
    """One draw.  y (n,p) with NaN for missing; returns the reference's six outputs as a dict."""
    y = np.asarray(y, dtype=np.float64)
    n, p = y.shape
    m = T.shape[0]
    c = np.zeros(m) if c is None else np.asarray(c, dtype=np.float64).reshape(m)
    RQR = R @ Q @ R.T
    P_steady = solve_discrete_are(T.T, Z.T, RQR, H)
    F_inv = np.linalg.solve(Z @ P_steady @ Z.T + H, np.eye(p))
    a, P = np.asarray(a0, dtype=np.float64).reshape(m), P_steady
    fa, pa, fP, pP, ll = [], [a.copy()], [], [np.asarray(P0, dtype=np.float64)], []
    eye = np.eye(m)
    for t in range(n):
        nan = np.isnan(y[t])
        all_nan = bool(nan.all())
        W = np.diag((~nan).astype(np.float64))
        Zm, Hm, ym = W @ Z, W @ H, np.where(nan, 0.0, y[t])
        v = ym - Zm @ a
        PZT = P @ Zm.T
        F = Zm @ PZT + Hm
        K = PZT @ F_inv
        IKZ = eye - K @ Zm
        af = a + K @ v
        Pf = IKZ @ P @ IKZ.T + K @ Hm @ K.T
        ll.append(0.0 if all_nan else -0.5 * (LOG_2PI + np.log(np.linalg.det(F)) + v @ F_inv @ v))
        a = T @ af + c
        P = T @ Pf @ T.T + RQR
        P = 0.5 * (P + P.T)
        fa.append(af); fP.append(Pf); pa.append(a.copy()); pP.append(P.copy())
    return dict(filtered_states=np.array(fa), predicted_states=np.array(pa), filtered_covs=np.array(fP),
                predicted_covs=np.array(pP), logp=float(np.sum(ll)), ll_obs=np.array(ll), P_steady=P_steady)


def simulate_statespace(T, Z, R, H, Q, state_noise, obs_noise, x0=None):
    """utils/simulation.py:29-64 with the standard normals given: state_noise (n,r), obs_noise (n,p)."""
    n_steps = state_noise.shape[0]
    n_obs, n_states = Z.shape
    k_obs_noise = H.shape[0] * (1 - int(np.all(H == 0)))
    state_innovations = state_noise @ np.linalg.cholesky(Q)
    if k_obs_noise != 0:
        obs_innovations = obs_noise @ np.linalg.cholesky(H)
    states = np.zeros((n_steps, n_states))
    obs = np.zeros((n_steps, n_obs))
    if x0 is not None:
        states[0] = x0
        obs[0] = Z @ x0
    for t in range(1, n_steps):
        states[t] = T @ states[t - 1] + R @ state_innovations[t]
        obs[t] = Z @ states[t - 1] + (obs_innovations[t] if k_obs_noise != 0 else 0.0)
    return states, obs


def dare_grad(A, B, Q, R, dX):
    """Reverse-mode gradient of X = solve_discrete_are(A, B, Q, R): the reference's SolveDiscreteARE.grad
    (utils/pytensor_scipy.py:33-60, Kao & Hennequin 2020), line by line.  Returns (A_bar, B_bar, Q_bar, R_bar)."""
    from scipy.linalg import solve_discrete_lyapunov

    X = solve_discrete_are(A, B, Q, R)
    K_inner = R + B.T @ X @ B                                   # :41
    K_inner_inv = np.linalg.solve(K_inner, np.eye(R.shape[0]))  # :42
    K = K_inner_inv @ B.T @ X @ A                               # :43
    A_tilde = A - B @ K                                         # :45
    dX_symm = 0.5 * (dX + dX.T)                                 # :47
    S = solve_discrete_lyapunov(A_tilde, dX_symm)               # :48   S = A~ S A~' + dX_symm
    A_bar = 2.0 * X @ A_tilde @ S                               # :50
    B_bar = -2.0 * X @ A_tilde @ S @ K.T                        # :51
    Q_bar = S                                                   # :52
    R_bar = K @ S @ K.T                                         # :53
    return A_bar, B_bar, Q_bar, R_bar


def steady_state_grad(T, Z, R, H, Q, g_P):
    """Gradient of <g_P, P_steady> with P_steady = solve_discrete_are(T', Z', R Q R', H) (kalman_filter.py:388) w.r.t.
    the filter's own matrices, every entry independent: the chain rule through A = T', B = Z', Q_d = R Q R', R_d = H."""
    A_bar, B_bar, Q_bar, R_bar = dare_grad(T.T, Z.T, R @ Q @ R.T, H, g_P)
    return dict(T=A_bar.T, Z=B_bar.T, H=R_bar, R=Q_bar @ R @ Q.T + Q_bar.T @ R @ Q, Q=R.T @ Q_bar @ R)


def steady_state_logp_and_grad(y, a0, P0, T, Z, R, H, Q, c=None):
    """(logp, gradients) of the SteadyStateFilter's log-likelihood, every matrix entry independent: torch autograd of
    the Standard recursion started from P_steady (oracle/kalman_torch.py; complete data: the same function as the
    frozen-F_inv recursion above) plus the Riccati adjoint of its g_P0 (steady_state_grad)."""
    from . import kalman_torch as kt

    y = np.asarray(y, dtype=np.float64)
    m = T.shape[0]
    Ps = solve_discrete_are(T.T, Z.T, R @ Q @ R.T, H)
    cc = np.zeros(m) if c is None else np.asarray(c, dtype=np.float64).reshape(m)
    lp, _, g = kt.logp_and_grads(y, np.asarray(a0, dtype=np.float64).reshape(1, m), Ps[None], T[None], Z[None], R[None],
                                 H[None], Q[None], c=cc[None], g_logp=np.ones(1))
    out = {k: g[k][0] for k in ("a0", "T", "Z", "R", "H", "Q", "c")}
    for k, v in steady_state_grad(T, Z, R, H, Q, g["P0"][0]).items():
        out[k] = out[k] + v
    return float(lp[0]), out
