"""Prototype of the STEADY-STATE ADJOINT: gradient of the Kalman log-likelihood when the covariance recursion has converged.

Exact gradient (what PyMC's autodiff of the reference scan returns, core/statespace.py:174): torch autograd of the recursion at
every step.  Hybrid: for t >= ts (covariance converged: P_t = P_inf to rounding) the log-likelihood depends on the parameters
  * directly and through the constant gain K = P Z' F^-1, F = Z P Z' + H: a backward VECTOR recursion
        lam_t = A' lam_{t+1} + gl_t Z' W v_t,   A = T (I - K Z),  W = F^-1
    plus sums of outer products for K_b, W_b, T_b, Z_b, c_b, d_b -- no m^3 work per step;
  * through P_inf(theta): ONE Riccati adjoint per draw (SolveDiscreteARE.grad, reference utils/pytensor_scipy.py:33-60);
and for t < ts the exact adjoint runs from (lam_ts, G = 0).  Checked here against the exact gradient."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import workloads as W
from oracle import steady_state_numpy as so

torch.set_default_dtype(torch.float64)


def exact_filter(y, a, P, T, Z, R, H, Q, c, d, gl, t0=0, t1=None, lam_end=None):
    """Joint (Standard) recursion on steps [t0, t1) from predicted (a, P); returns sum gl_t ll_t (+ lam_end . a_t1) and (a_t1, P_t1)."""
    n = y.shape[0] if t1 is None else t1
    RQR = R @ Q @ R.T
    obj = 0.0
    for t in range(t0, n):
        v = y[t] - Z @ a - d
        F = Z @ P @ Z.T + H
        Fi = torch.linalg.inv(F)
        K = P @ Z.T @ Fi
        obj = obj + gl[t] * (-0.5 * (len(v) * np.log(2 * np.pi) + torch.logdet(F) + v @ Fi @ v))
        af = a + K @ v
        Pf = P - K @ F @ K.T
        a = T @ af + c
        P = T @ Pf @ T.T + RQR
        P = 0.5 * (P + P.T)
    if lam_end is not None:
        obj = obj + (torch.as_tensor(lam_end) * a).sum()
    return obj, a, P


def run(w, b, ts_rule="conv"):
    names = ("a0", "P0", "T", "Z", "R", "H", "Q")
    vals = {k: np.asarray(w[k][b], dtype=np.float64) for k in names}
    m, p = vals["T"].shape[0], vals["Z"].shape[0]
    vals["a0"] = vals["a0"].reshape(m)
    vals["c"] = 0.01 * np.arange(m) / m
    vals["d"] = 0.02 * np.ones(p)
    y = torch.tensor(w["y"])
    n = y.shape[0]
    rng = np.random.default_rng(b)
    gl = torch.tensor(1.0 + 0.3 * rng.standard_normal(n))
    # ---- exact ------------------------------------------------------------------------------------------------------------
    th = {k: torch.tensor(v, requires_grad=True) for k, v in vals.items()}
    obj, _, _ = exact_filter(y, th["a0"], th["P0"], th["T"], th["Z"], th["R"], th["H"], th["Q"], th["c"], th["d"], gl)
    obj.backward()
    gex = {k: th[k].grad.numpy().copy() for k in th}
    # ---- where has the covariance converged?  (the kernels' test: kf_forward_b4.hip) -----------------------------------------
    with torch.no_grad():
        T_, Z_, R_, H_, Q_ = (torch.tensor(vals[k]) for k in ("T", "Z", "R", "H", "Q"))
        P = torch.tensor(vals["P0"]); a = torch.tensor(vals["a0"])
        P = 0.5 * (P + P.T)
        Ps, As = [P.clone()], [a.clone()]
        for t in range(n):
            _, a, P = exact_filter(y, a, P, T_, Z_, R_, H_, Q_, torch.tensor(vals["c"]), torch.tensor(vals["d"]), gl, t0=t, t1=t + 1)
            Ps.append(P.clone()); As.append(a.clone())
    ts, npass = n, 0
    for t in range(7, n - 16, 8):
        dP = (Ps[t + 1] - Ps[t]).abs()
        dg = torch.sqrt(torch.diag(Ps[t + 1]).abs())
        ok = bool(dP.max() <= 1e-15 * Ps[t + 1].abs().max()) and bool((dP <= 1e-13 * torch.outer(dg, dg)).all())
        npass = npass + 1 if ok else 0
        if npass >= 2:
            ts = t + 1
            break
    # ---- steady part: t in [ts, n) with P = P_ts ------------------------------------------------------------------------------
    P = Ps[ts].numpy(); T, Z, R, H, Q, c, d = (vals[k] for k in ("T", "Z", "R", "H", "Q", "c", "d"))
    F = Z @ P @ Z.T + H
    Wm = np.linalg.inv(F)
    M = P @ Z.T
    K = M @ Wm
    A = T @ (np.eye(m) - K @ Z)
    lam = np.zeros(m)
    Tb = np.zeros((m, m)); Kb = np.zeros((m, p)); Zb = np.zeros((p, m)); cb = np.zeros(m); db = np.zeros(p)
    Wb = np.zeros((p, p))
    yn = y.numpy(); gln = gl.numpy()
    for t in range(n - 1, ts - 1, -1):
        at = As[t].numpy()
        v = yn[t] - Z @ at - d
        wv = Wm @ v
        mu = T.T @ lam                       # adjoint of a_f
        Tb += np.outer(lam, at + K @ v)
        cb += lam
        Kb += np.outer(mu, v)
        vb = -gln[t] * wv + K.T @ mu
        Zb += -np.outer(vb, at)
        db += -vb
        Wb += gln[t] * 0.5 * (F - np.outer(v, v))
        lam = mu - Z.T @ vb
    # K = M W, M = P Z', W = F^-1, F = Z P Z' + H
    Mb = Kb @ Wm
    Wb = Wb + M.T @ Kb
    Wb = 0.5 * (Wb + Wb.T)
    Fb = -Wm @ Wb @ Wm
    Pb = Mb @ Z + Z.T @ Fb @ Z
    Zb = Zb + Mb.T @ P + 2.0 * Fb @ Z @ P
    Hb = Fb
    g = dict(T=Tb, Z=Zb, H=Hb, c=cb, d=db, R=np.zeros_like(R), Q=np.zeros_like(Q), a0=np.zeros(m), P0=np.zeros((m, m)))
    # ---- P_inf(theta): one Riccati adjoint ------------------------------------------------------------------------------------------
    if ts < n:
        for k, v in so.steady_state_grad(T, Z, R, H, Q, Pb).items():
            g[k] = g[k] + v
    # ---- transient part: exact adjoint of steps [0, ts) with the mean's cotangent at ts, none for the covariance ------------------------
    th = {k: torch.tensor(v, requires_grad=True) for k, v in vals.items()}
    obj, _, _ = exact_filter(y, th["a0"], th["P0"], th["T"], th["Z"], th["R"], th["H"], th["Q"], th["c"], th["d"], gl, t0=0, t1=ts,
                             lam_end=lam)
    obj.backward()
    for k in th:
        g[k] = g[k] + th[k].grad.numpy()
    worst = 0.0
    for k in gex:
        sc = max(np.abs(gex[k]).max(), 1e-300)
        err = np.abs(g[k] - gex[k]).max() / sc
        worst = max(worst, err)
    return ts, worst, {k: np.abs(g[k] - gex[k]).max() / max(np.abs(gex[k]).max(), 1e-300) for k in gex}


if __name__ == "__main__":
    for name, B, n in (("c4", 6, 400), ("headline", 6, 900), ("c2", 4, 400)):
        w = W.WORKLOADS[name](B=B, n=n)
        for b in range(B):
            ts, worst, per = run(w, b)
            print(f"{name} draw {b}: ts = {ts}/{n}  max rel gradient error {worst:.2e}  " +
                  " ".join(f"{k}:{v:.0e}" for k, v in per.items()), flush=True)


def debug_H():
    w = W.WORKLOADS["c4"](B=1, n=200)
    import types
    global _dbg
    _dbg = True
