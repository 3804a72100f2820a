"""Development aid (CPU): RTS smoother with the inverse of P_hat by warm-started Newton-Schulz iterations
X <- X (2 I - A X): how many iterations per step, and how close to the pinv-based oracle?"""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn

def run(w, b, tol=1e-13, maxit=60, verbose=False):
    T, R, Q = w["T"][b], w["R"][b], w["Q"][b]
    m = T.shape[0]
    flt = kn.kalman_filter(w["y"], *[w[k][b:b + 1] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
    fa, fP = flt["filtered_states"][0], flt["filtered_covs"][0]
    n = fa.shape[0]
    RQR = R @ Q @ R.T
    sa, sP = fa.copy(), fP.copy()
    X = None
    its = []
    I = np.eye(m)
    for t in range(n - 2, -1, -1):
        A = T @ fP[t] @ T.T + RQR
        k = 0
        if X is not None:
            Rm = I - A @ X
            if np.abs(Rm).sum(1).max() >= 0.5:
                X = None
        if X is None:
            X = A.T / (np.abs(A).sum(0).max() * np.abs(A).sum(1).max())
        while k < maxit:
            Rm = I - A @ X
            if np.abs(Rm).max() < tol:
                break
            X = X @ (I + Rm)
            k += 1
        its.append(k)
        J = fP[t] @ T.T @ X
        sa[t] = fa[t] + J @ (sa[t + 1] - T @ fa[t])
        sP[t] = fP[t] + J @ (sP[t + 1] - A) @ J.T
    ra, rP = kn.kalman_smoother(w["T"][b:b + 1], w["R"][b:b + 1], w["Q"][b:b + 1], fa[None], fP[None])
    return np.array(its), np.abs(sa - ra[0]).max() / (1 + np.abs(ra).max()), np.abs(sP - rP[0]).max() / (1 + np.abs(rP).max())

for name, kw in (("headline", dict(B=3)), ("c3", dict(B=2)), ("c2", dict(B=3))):
    w = W.WORKLOADS[name](**kw)
    for b in range(kw["B"]):
        its, ea, eP = run(w, b)
        print(f"{name} draw {b}: iterations mean {its.mean():.2f} median {np.median(its):.0f} max {its.max()} first {its[0]} "
              f"last5 {its[-5:]}  hist {np.bincount(its)[:8]}  err a {ea:.1e} P {eP:.1e}")
w = W.random_dense(3, 200, 1, 10, 10, seed=1)
for b in range(3):
    its, ea, eP = run(w, b)
    print(f"dense draw {b}: iterations mean {its.mean():.2f} max {its.max()} first {its[0]} hist {np.bincount(its)[:8]} err a {ea:.1e} P {eP:.1e}")
