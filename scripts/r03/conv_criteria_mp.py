"""Entrywise convergence test (|dP_ij| <= tol sqrt(P_ii P_jj) on two consecutive checks) on the C2 (ARMA(1,1), check every 16
steps, tol 3e-16) and C4 (VARMAX, check every 8 steps, tol 1e-15) workloads: when does it fire, next to the round-2 test
(max |dP| <= tol max |P|, one check)?"""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W


def riccati(P, T, Z, H, RQR):
    PZ = P @ Z.transpose(0, 2, 1)
    F = Z @ PZ + H
    Pf = P - PZ @ np.linalg.solve(F, PZ.transpose(0, 2, 1))
    S = T @ Pf @ T.transpose(0, 2, 1) + RQR
    return 0.5 * (S + S.transpose(0, 2, 1))


def run(name, B, n, every, tol):
    w = W.WORKLOADS[name](B=B, n=n)
    T, Z, H, R, Q = w["T"], w["Z"], w["H"], w["R"], w["Q"]
    RQR = R @ Q @ R.transpose(0, 2, 1)
    P = 0.5 * (w["P0"] + w["P0"].transpose(0, 2, 1))
    ca = np.full(B, n); cb = np.full(B, n); cb1 = np.full(B, n); passed = np.zeros(B, bool)
    for t in range(n):
        with np.errstate(all="ignore"):
            Pn = riccati(P, T, Z, H + 1e-300 * np.eye(H.shape[-1]), RQR)
        if (t % every) == every - 1 and t + 2 * every < n:
            dP = np.abs(Pn - P)
            dg = np.sqrt(np.abs(np.einsum("bii->bi", Pn)))
            sc = dg[:, :, None] * dg[:, None, :]
            oka = dP.max((1, 2)) <= tol * np.abs(Pn).max((1, 2))
            ca[oka & (ca == n)] = t + 1
            okb = (dP <= tol * sc).all((1, 2))
            cb1[okb & (cb1 == n)] = t + 1
            cb[okb & passed & (cb == n)] = t + 1
            passed = okb
        P = Pn
    f = lambda v: f"median {np.median(v):.0f} p99 {np.percentile(v, 99):.0f} max {v.max()} never {(v == n).sum()}"
    print(f"{name} (check every {every}, tol {tol:g}): round-2 test {f(ca)} | entrywise, one check {f(cb1)} | entrywise, two checks {f(cb)}")


run("c2", 4096, 1000, 16, 3e-16)
run("c2", 4096, 1000, 16, 1e-15)
run("c4", 2048, 1000, 8, 1e-15)
