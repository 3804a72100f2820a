"""Numpy prototype of the time-invariant composition: for a static model without missing observations the
covariance sequence does not depend on y, and the s-step map of the FILTERED covariance is ONE element (A, C, J) of the associative
operator of Sarkka & Garcia-Fernandez (2021) raised to the s-th power -- log2(s) squarings (the structure-preserving doubling of the
Riccati literature).  Checks, per workload: chunk-start covariances from the doubling against the sequential recursion, and the mean
transition Psi_j = A_s (I + P_f J_s)^-1 of a chunk against the product of the per-step closed-loop matrices."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402


def element(T, Z, RQR, H):
    z = Z[0]
    S0 = z @ RQR @ z + H
    K0 = RQR @ z / S0
    IKZ = np.eye(len(z)) - np.outer(K0, z)
    zt = z @ T
    return IKZ @ T, IKZ @ RQR, np.outer(zt, zt) / S0, S0


def compose(e1, e2):   # e1 first, then e2
    A1, C1, J1 = e1
    A2, C2, J2 = e2
    m = A1.shape[0]
    D = np.linalg.inv(np.eye(m) + C1 @ J2)
    A = A2 @ D @ A1
    C = A2 @ D @ C1 @ A2.T + C2
    J = A1.T @ J2 @ D @ A1 + J1     # (I + J2 C1)^-1 J2 = J2 (I + C1 J2)^-1
    return A, 0.5 * (C + C.T), 0.5 * (J + J.T)


def apply(e, Pf):
    A, C, J = e
    m = A.shape[0]
    D = np.linalg.inv(np.eye(m) + Pf @ J)
    Psi = A @ D
    Pn = Psi @ Pf @ A.T + C
    return 0.5 * (Pn + Pn.T), Psi


def seq_filter(P0, T, Z, RQR, H, n):
    z = Z[0]
    P = P0.copy()
    preds, filts, Ls = [P.copy()], [], []
    for t in range(n):
        Mv = P @ z
        F = z @ Mv + H
        K = Mv / F
        Pf = P - np.outer(K, Mv)
        Pf = 0.5 * (Pf + Pf.T)
        filts.append(Pf)
        Ls.append((np.eye(len(z)) - np.outer(K, z)))
        P = T @ Pf @ T.T + RQR
        P = 0.5 * (P + P.T)
        preds.append(P.copy())
    return preds, filts, Ls


def run(name, w, s=128, draws=8):
    n = w["y"].shape[0]
    worst = 0.0
    worst_psi = 0.0
    for b in range(min(draws, w["T"].shape[0])):
        T, Z, R, Q, H, P0 = w["T"][b], w["Z"][b], w["R"][b], w["Q"][b], w["H"][b][0, 0], w["P0"][b]
        RQR = R @ Q @ R.T
        preds, filts, Ls = seq_filter(P0, T, Z, RQR, H, n)
        e1 = element(T, Z, RQR, H)
        if not e1[3] > 0:
            print(name, b, "S0 <= 0: not eligible")
            continue
        e = e1[:3]
        k = 1
        while k < s:
            e = compose(e, e)
            k *= 2
        assert k == s
        Pf = filts[0]
        j = 0
        while (j + 1) * s < n:
            Pf_next, Psi = apply(e, Pf)
            t_next = (j + 1) * s   # filtered covariance of step t_next
            ref = filts[t_next]
            err = np.abs(Pf_next - ref).max() / max(1e-300, np.abs(ref).max())
            worst = max(worst, err)
            # mean transition: a_f(t_next) = Psi a_f(j s) + ...;  per step a_f(t+1) = (I - K_{t+1} Z) T a_f(t) + ...
            prod = np.eye(T.shape[0])
            for t in range(j * s + 1, t_next + 1):
                prod = Ls[t] @ T @ prod
            epsi = np.abs(Psi - prod).max() / max(1e-300, np.abs(prod).max(), 1e-30)
            worst_psi = max(worst_psi, np.abs(Psi - prod).max())
            Pf = Pf_next
            j += 1
    print(f"{name:22s} s={s:4d}  worst rel. error of a chunk-start filtered covariance {worst:.2e}   worst abs. error of Psi {worst_psi:.2e}")


if __name__ == "__main__":
    run("headline structural", W.structural(B=8, n=1000))
    run("structural s=125", W.structural(B=8, n=1000), s=128)
    run("arma11 (H = 0)", W.arma11(B=8, n=1000))
    run("dense m=6", W.random_dense(8, 1000, 1, 6, 2, seed=3))
    run("dense m=10 r=1", W.random_dense(8, 1000, 1, 10, 1, seed=4))
    wn = W.nile_local_linear_trend(B=4)
    wn["y"] = np.tile(wn["y"], (6, 1))
    run("nile LLT P0=1e6 (x6)", wn, s=128)
    run("local level", W.nile_local_level(B=4) | {"y": np.tile(W.nile_local_level(B=1)["y"], (6, 1))}, s=128)
