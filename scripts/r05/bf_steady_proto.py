"""Development aid (CPU): the steady phase of kf_smoother_bf.hip.  Once the predicted covariance on the tape equals a reference P_bar
(within 1e-14 of its largest entry) AND the state block of M_t has stopped moving (1e-15 of its largest entry, two checks four steps
apart), a step is two vector recursions with constant matrices:
    v_t = y_t - Z a_t - d        a_s[t] = a_t + K v_t + P_f m_t        m_{t-1} = G_ss' m_t + (ZT)' v_t / F        P_s[t] = P_f - P_f M_bar P_f
(m_t = M_t[:, JC]); how many steps qualify on the BASELINE workloads, and the deviation from every step in full."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "scripts/r05")
import workloads as W
from oracle import kalman_numpy as kn
from bf_proto2 import bf_m


def bf_steady(y, T, Z, H, d, pred_a, pred_P, tolP=1e-14, tolM=1e-15):
    n, m = y.shape[0], T.shape[0]
    Ta = np.zeros((m + 1, m + 1)); Ta[:m, :m] = T; Ta[m, m] = 1.0
    ZT = np.zeros(m + 1); ZT[:m] = Z @ T
    M = np.zeros((m + 1, m + 1))
    sa, sP = np.empty((n, m)), np.empty((n, m, m))
    nsteady = 0
    t = n - 1
    Mprev, passes = None, 0
    while t >= 0:
        P, a = pred_P[t], pred_a[t]
        zP = Z @ P
        F = H + zP @ Z
        ob = (not np.isnan(y[t])) and F > 0
        rF = 1.0 / F if ob else 0.0
        v = (y[t] - Z @ a - d) if ob else 0.0
        kap = np.zeros(m + 1); kap[:m] = zP * rF; kap[m] = -v * rF
        Pe = np.zeros((m + 1, m + 1)); Pe[:m, :m] = P; Pe[:m, m] = a; Pe[m, :m] = a
        Pf = Pe - F * np.outer(kap, kap) if ob else Pe
        Pfb = Pf.copy(); Pfb[m, :] = 0.0
        Y = M @ Pfb
        Pl = Pf.copy(); Pl[:, m] = 0.0
        S = Pf - Pl @ Y
        sP[t] = S[:m, :m]; sa[t] = Pf[m, :m] + Y[m, :m]
        K = kap.copy(); K[m] = 0.0
        G = Ta - np.outer(K, ZT)
        zeta = ZT.copy(); zeta[m] = v
        Mn = G.T @ M @ G + rF * np.outer(zeta, zeta)
        # convergence test every 4th step
        enter = False
        if t >= 2 and (t & 3) == 0 and ob:
            sameP = np.abs(pred_P[t - 1] - P).max() <= tolP * np.abs(P).max()
            convM = Mprev is not None and np.abs(Mn[:m, :m] - Mprev).max() <= tolM * np.abs(Mn[:m, :m]).max()
            passes = passes + 1 if (sameP and convM) else 0
            Mprev = Mn[:m, :m].copy()
            enter = passes >= 2
        M = Mn
        t -= 1
        if enter:
            Pbar, Gss, Pfs, Ks = P, G[:m, :m].copy(), Pf[:m, :m].copy(), kap[:m].copy()
            Mbar = M[:m, :m].copy()
            Psc = Pfs - Pfs @ Mbar @ Pfs
            mv = M[:m, m].copy()
            pmax = np.abs(Pbar).max()
            while t >= 0 and not np.isnan(y[t]) and np.abs(pred_P[t] - Pbar).max() <= tolP * pmax:
                a = pred_a[t]
                v = y[t] - Z @ a - d
                sa[t] = a + Ks * v + Pfs @ mv
                sP[t] = Psc
                mv = Gss.T @ mv + ZT[:m] * (v * rF)
                nsteady += 1
                t -= 1
            M = np.zeros((m + 1, m + 1)); M[:m, :m] = Mbar; M[:m, m] = mv; M[m, :m] = mv
            Mprev, passes = None, 0
    return sa, sP, nsteady


if __name__ == "__main__":
    for name, kw, ex in (("headline", dict(B=6), True), ("c3", dict(B=3), False)):
        w = W.WORKLOADS[name](**kw)
        y = w["y"].copy()
        y[[500]] = np.nan
        flt = kn.kalman_filter(y, *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
        for b in range(kw["B"]):
            pa = flt["predicted_states"][b].reshape(-1, w["m"])[:-1]
            args = (y[:, 0], w["T"][b], w["Z"][b, 0], w["H"][b, 0, 0], 0.0, pa, flt["predicted_covs"][b, :-1])
            ra, rP = bf_m(*args)
            sa, sP, ns = bf_steady(*args)
            print(name, b, "steady steps", ns, "of", y.shape[0], "dev a", np.abs(sa - ra).max() / np.abs(ra).max(), "dev P", np.abs(sP - rP).max() / np.abs(rP).max())
