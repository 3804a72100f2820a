"""Development aid (CPU): the M-form of kf_smoother_bf.hip -- carry M_t = T_aug' N_t T_aug, fold the gain into the static factors:
    kappa = (K, -v/F)      Pf_ext = P_ext - F kappa kappa'   (state block: P_f; column JC: a_f)
    G = T_aug - K_aug (Z T)_aug        C~ = zeta zeta' / F,  zeta = ((Z T)', v)
    output of step t:  Y = M_t Pf_ext (row JC zeroed),  P_s = P_f - P_f Y,  a_s = a_f + row JC of Y
    recursion:         M_{t-1} = G' M_t G + C~
against the RTS oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn


def bf_m(y, T, Z, H, d, pred_a, pred_P):
    n, m = y.shape[0], T.shape[0]
    Ta = np.zeros((m + 1, m + 1)); Ta[:m, :m] = T; Ta[m, m] = 1.0
    ZT = np.zeros(m + 1); ZT[:m] = Z @ T
    M = np.zeros((m + 1, m + 1))
    sa, sP = np.empty((n, m)), np.empty((n, m, m))
    for t in range(n - 1, -1, -1):
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
        sP[t] = S[:m, :m]
        sa[t] = Pf[m, :m] + Y[m, :m]
        K = kap.copy(); K[m] = 0.0
        G = Ta - np.outer(K, ZT)
        zeta = ZT.copy(); zeta[m] = v
        M = G.T @ M @ G + rF * np.outer(zeta, zeta)
    return sa, sP


if __name__ == "__main__":
    for name, kw in (("headline", dict(B=4)), ("c3", dict(B=2)), ("c2", dict(B=2))):
        w = W.WORKLOADS[name](**kw)
        y = w["y"].copy()
        y[[5, 6, 40, y.shape[0] - 1]] = np.nan
        flt = kn.kalman_filter(y, *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        for b in range(kw["B"]):
            pa = flt["predicted_states"][b].reshape(-1, w["m"])[:-1]
            sa, sP = bf_m(y[:, 0], w["T"][b], w["Z"][b, 0], w["H"][b, 0, 0], 0.0, pa, flt["predicted_covs"][b, :-1])
            ra_b = ra[b].reshape(sa.shape)
            print(name, b, "max|da|/max|a|", np.abs(sa - ra_b).max() / np.abs(ra_b).max(), "max|dP|/max|P|", np.abs(sP - rP[b]).max() / np.abs(rP[b]).max())
