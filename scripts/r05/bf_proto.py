"""Development aid (CPU): the smoothed moments from the filter's PREDICTED tapes without inverting the prediction covariance
(de Jong / Durbin-Koopman backward recursion, p = 1), in exactly the augmented-tile algebra of kf_smoother_bf.hip:
    N_aug = [[N, r], [r', .]]   N~ = T_aug' N_aug T_aug    w = N~ K_aug   s = K' w
    w~ = w - (s + 1/F)/2 Z_aug - (v/F) e_JC      N_aug <- N~ - Z_aug w~' - w~ Z_aug'
    Y = N_aug P_ext (row JC = (P r)')   P_s = P - P Y[:m,:m]   a_s = a + Y[JC,:m]
against the RTS oracle (reference kalman_smoother.py:56-104 restated in oracle/kalman_numpy.py)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn


def bf_smoother(y, T, Z, H, d, pred_a, pred_P):
    n, m = y.shape[0], T.shape[0]
    Ta = np.zeros((m + 1, m + 1)); Ta[:m, :m] = T; Ta[m, m] = 1.0
    Za = np.zeros(m + 1); Za[:m] = Z
    e = np.zeros(m + 1); e[m] = 1.0
    N = np.zeros((m + 1, m + 1))
    sa = np.empty((n, m)); sP = np.empty((n, m, m))
    for t in range(n - 1, -1, -1):
        P, a = pred_P[t], pred_a[t]
        Nt = Ta.T @ N @ Ta
        ob = not np.isnan(y[t])
        if ob:
            zP = Z @ P
            F = H + zP @ Z
            v = y[t] - Z @ a - d
            K = np.zeros(m + 1); K[:m] = zP / F
            w = Nt @ K
            s = K @ w
            wt = w - 0.5 * (s + 1.0 / F) * Za - (v / F) * e
            N = Nt - np.outer(Za, wt) - np.outer(wt, Za)
        else:
            N = Nt
        Pe = np.zeros((m + 1, m + 1)); Pe[:m, :m] = P
        Y = N @ Pe
        sa[t] = a + Y[m, :m]
        sP[t] = P - P @ Y[:m, :m]
    return sa, sP


if __name__ == "__main__":
    for name, kw in (("headline", dict(B=4)), ("c3", dict(B=2)), ("c2", dict(B=2))):
        w = W.WORKLOADS[name](**kw)
        y = w["y"].copy()
        y[[5, 6, 40]] = np.nan
        flt = kn.kalman_filter(y, *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
        for b in range(kw["B"]):
            sa, sP = bf_smoother(y[:, 0], w["T"][b], w["Z"][b, 0], w["H"][b, 0, 0], 0.0, flt["predicted_states"][b, :-1, :, 0] if flt["predicted_states"].ndim == 4 else flt["predicted_states"][b, :-1], flt["predicted_covs"][b, :-1])
            ra_b = ra[b].reshape(sa.shape)
            print(name, b, "max|da|/max|a|", np.abs(sa - ra_b).max() / np.abs(ra_b).max(), "max|dP|/max|P|", np.abs(sP - rP[b]).max() / np.abs(rP[b]).max())
