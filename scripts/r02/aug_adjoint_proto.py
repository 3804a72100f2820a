"""Development aid (CPU, numpy): reverse mode of the AUGMENTED forward recursion of kf_forward_b4.hip, checked against
torch autograd of the reference recursion (oracle/kalman_torch.py).  p = 1.

tile S (N x N, symmetric), indices 0..m-1 states, A = m (Z alpha), C = m + 1 (carrier of the mean, entries 2 a):
    u  = S e_A + yo e_C          yo = -2 (y -+ d) (observed) else 0
    F  = H + u_A ;  rs = ob / sqrt(F) ;  ut = rs u ;  Sf = S - ut ut'
    S' = T_e Sf T_e' + C_e       ll = ob (-1/2)(log 2 pi + log F + ut_C^2 / 4)
"""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_torch as kt


def build(w, b):
    m = w["m"]
    T, Z, R, H, Q, a0, P0 = (w[k][b] for k in ("T", "Z", "R", "H", "Q", "a0", "P0"))
    z = Z[0]
    N = m + 2
    A, C = m, m + 1
    Te = np.zeros((N, N)); Te[:m, :m] = T; Te[A, :m] = z @ T; Te[C, C] = 1.0
    RQR = R @ Q @ R.T
    c = np.zeros(m)
    Ce = np.zeros((N, N)); Ce[:m, :m] = RQR; Ce[A, :m] = z @ RQR; Ce[:m, A] = RQR @ z; Ce[A, A] = z @ RQR @ z
    Ce[C, :m] = 2 * c; Ce[:m, C] = 2 * c; Ce[A, C] = Ce[C, A] = 2 * z @ c
    S0 = np.zeros((N, N)); S0[:m, :m] = P0; S0[:m, A] = P0 @ z; S0[A, :m] = P0 @ z; S0[A, A] = z @ P0 @ z
    S0[:m, C] = 2 * a0; S0[C, :m] = 2 * a0; S0[A, C] = S0[C, A] = 2 * z @ a0
    return Te, Ce, S0, z, H[0, 0], RQR, T, R, Q, a0, P0


def run(w, b, y, g_logp=1.0, g_ll=None, stop_at=0):
    m = w["m"]; N = m + 2; A, C = m, m + 1
    Te, Ce, S0, z, H, RQR, T, R, Q, a0, P0 = build(w, b)
    n = len(y)
    eA = np.zeros(N); eA[A] = 1; eC = np.zeros(N); eC[C] = 1
    S = [S0]; logp = 0.0
    for t in range(n):
        ob = 0.0 if np.isnan(y[t]) else 1.0
        yo = -2 * y[t] if ob else 0.0
        u = S[t] @ eA + yo * eC
        F = H + u[A]
        rs = ob / np.sqrt(F)
        ut = rs * u
        Sf = S[t] - np.outer(ut, ut)
        S.append(Te @ Sf @ Te.T + Ce)
        logp += ob * -0.5 * (np.log(2 * np.pi) + np.log(F) + ut[C] ** 2 / 4)
    # backward
    G = np.zeros((N, N)); Tb = np.zeros((N, N)); Cb = np.zeros((N, N)); Hb = 0.0; yob = np.zeros(n)
    for t in range(n - 1, stop_at - 1, -1):
        gl = g_logp + (0.0 if g_ll is None else g_ll[t])
        ob = 0.0 if np.isnan(y[t]) else 1.0
        yo = -2 * y[t] if ob else 0.0
        u = S[t] @ eA + yo * eC
        F = H + u[A]
        rs = ob / np.sqrt(F)
        ut = rs * u
        Sf = S[t] - np.outer(ut, ut)
        Cb += G
        Y1t = G @ Te                      # (T_e' G)'
        Tb += 2 * Y1t @ Sf
        Gf = Te.T @ Y1t
        gut = -2 * Gf @ ut
        gut[C] += gl * ob * (-0.5) * ut[C] / 2
        Fb = gl * ob * (-0.5) / F if ob else 0.0
        ub = rs * gut
        rsb = gut @ u
        Fb += -0.5 * rsb * rs / F if ob else 0.0
        Hb += Fb
        yob[t] = ub[C]
        G = Gf + 0.5 * (np.outer(ub, eA) + np.outer(eA, ub))
        G[A, A] += Fb
    return logp, dict(G=G, Tb=Tb, Cb=Cb, Hb=Hb, yob=yob, S=S)


def chain(w, b, out, S_first, a_first, P_first):
    """accumulators -> gradients w.r.t. (T, Z, R, Q, H, c, d) and the adjoint (ab, Pb) of the first (a, P) the
    augmented loop saw (a0, P0 when it ran down to t = 0)"""
    m = w["m"]; A, C = m, m + 1
    Te, Ce, S0, z, H, RQR, T, R, Q, a0, P0 = build(w, b)
    Tb, Cb, G = out["Tb"], out["Cb"], out["G"]
    gT = Tb[:m, :m] + np.outer(z, Tb[A, :m])
    gZ = T @ Tb[A, :m]
    # C_e
    gA, gC, GAA, GAC = Cb[:m, A], Cb[:m, C], Cb[A, A], Cb[A, C]
    Mb = Cb[:m, :m] + np.outer(gA, z) + np.outer(z, gA) + GAA * np.outer(z, z)
    gZ = gZ + 2 * RQR @ gA + 2 * GAA * (RQR @ z)
    c = np.zeros(m)
    gc = 4 * gC + 4 * GAC * z
    gZ = gZ + 4 * GAC * c
    gR = Mb @ R @ Q.T + Mb.T @ R @ Q
    gQ = R.T @ Mb @ R
    # first tile
    gA, gC, GAA, GAC = G[:m, A], G[:m, C], G[A, A], G[A, C]
    Pb = G[:m, :m] + np.outer(gA, z) + np.outer(z, gA) + GAA * np.outer(z, z)
    ab = 4 * gC + 4 * GAC * z
    gZ = gZ + 2 * P_first @ gA + 2 * GAA * (P_first @ z) + 4 * GAC * a_first
    return dict(T=gT, Z=gZ[None], R=gR, Q=gQ, H=np.array([[out["Hb"]]]), c=gc, d=np.array([2.0 * out["yob"].sum()]),
                a0=ab, P0=Pb)


if __name__ == "__main__":
    w = W.random_dense(2, 25, 1, 5, 3, seed=4)
    y = w["y"][:, 0].copy(); y[7] = np.nan
    rng = np.random.default_rng(0)
    gll = rng.standard_normal(25) * 0.3
    for b in range(2):
        lp, out = run(w, b, y, 0.7, gll)
        g = chain(w, b, out, None, w["a0"][b], w["P0"][b])
        rl, _, gt = kt.logp_and_grads(w["y"] * 1.0 if False else y[:, None], w["a0"][b:b + 1], w["P0"][b:b + 1], w["T"][b:b + 1], w["Z"][b:b + 1],
                                      w["R"][b:b + 1], w["H"][b:b + 1], w["Q"][b:b + 1], g_logp=np.array([0.7]), g_ll_obs=gll[None])
        lp_t = kt.logp_and_grads(y[:, None], w["a0"][b:b + 1], w["P0"][b:b + 1], w["T"][b:b + 1], w["Z"][b:b + 1],
                                 w["R"][b:b + 1], w["H"][b:b + 1], w["Q"][b:b + 1])[0][0]
        print("logp", lp, lp_t)
        for k in ("T", "Z", "R", "Q", "H", "c", "d", "a0", "P0"):
            a_, b_ = np.asarray(g[k]), gt[k][0]
            if k == "P0":
                b_ = 0.5 * (b_ + b_.T)
            print(k, np.abs(a_.reshape(b_.shape) - b_).max(), np.abs(b_).max())
