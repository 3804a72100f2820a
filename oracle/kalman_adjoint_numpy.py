"""Hand-derived reverse-mode (adjoint) recursion of the Kalman log-likelihood, in numpy.

TEST INFRASTRUCTURE ONLY.  This is the algorithm the HIP backward kernel
(pymc_statespace_amd/csrc/kf_backward_lds.hip) implements, written down once in numpy so the
derivation can be checked on CPU against torch.autograd of the reference operation sequence
(oracle/kalman_torch.py) -- see tests/test_oracle_torch.py.

Per step t (inputs a = a_pred[t], P = P_pred[t] from the forward tape; ~ = masked):
  forward recompute   v = W(y - Z a - d);  M = P Z~';  F = Z~ M + H~ + D_miss;  Fi = F^-1
                      K = M Fi;  a_f = a + K v;  AP = P - K M';  P_f = AP A' + K H~ K'
  predict reverse     Gm = (Pb' + Pb'')/2;  Tb += ab' a_f' + 2 Gm (T P_f);  cb += ab'
                      Gs += Gm;  ab_f = T' ab';  Sb = T' Gm T
  update reverse      U = Sb K;  Pb = A' Sb A;  Kb = U (H~+H~') + ab_f v' - 2 Sb (A M),  A M = M - K (Z~ M)
                      Hb += K' U;  Zb -= 2 U' AP;  vb = K' ab_f - g Fi v
                      Fib = -g/2 v v' + M' Kb;  Fb = -g/2 Fi - Fi Fib Fi
                      Mb = Kb Fi + Z~' Fb;  Zb += Fb M';  Hb += Fb;  Pb += Mb Z~;  Zb += Mb' P
                      Zb -= vb a';  ab = ab_f - Z~' vb;  db -= vb
  finally             R_b = Gs R (Q + Q');  Q_b = R' Gs R;  a0_b = ab;  P0_b = Pb
"""
import numpy as np


def _T(x):
    return np.swapaxes(x, -1, -2)


def logp_grads(y, a0, P0, T, Z, R, H, Q, c=None, d=None, g_logp=None, g_ll_obs=None,
               single_dsign=False):
    """Batched (leading B).  Returns dict of gradients a0,P0,T,Z,R,H,Q,c,d."""
    y = np.asarray(y, dtype=np.float64)
    B, m = a0.shape
    p = Z.shape[1]
    n = y.shape[0]
    cz = np.zeros((B, m)) if c is None else c
    dz = np.zeros((B, p)) if d is None else d
    dsign = 1.0 if single_dsign else -1.0
    gl = np.ones(B) if g_logp is None else np.asarray(g_logp)
    gll = np.zeros((B, n)) if g_ll_obs is None else np.asarray(g_ll_obs)
    RQR = R @ Q @ _T(R)
    eye_m = np.eye(m)

    # forward tape of predicted moments
    pa = np.empty((B, n + 1, m))
    pP = np.empty((B, n + 1, m, m))
    a, P = a0.copy(), P0.copy()
    pa[:, 0], pP[:, 0] = a, P

    def fwd(a, P, t):
        yt = y[t]
        miss = np.isnan(yt)
        obs = (~miss).astype(float)
        ym = np.where(miss, 0.0, yt)
        Zm = Z * obs[None, :, None]
        Hm = H * obs[None, :, None] * obs[None, None, :]
        v = (ym[None] - np.einsum("bij,bj->bi", Zm, a) + dsign * dz) * obs[None]
        M = P @ _T(Zm)
        Fp = Zm @ M + Hm
        F = Fp + np.diag(miss.astype(float))[None]
        Fi = np.linalg.inv(F)
        K = M @ Fi
        a_f = a + np.einsum("bij,bj->bi", K, v)
        AP = P - K @ _T(M)
        A = eye_m[None] - K @ Zm
        P_f = AP @ _T(A) + K @ Hm @ _T(K)
        return dict(obs=obs, Zm=Zm, Hm=Hm, v=v, M=M, Fp=Fp, Fi=Fi, K=K, a_f=a_f, AP=AP, A=A,
                    P_f=P_f, all_missing=bool(miss.all()))

    for t in range(n):
        s = fwd(a, P, t)
        a = np.einsum("bij,bj->bi", T, s["a_f"]) + cz
        S = T @ s["P_f"] @ _T(T) + RQR
        P = 0.5 * (S + _T(S))
        pa[:, t + 1], pP[:, t + 1] = a, P

    ab = np.zeros((B, m))
    Pb = np.zeros((B, m, m))
    Tb = np.zeros((B, m, m))
    Gs = np.zeros((B, m, m))
    Zb = np.zeros((B, p, m))
    Hb = np.zeros((B, p, p))
    cb = np.zeros((B, m))
    db = np.zeros((B, p))
    for t in range(n - 1, -1, -1):
        a, P = pa[:, t], pP[:, t]
        s = fwd(a, P, t)
        g = gl + gll[:, t]
        # predict reverse
        Gm = 0.5 * (Pb + _T(Pb))
        W = T @ s["P_f"]
        Tb += np.einsum("bi,bj->bij", ab, s["a_f"]) + 2.0 * Gm @ W
        cb += ab
        Gs += Gm
        ab_f = np.einsum("bji,bj->bi", T, ab)
        Sb = _T(T) @ Gm @ T
        if s["all_missing"]:
            ab, Pb = ab_f, Sb
            continue
        obs, Zm, Hm, v, M, Fi, K = s["obs"], s["Zm"], s["Hm"], s["v"], s["M"], s["Fi"], s["K"]
        U = Sb @ K
        KtU = _T(K) @ U
        Pb_new = Sb - _T(Zm) @ _T(U) - U @ Zm + _T(Zm) @ KtU @ Zm
        AM = M - K @ (Zm @ M)
        Kb = U @ (Hm + _T(Hm)) + np.einsum("bi,bq->biq", ab_f, v) - 2.0 * Sb @ AM
        Hb_s = KtU.copy()
        Zb_s = -2.0 * _T(U) @ s["AP"]
        vb = np.einsum("bjq,bj->bq", K, ab_f) - g[:, None] * np.einsum("bij,bj->bi", Fi, v)
        Fib = -0.5 * g[:, None, None] * np.einsum("bi,bj->bij", v, v) + _T(M) @ Kb
        Fb = -0.5 * g[:, None, None] * Fi - Fi @ Fib @ Fi
        Mb = Kb @ Fi + _T(Zm) @ Fb
        Zb_s += Fb @ _T(M)
        Hb_s += Fb
        Pb_new += Mb @ Zm
        Zb_s += _T(Mb) @ P
        Zb_s -= np.einsum("bq,bj->bqj", vb, a)
        ab = ab_f - np.einsum("bqj,bq->bj", Zm, vb)
        db += dsign * vb * obs[None]
        Zb += Zb_s * obs[None, :, None]
        Hb += Hb_s * obs[None, :, None] * obs[None, None, :]
        Pb = Pb_new
    Rb = Gs @ R @ (Q + _T(Q))
    Qb = _T(R) @ Gs @ R
    return dict(a0=ab, P0=Pb, T=Tb, Z=Zb, R=Rb, H=Hb, Q=Qb, c=cb, d=db)
