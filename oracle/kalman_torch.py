"""CPU oracle for GRADIENTS: torch (fp64, CPU) restatement of the reference filter graph,
differentiated by torch.autograd -- the analogue of what PyTensor's autodiff does to the
reference's scan (the reference has no hand-written gradient).

TEST INFRASTRUCTURE ONLY (tests/, smoke, never the product path).

The forward follows the same operation sequence as oracle/kalman_numpy.py (which is pinned to
statsmodels golden vectors); tests/test_oracle_torch.py checks that the two forwards agree to
rounding, so the autograd gradients are gradients of the pinned function.  Every matrix entry
is an independent variable (general-matrix derivatives: d logdet F = F^-T, d F^-1 =
-F^-T (.) F^-T), exactly as an autodiff of `solve` / `det` treats them.

Reference lines: filters/kalman_filter.py:195-223 (mask, predict), :255-284 (Standard),
:444-505 (Univariate).
"""
from __future__ import annotations

import math

import numpy as np
import torch

LOG_2PI = math.log(2.0 * math.pi)


def _t(x):
    return x.transpose(-1, -2)


def kalman_logp(y, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                mvn_const="reference", time_varying=(), tapes=None):
    """Batched (leading B) torch tensors -> (logp (B,), ll_obs (B,n)). y: (n,p) numpy/tensor.
    Names in ``time_varying`` (subset of T,Z,R,H,Q,c,d) carry a time axis after the draw axis and step t uses slice t
    for its update and its predict (reference filters/kalman_filter.py:231-252).
    ``tapes``: a dict that receives the other four outputs of the reference's scan (filters/kalman_filter.py:166-212) --
    filtered_states (B,n,m), predicted_states (B,n+1,m), filtered_covs (B,n,m,m), predicted_covs (B,n+1,m,m), slot 0 of the
    predicted tapes = (a0, P0) -- as differentiable tensors."""
    y = torch.as_tensor(np.asarray(y), dtype=torch.float64)
    B, m = a0.shape
    tv = set(time_varying)
    p = Z.shape[-2]
    n = y.shape[0]
    eye_m = torch.eye(m, dtype=torch.float64)
    eye_p = torch.eye(p, dtype=torch.float64)
    a, P = a0, P0
    c = torch.zeros(B, m, dtype=torch.float64) if c is None else c
    d = torch.zeros(B, p, dtype=torch.float64) if d is None else d
    full = dict(T=T, Z=Z, R=R, H=H, Q=Q, c=c, d=d)
    lls = []
    fa, pa, fP, pP = [], [a0], [], [P0]
    for t in range(n):
        T, Z, R, H, Q, cz, dz = (full[k][:, t] if k in tv else full[k] for k in ("T", "Z", "R", "H", "Q", "c", "d"))
        RQR = R @ Q @ _t(R)
        yt = y[t]
        miss = torch.isnan(yt)
        obs = (~miss).to(torch.float64)
        nobs = int(obs.sum().item())
        ym = torch.where(miss, torch.zeros_like(yt), yt)
        Zm = Z * obs[None, :, None]
        Hm = H * obs[None, :, None] * obs[None, None, :]
        if filter_type in ("standard", "cholesky", "single"):
            dsign = 1.0 if filter_type == "single" else -1.0
            v = (ym[None] - torch.einsum("bij,bj->bi", Zm, a) + dsign * dz) * obs[None]
            M = P @ _t(Zm)
            F = Zm @ M + Hm + torch.diag(miss.to(torch.float64))[None]
            Fi = torch.linalg.inv(F)
            K = M @ Fi
            A = eye_m[None] - K @ Zm
            a_f = a + torch.einsum("bij,bj->bi", K, v)
            P_f = A @ P @ _t(A) + K @ Hm @ _t(K)
            quad = torch.einsum("bi,bij,bj->b", v, Fi, v)
            logdet = torch.log(torch.linalg.det(F))
            if mvn_const == "exact":
                nc = float(nobs)
            else:
                nc = float(p) if filter_type == "cholesky" else 1.0
            ll = -0.5 * (nc * LOG_2PI + logdet + quad) if nobs > 0 else torch.zeros(B, dtype=torch.float64)
        elif filter_type == "univariate":
            a_f, P_f = a, P
            llsum = torch.zeros(B, dtype=torch.float64)
            cnt = 0
            for q in range(p):
                if bool(miss[q]):
                    continue
                Zr = Z[:, q, :]
                v = ym[q] - torch.einsum("bj,bj->b", Zr, a_f) - dz[:, q]
                Mq = torch.einsum("bij,bj->bi", P_f, Zr)
                Fq = torch.einsum("bj,bj->b", Zr, Mq) + H[:, q, q]
                Kq = Mq / Fq[:, None]
                a_f = a_f + Kq * v[:, None]
                P_f = P_f - torch.einsum("bi,bj->bij", Kq, Kq) * Fq[:, None, None]
                llsum = llsum + torch.log(Fq) + v * v / Fq
                cnt += 1
            ll = -0.5 * (cnt * LOG_2PI + llsum)
        else:
            raise NotImplementedError(filter_type)
        lls.append(ll)
        a = torch.einsum("bij,bj->bi", T, a_f) + cz
        S = T @ P_f @ _t(T) + RQR
        P = 0.5 * (S + _t(S))
        fa.append(a_f)
        fP.append(P_f)
        pa.append(a)
        pP.append(P)
    ll_obs = torch.stack(lls, dim=1)
    if tapes is not None:
        tapes.update(filtered_states=torch.stack(fa, dim=1), predicted_states=torch.stack(pa, dim=1),
                     filtered_covs=torch.stack(fP, dim=1), predicted_covs=torch.stack(pP, dim=1))
    return ll_obs.sum(dim=1), ll_obs


NAMES = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")


def logp_and_grads(y, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                   mvn_const="reference", g_logp=None, g_ll_obs=None, time_varying=(), g_tapes=None):
    """Returns (logp, ll_obs, grads dict) with numpy arrays; grads of
    sum_b g_logp[b] logp[b] + sum_{b,t} g_ll_obs[b,t] ll_obs[b,t] (+ sum over ``g_tapes``: a dict of cotangents keyed by the
    tape names of ``kalman_logp``; then ``g_logp=None`` means 0, not ones).  The gradient of a matrix named in
    ``time_varying`` has that matrix's (B,n,.,.) shape."""
    B, m = np.asarray(a0).shape
    p = np.asarray(Z).shape[-2]
    n = np.asarray(y).shape[0]
    tv = set(time_varying)
    vals = dict(a0=a0, P0=P0, T=T, Z=Z, R=R, H=H, Q=Q,
                c=np.zeros((B, n, m) if "c" in tv else (B, m)) if c is None else c,
                d=np.zeros((B, n, p) if "d" in tv else (B, p)) if d is None else d)
    ts = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in vals.items()}
    tapes = {} if g_tapes else None
    logp, ll_obs = kalman_logp(y, ts["a0"], ts["P0"], ts["T"], ts["Z"], ts["R"], ts["H"], ts["Q"],
                               ts["c"], ts["d"], filter_type=filter_type, mvn_const=mvn_const, time_varying=tv, tapes=tapes)
    if g_logp is None:
        gl = torch.zeros(B, dtype=torch.float64) if g_tapes else torch.ones(B, dtype=torch.float64)
    else:
        gl = torch.as_tensor(g_logp)
    obj = (gl * logp).sum()
    for k, gk in (g_tapes or {}).items():
        obj = obj + (torch.as_tensor(np.asarray(gk, dtype=np.float64)) * tapes[k]).sum()
    if g_ll_obs is not None:
        obj = obj + (torch.as_tensor(g_ll_obs) * ll_obs).sum()
    obj.backward()
    grads = {k: (ts[k].grad.numpy() if ts[k].grad is not None else np.zeros(ts[k].shape))
             for k in NAMES}
    return logp.detach().numpy(), ll_obs.detach().numpy(), grads
