"""Random-shape parity sweep of the round-3 paths.  usage: python scripts/r03/fuzz_r03.py [n_cases] [seed]
  forward, p = 1, m <= 10: library default, KF_FLAG_STEADY_STATE (split pipeline from 256 steps; pair-kernel tail) and
      KF_FLAG_EXACT_RECURSION (pair kernel for <= 1024 draws with tapes) against the numpy oracle; batches up to 1600 draws
  logp + gradient, 3 <= p <= 5, m >= 8: steady-state adjoint against the exact adjoint at every step (KF_VG_STEADY=0) and, on
      the first draws, against torch autograd
  smoother, m <= 4: lane kernel (pseudo-inverse in registers) against the numpy oracle (numpy.linalg.pinv per step)
  smoother, 5 <= m <= 15, up to 500 steps: block kernel with the gain reuse against every step in full and the oracle
  forward, 65 <= m <= 140, p = 1, log-likelihood + state tapes: the LDS-resident kernel (m <= 96) / the three-launch workspace
      step (m > 96) against the oracle, missing observations and intercepts"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W
from oracle import kalman_numpy as kn
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine

TAPES = ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs")
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
eng = KalmanEngine()
worst = dict(fwd_logp=0.0, fwd_tapes=0.0, steady_vs_exact=0.0, vg_vs_exact=0.0, vg_vs_autograd=0.0, smooth=0.0, smooth_b4=0.0,
             smooth_b4_vs_full=0.0, mid_m_logp=0.0)
flagged, counts = 0, dict(fwd=0, vg=0, smooth=0, smooth_b4=0, mid_m=0)
t0 = time.time()


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


for case in range(n_cases):
    kind = int(rng.integers(0, 14))
    if kind >= 12:  # ---- forward, 65 <= m <= 140, log-likelihood and state tapes ----------------------------------------------------
        import torch
        from pymc_statespace_amd.engine import FilterResult
        m = int(rng.integers(65, 141))
        r = int(rng.integers(1, 6))
        n = int(rng.choice([3, 40, 120]))
        B = int(rng.choice([1, 9, 20]))
        w = W.random_dense(B, n, 1, m, r, seed=int(rng.integers(1 << 30)))
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 4))):
            y[rng.integers(0, n)] = np.nan
        c = 0.1 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
        d = 0.1 * rng.standard_normal((B, 1)) if rng.integers(0, 2) else None
        ft = str(rng.choice(["standard", "cholesky", "single"]))
        ref = kn.kalman_filter(y, *[w[x] for x in KEYS], c=c, d=d, filter_type=ft, mvn_const="exact")
        dev = torch.device("cuda")
        out = FilterResult(logp=torch.empty(B, device=dev, dtype=torch.float64), ll_obs=torch.empty(B, n, device=dev, dtype=torch.float64),
                           filtered_states=torch.empty(B, n, m, device=dev, dtype=torch.float64),
                           predicted_states=torch.empty(B, n + 1, m, device=dev, dtype=torch.float64),
                           filtered_covs=None, predicted_covs=None, info=torch.empty(B, device=dev, dtype=torch.int32))
        res = eng.forward(y, *[w[x] for x in KEYS], c=c, d=d, filter_type=ft, mvn_const="exact", out=out)
        e = max(rel(res.logp.cpu().numpy(), ref["logp"]), rel(res.ll_obs.cpu().numpy(), ref["ll_obs"]),
                rel(res.filtered_states.cpu().numpy(), ref["filtered_states"]), rel(res.predicted_states.cpu().numpy(), ref["predicted_states"]))
        worst["mid_m_logp"] = max(worst["mid_m_logp"], e)
        if e > 1e-9 or res.info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG mid-m forward case {case}: m={m} r={r} n={n} B={B} {ft}: {e:.2e}", flush=True)
        counts["mid_m"] += 1
        continue
    if kind >= 10:  # ---- smoother, 5 <= m <= 15: gain reuse ------------------------------------------------------------------------
        m = int(rng.integers(5, 16))
        n = int(rng.choice([40, 200, 500]))
        B = int(rng.choice([3, 40]))
        w = W.random_dense(B, n, 1, m, m, seed=int(rng.integers(1 << 30)))
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 3))):
            y[rng.integers(0, n)] = np.nan
        res = eng.forward(y, *[w[x] for x in KEYS], mvn_const="exact", exact_recursion=bool(rng.integers(0, 2)))
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        ea, eP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs, exact_recursion=True)
        k = min(B, 8)
        ra, rP = kn.kalman_smoother(w["T"][:k], w["R"][:k], w["Q"][:k], res.filtered_states[:k].cpu().numpy(),
                                    res.filtered_covs[:k].cpu().numpy())
        e = max(rel(sa[:k].cpu().numpy(), ra), rel(sP[:k].cpu().numpy(), rP))
        e2 = max(rel(sa.cpu().numpy(), ea.cpu().numpy()), rel(sP.cpu().numpy(), eP.cpu().numpy()))
        worst["smooth_b4"] = max(worst["smooth_b4"], e)
        worst["smooth_b4_vs_full"] = max(worst["smooth_b4_vs_full"], e2)
        if e > 1e-9 or e2 > 3e-11:  # (the inverse amplifies the tolerated drift of the tape by cond(P_hat))
            flagged += 1
            print(f"FLAG block smoother case {case}: m={m} n={n} B={B}: vs oracle {e:.2e} vs every step in full {e2:.2e}", flush=True)
        counts["smooth_b4"] += 1
        continue
    if kind < 6:  # ---- forward, p = 1 ------------------------------------------------------------------------------------------
        m = int(rng.integers(1, 11))
        r = int(rng.integers(1, m + 1))
        n = int(rng.choice([64, 130, 256, 300, 513, 800]))
        B = int(rng.choice([3, 17, 40, 1100, 1600], p=[0.3, 0.3, 0.2, 0.1, 0.1]))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        mvn = str(rng.choice(["reference", "exact"]))
        w = W.random_dense(B, n, 1, m, r, seed=int(rng.integers(1 << 30)))
        y = w["y"].copy()
        if rng.random() < 0.5:
            y[rng.integers(0, n)] = np.nan
        if rng.random() < 0.25:
            y = np.repeat(y[None], B, 0) + 0.1 * rng.standard_normal((B,) + y.shape)
        c = 0.1 * rng.standard_normal((B, m)) if rng.random() < 0.5 else None
        d = 0.1 * rng.standard_normal((B, 1)) if rng.random() < 0.5 else None
        k = min(B, 24)
        yk = y if y.ndim == 2 else y[:k]
        ref = kn.kalman_filter(yk, *[w[x][:k] for x in KEYS], c=None if c is None else c[:k], d=None if d is None else d[:k],
                               filter_type=ft, mvn_const=mvn)
        outs = {}
        for mode in (None, False, True):
            res = eng.forward(y, *[w[x] for x in KEYS], c=c, d=d, filter_type=ft, mvn_const=mvn, exact_recursion=mode)
            e_lp = float(np.max(np.abs(res.logp[:k].cpu().numpy() - ref["logp"]) / np.maximum(1.0, np.abs(ref["logp"]) / 1e4)))
            e_tp = max(rel(getattr(res, t)[:k].cpu().numpy(), ref[t]) for t in TAPES)
            worst["fwd_logp"] = max(worst["fwd_logp"], e_lp)
            worst["fwd_tapes"] = max(worst["fwd_tapes"], e_tp)
            bad = e_lp > 1e-8 or e_tp > 1e-9 or res.info.abs().sum().item() != 0
            if bad:
                flagged += 1
                print(f"FLAG forward case {case}: m={m} r={r} n={n} B={B} {ft} {mvn} mode={mode}: dlogp {e_lp:.2e} tapes {e_tp:.2e}", flush=True)
            outs[mode] = res
        worst["steady_vs_exact"] = max(worst["steady_vs_exact"], max(rel(getattr(outs[False], t).cpu().numpy(),
                                                                       getattr(outs[True], t).cpu().numpy()) for t in TAPES))
        counts["fwd"] += 1
    elif kind < 8:  # ---- logp + gradient, VARMAX-like shapes ------------------------------------------------------------------------
        p = int(rng.integers(3, 6))
        m = int(rng.integers(max(8, p), 16))
        r = int(rng.integers(1, m + 1))
        n = int(rng.choice([64, 120, 260]))
        B = int(rng.choice([5, 30, 200]))
        ft = str(rng.choice(["standard", "cholesky", "univariate"]))
        w = W.random_dense(B, n, p, m, r, seed=int(rng.integers(1 << 30)))
        y = w["y"].copy()
        if rng.random() < 0.5:
            y[rng.integers(0, n // 2)] = np.nan
            y[rng.integers(0, n // 2), rng.integers(0, p)] = np.nan
        c = 0.1 * rng.standard_normal((B, m)) if rng.random() < 0.5 else None
        d = 0.1 * rng.standard_normal((B, p)) if rng.random() < 0.5 else None
        gl = rng.standard_normal(B)
        kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact", h_diagonal=True)
        lp, info, g = eng.value_and_grad(y, *[w[x] for x in KEYS], **kw)
        os.environ["KF_VG_STEADY"] = "0"
        lpx, _, gx = eng.value_and_grad(y, *[w[x] for x in KEYS], **kw)
        os.environ.pop("KF_VG_STEADY")
        e1 = max(float((g[nm] - gx[nm]).abs().max().item() / (1.0 + gx[nm].abs().max().item())) for nm in g)
        k = 3
        rlp, _, rg = kt.logp_and_grads(y, *[w[x][:k] for x in KEYS], c=None if c is None else c[:k], d=None if d is None else d[:k],
                                       filter_type=ft, mvn_const="exact", g_logp=gl[:k])
        e2 = 0.0
        for nm in g:
            a_, b_ = g[nm][:k].cpu().numpy(), rg[nm]
            if nm == "H":
                b_ = b_ * np.eye(p)
            a_ = a_.reshape(b_.shape)
            if nm in ("P0", "Q"):
                a_, b_ = a_ + np.swapaxes(a_, -1, -2), b_ + np.swapaxes(b_, -1, -2)
            e2 = max(e2, float(np.max(np.abs(a_ - b_)) / (1.0 + np.max(np.abs(b_)))))
        worst["vg_vs_exact"] = max(worst["vg_vs_exact"], e1)
        worst["vg_vs_autograd"] = max(worst["vg_vs_autograd"], e2)
        if e1 > 1e-10 or e2 > 1e-9 or float(np.max(np.abs(lp[:k].cpu().numpy() - rlp))) > 1e-8:
            flagged += 1
            print(f"FLAG gradient case {case}: p={p} m={m} r={r} n={n} B={B} {ft}: vs exact {e1:.2e} vs autograd {e2:.2e}", flush=True)
        counts["vg"] += 1
    else:  # ---- smoother, m <= 4 -------------------------------------------------------------------------------------------------
        m = int(rng.integers(1, 5))
        r = int(rng.integers(1, m + 1))
        n = int(rng.choice([2, 33, 100]))
        B = int(rng.choice([3, 70, 300]))
        w = W.random_dense(B, n, 1, m, r, seed=int(rng.integers(1 << 30)))
        res = eng.forward(w["y"], *[w[x] for x in KEYS], mvn_const="exact", exact_recursion=True)
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        k = min(B, 16)
        ra, rP = kn.kalman_smoother(w["T"][:k], w["R"][:k], w["Q"][:k], res.filtered_states[:k].cpu().numpy(),
                                    res.filtered_covs[:k].cpu().numpy())
        e = max(rel(sa[:k].cpu().numpy(), ra), rel(sP[:k].cpu().numpy(), rP))
        if r == m:  # (r < m: P_hat may be singular -- the rank decision of the pseudo-inverse sits at rounding level, as in the reference)
            worst["smooth"] = max(worst["smooth"], e)
            if e > 1e-8:
                flagged += 1
                print(f"FLAG smoother case {case}: m={m} r={r} n={n} B={B}: {e:.2e}", flush=True)
        counts["smooth"] += 1
print(f"{n_cases} cases ({counts}) in {time.time() - t0:.0f} s: flagged {flagged}; worst errors {worst}")
