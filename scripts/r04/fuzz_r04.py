"""Random-shape parity sweep of the round-4 paths.  usage: python scripts/r04/fuzz_r04.py [n_cases] [seed]
  tv      time-varying T / Z / H / c / d on the block tiles (p = 1, m <= 10; random flag subsets, static intercepts beside varying
          ones, missing observations, every filter type, partial tape requests) against the numpy oracle and the LDS kernel
  dense   dense H with 2 <= p <= 8 (Standard / Cholesky) through the whitening path against the oracle: whole-row and PARTIAL missing
          observations (device-side fallback), both mvn_const modes
  vgs     logp + gradient of the p = 1 block kernel with the steady-state adjoint (forced, KF_FLAG_STEADY_STATE) against the exact
          adjoint at every step and, on the first draws, torch autograd; missing observations, intercepts, slow draws
  fwd     the every-step forward of the block kernel after this round's changes (no exec masks, u in DPP row 2, absolute ring
          addresses) at random m <= 10, batch sizes around the pair-kernel boundary, against the oracle"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from oracle import kalman_torch as kt  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

TAPES = ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs")
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
GRADS = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
eng = KalmanEngine()
worst = dict(tv=0.0, tv_vs_lds=0.0, dense=0.0, vgs_vs_exact=0.0, vgs_vs_autograd=0.0, fwd=0.0)
counts = dict(tv=0, dense=0, vgs=0, fwd=0)
flagged = 0
t0 = time.time()


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def tapes_err(res, ref, k=None):
    e = rel(res.logp.cpu().numpy()[:k], ref["logp"][:k])
    for nm in TAPES:
        v = getattr(res, nm)
        if v is not None:
            e = max(e, rel(v.cpu().numpy()[:k], ref[nm][:k]))
    return e


for case in range(n_cases):
    kind = int(rng.integers(0, 10))
    seed = int(rng.integers(1 << 30))
    if kind < 3:  # ---- time-varying block kernel ---------------------------------------------------------------------------------
        m = int(rng.integers(1, 11))
        r = int(rng.integers(1, 5))
        n = int(rng.choice([1, 7, 33, 64, 150]))
        B = int(rng.choice([1, 5, 40]))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        w = W.random_dense(B, n, 1, m, r, seed=seed)
        names = [x for x in ("T", "Z", "H", "c", "d") if rng.integers(0, 2)] or ["T"]
        mats = dict(T=w["T"], Z=w["Z"], R=w["R"], H=w["H"], Q=w["Q"],
                    c=0.1 * rng.standard_normal((B, m)) if rng.integers(0, 2) or "c" in names else None,
                    d=0.1 * rng.standard_normal((B, 1)) if rng.integers(0, 2) or "d" in names else None)
        for nm in names:
            x = np.repeat(mats[nm][:, None], n, axis=1)
            wob = 0.8 + 0.2 * np.cos(0.4 * np.arange(n) + rng.uniform(0, 3))
            mats[nm] = x * wob.reshape((1, n) + (1,) * (x.ndim - 2)) + (0.03 * rng.standard_normal(x.shape) if nm in ("Z", "c", "d") else 0.0)
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 4))):
            y[rng.integers(0, n)] = np.nan
        args = (y, w["a0"], w["P0"], mats["T"], mats["Z"], mats["R"], mats["H"], mats["Q"], mats["c"], mats["d"])
        tv = tuple(names)
        kw = dict(filter_type=ft, mvn_const=str(rng.choice(["exact", "reference"])), time_varying=tv)
        stf, stp = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        assert eng.forward_kernel_name(B, n, 1, m, r, filter_type=ft, time_varying=tv) == "kf_fwd_b4tv_kernel"
        res = eng.forward(*args, store_filtered=stf, store_predicted=stp, **kw)
        ref = kn.kalman_filter_time_varying(*args, time_varying=tv, filter_type=ft, mvn_const=kw["mvn_const"])
        e = tapes_err(res, ref)
        os.environ["KF_FORWARD_B4TV"] = "0"
        lds = eng.forward(*args, store_filtered=stf, store_predicted=stp, **kw)
        os.environ.pop("KF_FORWARD_B4TV")
        e2 = rel(res.logp.cpu().numpy(), lds.logp.cpu().numpy())
        worst["tv"], worst["tv_vs_lds"] = max(worst["tv"], e), max(worst["tv_vs_lds"], e2)
        counts["tv"] += 1
        if e > 1e-9 or e2 > 1e-9 or res.info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG tv case {case}: m={m} r={r} n={n} B={B} {ft} {tv}: {e:.2e} {e2:.2e}", flush=True)
    elif kind < 6:  # ---- dense H by whitening ------------------------------------------------------------------------------------------
        p = int(rng.integers(2, 9))
        m = int(rng.integers(max(2, p // 2), 41))
        r = int(rng.integers(1, 6))
        n = int(rng.choice([5, 40, 90]))
        B = int(rng.choice([1, 7, 30]))
        ft = str(rng.choice(["standard", "cholesky"]))
        mv = str(rng.choice(["exact", "reference"]))
        w = W.random_dense(B, n, p, m, r, seed=seed, h_diag=False)
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 3))):
            y[rng.integers(0, n)] = np.nan
        partial = bool(rng.integers(0, 3) == 0)
        if partial:
            y[rng.integers(0, n), rng.integers(0, p)] = np.nan
        c = 0.1 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
        d = 0.1 * rng.standard_normal((B, p)) if rng.integers(0, 2) else None
        args = (y,) + tuple(w[x] for x in KEYS)
        res = eng.forward(*args, c, d, filter_type=ft, mvn_const=mv)
        ref = kn.kalman_filter(*args, c=c, d=d, filter_type=ft, mvn_const=mv, cholesky_triangle="correct", partial_missing="delete")
        e = tapes_err(res, ref)
        worst["dense"] = max(worst["dense"], e)
        counts["dense"] += 1
        if e > 1e-9 or res.info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG dense case {case}: p={p} m={m} r={r} n={n} B={B} {ft} {mv} partial={partial} "
                  f"kernel={eng.forward_kernel_name(B, n, p, m, r, filter_type=ft)}: {e:.2e}", flush=True)
    elif kind < 8:  # ---- steady-state adjoint behind the p = 1 block kernel ------------------------------------------------------------
        m = int(rng.integers(3, 11))
        r = int(rng.integers(1, 5))
        n = int(rng.choice([64, 130, 300, 700]))
        B = int(rng.choice([3, 20, 70]))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        w = W.random_dense(B, n, 1, m, r, seed=seed) if rng.integers(0, 2) else W.structural(B=B, n=n, seed=seed)
        m = w["m"]
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 3))):
            y[rng.integers(0, n)] = np.nan
        c = 0.05 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
        d = 0.1 * rng.standard_normal((B, 1)) if rng.integers(0, 2) else None
        gl = 1.0 + 0.3 * rng.standard_normal(B)
        args = (y,) + tuple(w[x] for x in KEYS)
        kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
        lp, info, g = eng.value_and_grad(*args, exact_recursion=False, **kw)
        lpx, _, gx = eng.value_and_grad(*args, exact_recursion=True, **kw)
        e = rel(lp.cpu().numpy(), lpx.cpu().numpy())
        for nm in GRADS:
            a_, x_ = g[nm].cpu().numpy().reshape(B, -1), gx[nm].cpu().numpy().reshape(B, -1)
            e = max(e, float(np.max(np.max(np.abs(a_ - x_), axis=1) / (1.0 + np.max(np.abs(x_), axis=1)))))
        worst["vgs_vs_exact"] = max(worst["vgs_vs_exact"], e)
        k = min(B, 2)
        e2 = 0.0
        if n <= 300:
            rlp, _, rg = kt.logp_and_grads(y, *[w[x][:k] for x in KEYS], c=None if c is None else c[:k], d=None if d is None else d[:k],
                                           filter_type=ft, mvn_const="exact", g_logp=gl[:k])
            for nm in GRADS:
                a_, ref = g[nm].cpu().numpy()[:k].reshape(rg[nm].shape), rg[nm]
                if nm in ("P0", "Q"):
                    a_, ref = a_ + np.swapaxes(a_, -1, -2), ref + np.swapaxes(ref, -1, -2)
                e2 = max(e2, float(np.max(np.abs(a_ - ref)) / (1.0 + np.max(np.abs(ref)))))
            worst["vgs_vs_autograd"] = max(worst["vgs_vs_autograd"], e2)
        counts["vgs"] += 1
        if e > 1e-10 or e2 > 1e-8 or info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG vgs case {case}: {w['name']} m={m} n={n} B={B} {ft}: vs exact {e:.2e}, vs autograd {e2:.2e}", flush=True)
    else:  # ---- every-step forward of the block kernel -------------------------------------------------------------------------------------
        m = int(rng.integers(1, 11))
        r = int(rng.integers(1, 5))
        n = int(rng.choice([1, 31, 32, 33, 100, 257]))
        B = int(rng.choice([1, 17, 1023, 1025, 2100]))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        w = W.random_dense(B, n, 1, m, r, seed=seed)
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 3))):
            y[rng.integers(0, n)] = np.nan
        k = min(B, 12)
        sel = np.unique(np.concatenate([np.arange(k), B - 1 - np.arange(min(k, B))]))
        eng.set_forward_kernel("b4")
        try:
            res = eng.forward(y, *[w[x] for x in KEYS], filter_type=ft, mvn_const="exact", exact_recursion=True)
        finally:
            eng.set_forward_kernel("auto")
        ref = kn.kalman_filter(y, *[w[x][sel] for x in KEYS], filter_type=ft, mvn_const="exact")
        e = rel(res.logp.cpu().numpy()[sel], ref["logp"])
        for nm in TAPES:
            e = max(e, rel(getattr(res, nm).cpu().numpy()[sel], ref[nm]))
        worst["fwd"] = max(worst["fwd"], e)
        counts["fwd"] += 1
        if e > 1e-9 or res.info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG fwd case {case}: m={m} r={r} n={n} B={B} {ft}: {e:.2e}", flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s: {counts}; flagged {flagged}; worst relative errors {worst}")
