"""Random shapes on the kernels that have a steady-state phase (long series, missing data, per-draw y, intercepts): default run and
KF_FLAG_EXACT_RECURSION run against the numpy oracle.  usage: python scripts/r02/steady_fuzz.py [n_cases] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine

TAPES = ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
eng = KalmanEngine()
worst = dict(logp=0.0, tapes=0.0, dflt_vs_exact=0.0)
engaged, flagged, kernels = 0, 0, {}
t0 = time.time()
for case in range(n_cases):
    kind = rng.integers(0, 3)
    if kind == 0:
        p, m = 1, int(rng.integers(1, 5))
    elif kind == 1:
        p, m = 1, int(rng.integers(5, 11))
    else:
        p = int(rng.integers(3, 6)); m = int(rng.integers(max(8, p), 16))
    r = int(rng.integers(1, m + 1))
    n = int(rng.choice([64, 100, 257, 400]))
    B = int(rng.integers(1, 7))
    ft = str(rng.choice(["standard", "cholesky", "univariate"] + (["single"] if p == 1 else [])))
    mvn = str(rng.choice(["reference", "exact"]))
    w = W.random_dense(B, n, p, m, r, seed=int(rng.integers(1 << 30)))
    y = w["y"].copy()
    if rng.random() < 0.5:
        y[rng.integers(0, n)] = np.nan
        if p > 1: y[rng.integers(0, n), rng.integers(0, p)] = np.nan
    if rng.random() < 0.3:
        y = y[None] + 0.1 * rng.standard_normal((B,) + y.shape)
        if rng.random() < 0.5: y[rng.integers(0, B), rng.integers(0, n)] = np.nan
    c = 0.1 * rng.standard_normal((B, m)) if rng.random() < 0.5 else None
    d = 0.1 * rng.standard_normal((B, p)) if rng.random() < 0.5 else None
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    name = eng.forward_kernel_name(B, n, p, m, r, ft, True)
    kernels[name] = kernels.get(name, 0) + 1
    ref = kn.kalman_filter(*args, c=c, d=d, filter_type=ft, mvn_const=mvn, cholesky_triangle="correct", partial_missing="delete")
    out = {}
    for exact in (False, True):
        res = eng.forward(*args, c=c, d=d, filter_type=ft, mvn_const=mvn, exact_recursion=exact, h_diagonal=True)
        out[exact] = res
        e_lp = float(np.max(np.abs(res.logp.cpu().numpy() - ref["logp"]) / (1.0 + np.abs(ref["logp"]) / 1e4)))
        e_tp = max(float(np.max(np.abs(getattr(res, k).cpu().numpy() - ref[k]) / (1.0 + np.abs(ref[k])))) for k in TAPES)
        worst["logp"] = max(worst["logp"], e_lp); worst["tapes"] = max(worst["tapes"], e_tp)
        if not (e_lp < 1e-8 and e_tp < 1e-9) or res.info.cpu().numpy().any():
            flagged += 1
            print(f"FLAG case {case} p={p} m={m} r={r} n={n} B={B} {ft} {mvn} exact={exact} {name}: dlogp {e_lp:.1e} dtape {e_tp:.1e} info {res.info.cpu().numpy()}", flush=True)
    dd = max(float(np.max(np.abs(getattr(out[False], k).cpu().numpy() - getattr(out[True], k).cpu().numpy()) / (1.0 + np.abs(ref[k])))) for k in TAPES)
    worst["dflt_vs_exact"] = max(worst["dflt_vs_exact"], dd)
    engaged += dd > 0
print("kernels:", kernels)
print(f"cases {n_cases}: steady phase changed the bits in {engaged}; worst {({k: f'{v:.2e}' for k, v in worst.items()})}; flagged: {flagged} ({time.time() - t0:.0f} s)")
