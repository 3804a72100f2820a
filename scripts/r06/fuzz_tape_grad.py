"""Random-shape sweep of kf_filter_backward_tapes against torch.autograd through the torch restatement of the reference's scan
(oracle/kalman_torch.py, tapes pinned to the numpy oracle), FIXED tolerance: every gradient within 1e-7 of its largest entry.
Random subsets of the six cotangents, every filter type, dense / diagonal H, c / d, missing observations, both kernels (the tile
adjoint for m <= 16 with scalar observations, the LDS adjoint else or when forced).
usage: python scripts/r06/fuzz_tape_grad.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_torch as kt  # noqa: E402  (the checker; this script is test infrastructure)
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 91
TOL = 1e-7
eng = KalmanEngine()
rng = np.random.default_rng(SEED)
flagged, worst, nlds = [], 0.0, 0
for case in range(N):
    seed = SEED * 100000 + case
    p = int(rng.choice([1, 1, 2, 3]))
    m = int(rng.integers(1, 13))
    r = int(rng.integers(1, m + 1))
    n = int(rng.integers(3, 40))
    B = int(rng.choice([1, 3, 7]))
    ft = str(rng.choice(["standard", "cholesky", "univariate", "single"])) if p == 1 else str(rng.choice(["standard", "univariate"]))
    dense_h = p > 1 and ft == "standard" and bool(rng.integers(0, 2))
    w = W.random_dense(B, n, p, m, r, seed=seed, h_diag=not dense_h)
    y = w["y"].copy()
    if rng.integers(0, 3) == 0:
        y[int(rng.integers(0, n))] = np.nan
        y[int(rng.integers(0, n)), int(rng.integers(0, p))] = np.nan
    c = 0.1 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
    d = 0.2 * rng.standard_normal((B, p)) if rng.integers(0, 2) else None
    gt = {}
    sym = lambda x: 0.5 * (x + np.swapaxes(x, -1, -2))  # noqa: E731
    if rng.integers(0, 2):
        gt["filtered_states"] = rng.standard_normal((B, n, m))
    if rng.integers(0, 2):
        gt["predicted_states"] = rng.standard_normal((B, n + 1, m))
    if rng.integers(0, 2):
        gt["filtered_covs"] = sym(rng.standard_normal((B, n, m, m)))
    if rng.integers(0, 2) or not gt:
        gt["predicted_covs"] = sym(rng.standard_normal((B, n + 1, m, m)))
    g_logp = rng.standard_normal(B) if rng.integers(0, 2) else None
    g_ll = rng.standard_normal((B, n)) if rng.integers(0, 2) else None
    force_lds = bool(rng.integers(0, 4) == 0)
    tag = f"p={p} m={m} r={r} n={n} B={B} ft={ft} denseH={dense_h} lds={force_lds} cot={sorted(gt)} seed={seed}"
    args = (y,) + tuple(w[k] for k in KEYS)
    if force_lds:
        eng.set_forward_kernel("lds")
    try:
        res = eng.forward(*args, c=c, d=d, filter_type=ft, mvn_const="exact")
        if res.info.any():
            continue
        g = eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g_logp,
                         g_ll_obs=g_ll, g_filtered_states=gt.get("filtered_states"), g_predicted_states=gt.get("predicted_states"),
                         g_filtered_covs=gt.get("filtered_covs"), g_predicted_covs=gt.get("predicted_covs"), filter_type=ft,
                         mvn_const="exact", want_cd=True)
    finally:
        if force_lds:
            eng.set_forward_kernel("auto")
    nlds += int(force_lds or dense_h or m > 16)
    oft = "standard" if ft in ("cholesky",) else ft
    _, _, ref = kt.logp_and_grads(y, *[w[k] for k in KEYS], c=c, d=d, filter_type=oft, mvn_const="exact",
                                  g_logp=(np.zeros(B) if g_logp is None else g_logp), g_ll_obs=g_ll, g_tapes=gt)
    dev = 0.0
    for k in kt.NAMES:
        a, b = g[k].cpu().numpy(), ref[k]
        if k == "P0" and (p > 1 or ft == "univariate"):   # scalar-observation kernels / the Univariate restatement: the symmetric part
            a, b = sym(a), sym(b)
        if k == "H" and p > 1 and not dense_h:            # diagonal-H models: the diagonal (include/kf.h)
            a, b = a * np.eye(p), b * np.eye(p)
        dev = max(dev, float(np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b)))))
    worst = max(worst, dev)
    if not dev <= TOL:
        flagged.append((tag, dev))
print(f"{N} cases, {nlds} on the LDS adjoint, worst deviation {worst:.2e} (tolerance {TOL:g}), {len(flagged)} flagged")
for f in flagged[:20]:
    print("  FLAGGED", f)
sys.exit(1 if flagged else 0)
