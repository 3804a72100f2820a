import sys, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
for (m, n, miss, cd) in ((10, 5, False, False), (10, 40, False, False), (10, 40, True, False), (10, 40, False, True), (5, 40, True, True), (13, 70, True, True)):
    B = 2
    w = W.random_dense(B, n, 1, m, 3, seed=m)
    y = w["y"].copy()
    if miss: y[[3, 17]] = np.nan
    rng = np.random.default_rng(0)
    c = 0.1 * rng.standard_normal((B, m)) if cd else None
    d = 0.1 * rng.standard_normal((B, 1)) if cd else None
    args = (y,) + tuple(w[k] for k in KEYS)
    _, _, gt = kt.logp_and_grads(*args, c=c, d=d, g_logp=np.ones(B))
    lp, info, g = eng.value_and_grad(*args, c, d)
    row = f"m={m} n={n} miss={miss} cd={cd}:"
    for k in KEYS + ("c", "d"):
        e = np.max(np.abs(g[k].cpu().numpy() - gt[k])) / (1 + np.max(np.abs(gt[k])))
        row += f" {k}:{e:.1e}"
    print(row)
