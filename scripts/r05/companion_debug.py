import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
for m in (24, 33, 40, 70, 100):
    w = W.random_dense(4, 12, 1, m, 1, seed=60 + m)
    T = np.zeros_like(w["T"])
    T[:, :, 0] = 0.5 * np.random.default_rng(m).standard_normal((4, m)) / np.arange(1, m + 1)
    T[:, np.arange(m - 1), np.arange(1, m)] = 1.0
    w["T"] = T
    y = w["y"]
    res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact")
    da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy())
    for env in ({}, {"KF_BIG_SPARSE": "0"}, {"KF_BIG_SPARSE": "0", "KF_BIG_TMASK": "0"}):
        for k, v in env.items():
            os.environ[k] = v
        sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs)
        for k in env:
            del os.environ[k]
        print(m, env, "max |dP|", float(np.abs(sP.cpu().numpy() - dP).max()), "max |da|", float(np.abs(sa.cpu().numpy() - da).max()), flush=True)
