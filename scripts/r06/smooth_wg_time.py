"""kf_smoother_from_filter for 16 <= m <= 79 on chip (kf_smoother_bf_wg.hip) against the HBM-workspace path it replaces
(KF_SMOOTHER_BF_WG=0) and the RTS tile kernels (kf_smoother): parity against the numpy restatement on a few draws, then time at
1024 draws x 300 steps (profiles/r05/smoother_mid_m_rank.txt's shape)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


small = len(sys.argv) > 1 and sys.argv[1] == "parity"
for m in (16, 20, 31, 32, 47, 48, 63, 64, 79):
    for r in (1, m):
        w = W.random_dense(3, 40, 1, m, r, seed=m + r)
        y = w["y"].copy()
        y[[0, 7, 39]] = np.nan
        d = 0.1 * np.random.default_rng(m).standard_normal((3, 1))
        res = eng.forward(y, *[w[k] for k in KEYS], d=d, mvn_const="exact")
        sa, sP = eng.smooth_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states, res.predicted_covs, d=d)
        da, dP = kn.smoother_from_filter(y, w["T"], w["Z"], w["H"], res.predicted_states.cpu().numpy(), res.predicted_covs.cpu().numpy(), d=d)
        sc = 1.0 + np.abs(res.filtered_covs.cpu().numpy()).max()
        ea, eP = np.abs(sa.cpu().numpy() - da).max() / sc, np.abs(sP.cpu().numpy() - dP).max() / sc
        line = f"m={m:3d} r={r:3d}  vs restatement: states {ea:.1e} covariances {eP:.1e}"
        if not small:
            wb = W.random_dense(1024, 300, 1, m, r, seed=1000 + m + r)
            dv = [eng.to_device(wb["y"])] + [eng.to_device(wb[k]) for k in KEYS]
            fr = eng.forward(*dv, mvn_const="exact", store_ll_obs=False)
            call = lambda: eng.smooth_from_filter(dv[0], dv[3], dv[4], dv[6], fr.predicted_states, fr.predicted_covs)  # noqa: E731
            os.environ.pop("KF_SMOOTHER_BF_WG", None)
            t_wg = ms(call)
            os.environ["KF_SMOOTHER_BF_WG"] = "0"
            t_ws = ms(call)
            os.environ.pop("KF_SMOOTHER_BF_WG", None)
            t_rts = ms(lambda: eng.smooth(dv[3], dv[5], dv[7], fr.filtered_states, fr.filtered_covs), reps=1) if (r == m or m <= 32) else float("nan")
            gb = 2 * 8 * 1024 * 300 * (m + m * m) / 1e9
            line += f"   1024 x 300: on chip {t_wg:8.2f} ms ({gb / t_wg / 8:.3f} of 8 TB/s)   workspace {t_ws:8.2f} ms   kf_smoother (RTS) {t_rts:9.2f} ms"
            del fr, dv
        print(line, flush=True)
