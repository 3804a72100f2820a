"""Lane smoother (m <= 4, pinv in registers) against the numpy oracle (numpy.linalg.pinv per step) and the LDS kernel; timing on C2."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()


def ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, w, k=64):
    res = eng.forward(*[w[x] for x in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")], mvn_const="exact", exact_recursion=True)
    fa, fP = res.filtered_states, res.filtered_covs
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], fa, fP)
    os.environ["KF_SMOOTHER_LANE"] = "0"
    la, lP = eng.smooth(w["T"], w["R"], w["Q"], fa, fP)
    os.environ.pop("KF_SMOOTHER_LANE")
    k = min(k, w["B"])
    ra, rP = kn.kalman_smoother(w["T"][:k], w["R"][:k], w["Q"][:k], fa[:k].cpu().numpy(), fP[:k].cpu().numpy())
    print(f"{name}: vs oracle states {np.abs(sa[:k].cpu().numpy() - ra).max():.2e} covs {np.abs(sP[:k].cpu().numpy() - rP).max():.2e} | "
          f"LDS kernel vs oracle {np.abs(la[:k].cpu().numpy() - ra).max():.2e} {np.abs(lP[:k].cpu().numpy() - rP).max():.2e} | "
          f"lane vs LDS {(sa - la).abs().max().item():.2e} {(sP - lP).abs().max().item():.2e}", flush=True)
    return fa, fP


for m in (1, 2, 3, 4):
    w = W.random_dense(50, 200, 1, m, min(m, 2), seed=40 + m)
    check(f"dense m={m}", w)
    w = W.random_dense(50, 200, 1, m, 1, seed=60 + m, h_zero=True)   # rank-deficient P_hat (r = 1 < m, no measurement noise)
    check(f"dense m={m} r=1 H=0", w)
w = W.WORKLOADS["c2"]()
fa, fP = check("c2", w, k=128)
T, R, Q = [eng.to_device(w[k]) for k in ("T", "R", "Q")]
t_lane = ms(lambda: eng.smooth(T, R, Q, fa, fP))
os.environ["KF_SMOOTHER_LANE"] = "0"
t_lds = ms(lambda: eng.smooth(T, R, Q, fa, fP))
print(f"c2 smoother: lane kernel {t_lane:.3f} ms, LDS kernel {t_lds:.3f} ms")
