"""GPU-box diagnostic: HIP RTS smoother vs the numpy oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


eng = KalmanEngine()
cases = [
    ("c1", W.nile_local_level(B=5)),
    ("c2", W.arma11(B=20, n=100)),
    ("headline", W.structural(B=21, n=100)),
    ("c4_m10", W.varmax(B=5, n=60, order=(2, 0))),
    ("c4_m15", W.varmax(B=5, n=40, order=(2, 1))),
    ("dense_p3m4", W.random_dense(9, 50, 3, 4, 2, seed=3)),
    ("dense_p1m7", W.random_dense(9, 50, 1, 7, 2, seed=9)),
    ("dense_p1m33", W.random_dense(3, 20, 1, 33, 2, seed=7)),
    ("reftest_m5", W.reference_test_inputs(1, 5, 1, 10)),
    ("nile_llt", W.nile_local_linear_trend(B=2)),
]
# exactly singular P_hat: two perfectly correlated states (duplicate of a random walk)
w = W.reference_test_inputs(1, 2, 1, 30)
w["T"] = np.array([[[1.0, 0.0], [1.0, 0.0]]])
w["R"] = np.array([[[1.0], [1.0]]])
w["P0"] = np.array([[[1.0, 1.0], [1.0, 1.0]]])
w["y"] = np.cumsum(np.random.default_rng(0).standard_normal((30, 1)), axis=0)
cases.append(("singular_m2", w))
for name, w in cases:
    args = (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    for lanes in (0, 16, 64):
        eng.set_lanes_per_draw(lanes)
        res = eng.forward(*args, mvn_const="exact")
        sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        torch.cuda.synchronize()
        fa, fP = res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy()
        na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], fa, fP)
        print(f"{name:12s} m{w['m']:<3d} G={lanes:<2d} sa {relerr(sa.cpu().numpy(), na):.1e} "
              f"sP {relerr(sP.cpu().numpy(), nP):.1e} nan={bool(torch.isnan(sP).any())}", flush=True)
eng.set_lanes_per_draw(0)
w = W.structural(B=4096, n=1000)
d = {k: eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")}
res = eng.forward(*[d[k] for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")])
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sa, sP = eng.smooth(d["T"], d["R"], d["Q"], res.filtered_states, res.filtered_covs)
    torch.cuda.synchronize()
    print(f"smoother headline B=4096 n=1000 m=10: {(time.perf_counter() - t0) * 1e3:.2f} ms")
