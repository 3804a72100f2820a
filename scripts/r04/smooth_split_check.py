"""RTS smoother with the time-parallel kernels behind the filter's convergence point (kf_smoother_b4_steady.hip) against every step
in kf_smooth_b4_kernel (KF_SMOOTHER_SPLIT=0): time and largest deviation, headline / C3 shapes, tapes of the default forward and of
a steady-state forward."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cases = [("headline", lambda: W.structural(B=4096, n=1000), "standard"), ("headline/8", lambda: W.structural(B=512, n=1000), "standard"),
         ("c3", lambda: W.structural(B=8192, n=2000), "cholesky")]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
print("KF_SMOOTHER_SPLIT_TOL =", os.environ.get("KF_SMOOTHER_SPLIT_TOL"))
for name, mk, ft in cases:
    w = mk()
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    for fwd in ("default", "steady"):
        res = eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_predicted=False,
                          exact_recursion=None if fwd == "default" else False)
        args = (d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
        t_s = ms(lambda: eng.smooth(*args))
        sa, sP = eng.smooth(*args)
        os.environ["KF_SMOOTHER_SPLIT"] = "0"
        t_x = ms(lambda: eng.smooth(*args))
        xa, xP = eng.smooth(*args)
        del os.environ["KF_SMOOTHER_SPLIT"]
        torch.cuda.synchronize()
        B = sa.shape[0]
        ea = float(((sa - xa).abs().reshape(B, -1).max(dim=1).values / (1.0 + xa.abs().reshape(B, -1).max(dim=1).values)).max())
        eP = float(((sP - xP).abs().reshape(B, -1).max(dim=1).values / (1.0 + xP.abs().reshape(B, -1).max(dim=1).values)).max())
        print(f"{name:11s} forward tapes: {fwd:8s} smoother {t_s:7.3f} ms   every step in kf_smooth_b4_kernel {t_x:7.3f} ms   "
              f"largest deviation (rel. to the draw's largest entry): states {ea:.2e}  covariances {eP:.2e}", flush=True)
        del res, sa, sP, xa, xP
