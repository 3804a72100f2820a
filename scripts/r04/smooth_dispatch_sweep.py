"""RTS smoother: AUTO against the Newton-Schulz block kernel forced on / off (KF_SMOOTHER_B4) over a grid of shapes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = 300
for m in (5, 6, 8, 10, 11, 12, 13, 15):
    for B in (512, 2048, 4096, 8192, 32768):
        w = W.random_dense(B, n, 1, m, 2, seed=m)
        d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
        res = eng.forward(*d, mvn_const="exact", store_ll_obs=False, store_predicted=False)
        args = (d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
        t = {}
        for tag, env in (("auto", None), ("ns", "1"), ("elim", "0")):
            if env is None:
                os.environ.pop("KF_SMOOTHER_B4", None)
            else:
                os.environ["KF_SMOOTHER_B4"] = env
            t[tag] = ms(lambda: eng.smooth(*args))
        os.environ.pop("KF_SMOOTHER_B4", None)
        best = min(t, key=t.get)
        flag = "" if t["auto"] <= 1.07 * t[best] else f"   <-- {best} is {t['auto'] / t[best]:.2f}x faster"
        print(f"m={m:2d} B={B:5d} x {n}: auto {t['auto']:7.3f}   Newton-Schulz block kernel {t['ns']:7.3f}   elimination kernel {t['elim']:7.3f}{flag}", flush=True)
        del res
