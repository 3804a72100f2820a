"""RTS smoother (kf_smoother), 5 <= m <= 15: the Newton-Schulz block kernel against the elimination kernel over the noise rank r and the
series length n (AUTO lost 1.9x at m = 10, 4096 x 300, r = 2)."""
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


def run(tag, w):
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    res = eng.forward(*d, mvn_const="exact", store_ll_obs=False, store_predicted=False)
    args = (d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
    t = {}
    for k, env in (("auto", None), ("ns", "1"), ("elim", "0")):
        if env is None:
            os.environ.pop("KF_SMOOTHER_B4", None)
        else:
            os.environ["KF_SMOOTHER_B4"] = env
        t[k] = ms(lambda: eng.smooth(*args))
    os.environ.pop("KF_SMOOTHER_B4", None)
    sff = None
    if 5 <= w["m"] <= 15:   # the same moments from the predicted tapes (kf_smoother_from_filter: no inverse)
        res2 = eng.forward(*d, mvn_const="exact", store_ll_obs=False, store_filtered=False)
        sff = ms(lambda: eng.smooth_from_filter(d[0], d[3], d[4], d[6], res2.predicted_states, res2.predicted_covs))
        del res2
    best = min(t, key=t.get)
    flag = "" if t["auto"] <= 1.07 * t[best] else f"   <-- {best} is {t['auto'] / t[best]:.2f}x faster"
    print(f"{tag}: auto {t['auto']:7.3f}   Newton-Schulz {t['ns']:7.3f}   elimination {t['elim']:7.3f}   from the filter's tapes "
          f"{'   -   ' if sff is None else format(sff, '7.3f')}{flag}", flush=True)


Bs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096"])]
for B in Bs:
    for n in (300, 1000):
        run(f"structural m=10 r=4 B={B} x {n}", W.structural(B=B, n=n))
        for m in (6, 8, 10, 12, 15):
            for r in (1, 2, 4, m):
                run(f"dense m={m:2d} r={r:2d} B={B} x {n}", W.random_dense(B, n, 1, m, r, seed=m + r))
