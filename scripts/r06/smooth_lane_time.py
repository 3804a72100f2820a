"""kf_smoother_from_filter for m <= 4: the lane-per-draw kernel (kf_smoother_bf_lane.hip) against round 5's block-tile kernel
(KF_SMOOTHER_BF_LANE=0) and the RTS lane kernel (kf_smoother) at C2 and neighbouring shapes: time per call."""
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


def ms(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cases = [("c2 4096x1000 m=2", lambda: W.arma11(B=4096, n=1000)), ("c2/8 512x1000 m=2", lambda: W.arma11(B=512, n=1000)),
         ("arma 16384x1000 m=2", lambda: W.arma11(B=16384, n=1000)), ("arma 65536x500 m=2", lambda: W.arma11(B=65536, n=500)),
         ("local level 4096x1000 m=1", lambda: W.random_dense(4096, 1000, 1, 1, 1, seed=1)),
         ("dense 4096x1000 m=3", lambda: W.random_dense(4096, 1000, 1, 3, 2, seed=3)),
         ("dense 4096x1000 m=4", lambda: W.random_dense(4096, 1000, 1, 4, 2, seed=4))]
for name, mk in cases:
    w = mk()
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    res = eng.forward(*d, mvn_const="exact", store_ll_obs=False)
    call = lambda: eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)  # noqa: E731
    os.environ.pop("KF_SMOOTHER_BF_LANE", None)
    t_lane = ms(call)
    os.environ["KF_SMOOTHER_BF_LANE"] = "0"
    t_tile = ms(call)
    os.environ.pop("KF_SMOOTHER_BF_LANE", None)
    t_rts = ms(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs))
    sa = res.filtered_states
    gb = 2 * sa.numel() * 8 / 1e9 + 2 * res.filtered_covs.numel() * 8 / 1e9
    print(f"{name:28s} lane {t_lane:7.3f} ms ({gb / t_lane / 8:.3f} of 8 TB/s)   tile kernel {t_tile:7.3f} ms   "
          f"kf_smoother (RTS lane) {t_rts:7.3f} ms", flush=True)
    del res
