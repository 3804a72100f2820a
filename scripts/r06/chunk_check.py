"""The time-chunked forward pipeline (kf_forward_b4c.hip) against the sequential kernel (KF_FLAG_EXACT_RECURSION) on the same inputs:
largest deviation of every output, and time per call, full tapes and log-likelihood only."""
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


def ms(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def rel(x, r):
    return float((x - r).abs().max() / (1e-300 + r.abs().max()))


def check(name, w, kernel=None, c=None, d=None, ft="standard", y=None, timing=True):
    if kernel:
        eng.set_forward_kernel(kernel)
    dv = [eng.to_device(w["y"] if y is None else y)] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(c=c, d=d, filter_type=ft, mvn_const="exact")
    ref = eng.forward(*dv, exact_recursion=True, **kw)
    os.environ.pop("KF_FORWARD_CHUNKED", None)
    got = eng.forward(*dv, **kw)
    torch.cuda.synchronize()
    dev = {k: rel(getattr(got, k), getattr(ref, k)) for k in ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs")}
    dl = float((got.logp - ref.logp).abs().max())
    info = int(got.info.abs().max())
    line = f"{name:34s} max|dlogp| {dl:.2e}  " + "  ".join(f"{k.split('_')[0][:4]}{k.split('_')[-1][:3]} {v:.1e}" for k, v in dev.items()) + f"  info {info}"
    if timing:
        t_c = ms(lambda: eng.forward(*dv, **kw))
        t_cl = ms(lambda: eng.forward(*dv, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
        os.environ["KF_FORWARD_CHUNKED"] = "0"
        t_s = ms(lambda: eng.forward(*dv, **kw))
        t_sl = ms(lambda: eng.forward(*dv, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
        os.environ.pop("KF_FORWARD_CHUNKED", None)
        line += f"   full tapes {t_c:.3f} ms (sequential {t_s:.3f})   logp only {t_cl:.3f} ms (sequential {t_sl:.3f})"
    print(line, flush=True)
    if kernel:
        eng.set_forward_kernel("auto")


check("headline/8  512 x 1000 m=10", W.structural(B=512, n=1000))
check("headline/16 256 x 1000 m=10", W.structural(B=256, n=1000))
check("1024 x 1000 m=10", W.structural(B=1024, n=1000))
check("c3/8 1024 x 2000 m=10 cholesky", W.structural(B=1024, n=2000), ft="cholesky")
check("64 x 300 m=10", W.structural(B=64, n=300))
rng = np.random.default_rng(3)
w6 = W.random_dense(100, 500, 1, 6, 2, seed=5)
check("dense 100 x 500 m=6 r=2, c, d", w6, c=0.2 * rng.standard_normal((100, 6)), d=0.3 * rng.standard_normal((100, 1)))
check("same, SingleTimeseriesFilter", w6, c=0.2 * rng.standard_normal((100, 6)), d=0.3 * rng.standard_normal((100, 1)), ft="single", timing=False)
check("dense 37 x 777 m=9 r=1", W.random_dense(37, 777, 1, 9, 1, seed=6), timing=False)
check("arma11 512 x 1000 on the block kernel", W.arma11(B=512, n=1000), kernel="b4")
wn = W.nile_local_linear_trend(B=16)
wn["y"] = np.tile(wn["y"], (4, 1))
check("nile LLT x4 (P0 = 1e6 I), b4", wn, kernel="b4", timing=False)
ws = W.structural(B=128, n=600)
ym = ws["y"].copy()
ym[[5, 300]] = np.nan
check("missing observations -> sequential", ws, y=ym, timing=False)
yb = ws["y"][None] + 0.05 * rng.standard_normal((128,) + ws["y"].shape)
check("one series per draw", ws, y=yb, timing=False)
