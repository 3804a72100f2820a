"""The chunked first 256 steps of the lane-per-draw forward kernel (kf_forward_lane.hip, KF_LANE_PRECHUNK) against the same kernel
stepping from step 0: largest deviation of every output (scaled), then the time per call at C2."""
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
OUT = ("logp", "ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs")
eng = KalmanEngine()


def ms(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, w, y=None, timing=False, expect_equal=False, **kw):
    dv = [eng.to_device(w["y"] if y is None else y)] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(mvn_const="exact", **kw)
    eng.set_option("KF_LANE_PRECHUNK", "0")
    ref = eng.forward(*dv, **kw)
    eng.set_option("KF_LANE_PRECHUNK", None)
    got = eng.forward(*dv, **kw)
    torch.cuda.synchronize()
    name_k = eng.forward_kernel_name(w["B"], w["n"], 1, w["m"], w["r"])
    cs = float(ref.predicted_covs.abs().max())
    dev = {}
    for k in OUT:
        a, b = getattr(got, k), getattr(ref, k)
        sc = cs if k.endswith("covs") else float(torch.nan_to_num(b).abs().max())
        dev[k] = float(torch.nan_to_num(a - b).abs().max() / (1e-300 + sc))
    same = all(torch.equal(torch.nan_to_num(getattr(got, k), nan=-7.0), torch.nan_to_num(getattr(ref, k), nan=-7.0)) for k in OUT)
    line = f"{name:44s} {name_k:20s} " + ("bitwise equal" if same else "max scaled deviation %.1e" % max(dev.values())) + f"  info {int(got.info.abs().max())}/{int(ref.info.abs().max())}"
    if timing:
        t1 = ms(lambda: eng.forward(*dv, **kw))
        t1l = ms(lambda: eng.forward(*dv, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
        eng.set_option("KF_LANE_PRECHUNK", "0")
        t0 = ms(lambda: eng.forward(*dv, **kw))
        t0l = ms(lambda: eng.forward(*dv, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
        eng.set_option("KF_LANE_PRECHUNK", None)
        line += f"   full tapes {t1:.3f} ms (from step 0: {t0:.3f})   logp only {t1l:.3f} ms ({t0l:.3f})"
    print(line, flush=True)
    bad = (expect_equal and not same) or max(dev.values()) > 1e-10
    return not bad


ok = True
ok &= check("C2 arma11 4096 x 1000", W.arma11(B=4096, n=1000), timing=True)
ok &= check("arma11 1000 x 600, cholesky", W.arma11(B=1000, n=600), filter_type="cholesky")
ok &= check("arma11 777 x 2000, univariate", W.arma11(B=777, n=2000), filter_type="univariate", timing=True)
rng = np.random.default_rng(2)
w1 = W.random_dense(513, 700, 1, 1, 1, seed=3)
ok &= check("dense m=1 513 x 700", w1)
ok &= check("same, c and d, single", w1, c=0.2 * rng.standard_normal((513, 1)), d=0.3 * rng.standard_normal((513, 1)), filter_type="single")
w2 = W.random_dense(300, 900, 1, 2, 2, seed=4)
ok &= check("dense m=2 r=2 300 x 900", w2)
ok &= check("same, H = 0", W.random_dense(300, 900, 1, 2, 2, seed=4, h_zero=True))
ym = w2["y"].copy()
ym[100] = np.nan
ok &= check("same, a missing observation in the first 256", w2, y=ym, expect_equal=True)
ym = w2["y"].copy()
ym[700] = np.nan
ok &= check("same, a missing observation behind them", w2, y=ym)
yb = w2["y"][None] + 0.05 * rng.standard_normal((300,) + w2["y"].shape)
ok &= check("same, one series per draw", w2, y=yb)
ok &= check("arma11 64 x 512 (shortest taken)", W.arma11(B=64, n=512))
ok &= check("arma11 64 x 511 (not taken)", W.arma11(B=64, n=511), expect_equal=True)
wn = W.nile_local_linear_trend(B=8)
wn["y"] = np.tile(wn["y"], (6, 1))
wn["n"] = 600
ok &= check("nile LLT x6 (P0 = 1e6 I)", wn)
print("ALL WITHIN 1e-10" if ok else "FAILED")
sys.exit(0 if ok else 1)
