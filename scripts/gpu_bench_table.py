"""GPU-box: timing table over the BASELINE configs (forward logp-only / full tape, backward,
smoother) with roofline fractions.  Output: markdown table on stdout."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
keys = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


rows = []
CASES = [("c1", {}, "standard"), ("c2", {}, "standard"), ("headline", {}, "standard"),
         ("c3", {}, "cholesky"), ("c4_m10", {"B": 2048}, "standard"), ("c4", {"B": 2048}, "standard"),
         ("c4", {"B": 2048}, "univariate"), ("c4_m10", {"B": 16384}, "standard"),
         ("c4", {"B": 16384}, "standard"), ("c4", {"B": 16384}, "univariate")]
print("| workload | B | T | p,m,r | filter | kernel | fwd logp-only ms | fwd full-tape ms | evals/s (full) | HBM GB/s (full) | frac of 8 TB/s | fwd predicted-tape-only ms | KF_FLAG_EXACT_RECURSION: logp-only ; full-tape ms | bwd ms | logp + gradient, one call (kf_filter_value_and_grad) ms | smoother ms |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, kw, ft in CASES:
    w = W.WORKLOADS[name](**kw)
    B, n, p, m, r = w["B"], w["n"], w["p"], w["m"], w["r"]
    need = B * 8 * (2 * n * (m + m * m) + n) * 1.05
    if need > 200e9:
        continue
    d = [eng.to_device(w[k]) for k in keys]
    res_l = eng.forward(*d, filter_type=ft, store_ll_obs=False, store_filtered=False, store_predicted=False)
    t_logp = timeit(lambda: eng.forward(*d, filter_type=ft, out=res_l, store_ll_obs=False,
                                        store_filtered=False, store_predicted=False))
    res = eng.forward(*d, filter_type=ft)
    t_full = timeit(lambda: eng.forward(*d, filter_type=ft, out=res))
    gl = torch.ones(B, device="cuda", dtype=torch.float64)
    res_p = eng.forward(*d, filter_type=ft, store_ll_obs=False, store_filtered=False)
    t_pred = timeit(lambda: eng.forward(*d, filter_type=ft, out=res_p, store_ll_obs=False, store_filtered=False))
    del res_p
    t_logp_x = timeit(lambda: eng.forward(*d, filter_type=ft, out=res_l, store_ll_obs=False, store_filtered=False,
                                          store_predicted=False, exact_recursion=True))
    res_x = eng.forward(*d, filter_type=ft, exact_recursion=True)
    t_full_x = timeit(lambda: eng.forward(*d, filter_type=ft, out=res_x, exact_recursion=True))
    del res_x
    try:
        t_bwd = timeit(lambda: eng.backward(*d, predicted_states=res.predicted_states,
                                            predicted_covs=res.predicted_covs, g_logp=gl, filter_type=ft,
                                            h_diagonal=True), 2)
    except Exception as e:  # noqa: BLE001
        t_bwd = float("nan")
    try:
        t_vg = timeit(lambda: eng.value_and_grad(*d, filter_type=ft, h_diagonal=True), 2)
    except Exception as e:  # noqa: BLE001
        t_vg = float("nan")
    try:
        t_sm = timeit(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs), 2)
    except Exception as e:  # noqa: BLE001
        t_sm = float("nan")
    ab = algorithmic_bytes(B, n, p, m, r, True)
    gbs = ab / t_full / 1e9
    print(f"| {name} | {B} | {n} | {p},{m},{r} | {ft} | {eng.forward_kernel_name(B, n, p, m, r, ft, True)} | "
          f"{t_logp * 1e3:.3f} | {t_full * 1e3:.3f} | {B / t_full:.3e} | {gbs:.0f} | {gbs / 8000:.3f} | "
          f"{t_pred * 1e3:.3f} | {t_logp_x * 1e3:.3f} ; {t_full_x * 1e3:.3f} | {t_bwd * 1e3:.2f} | {t_vg * 1e3:.2f} | {t_sm * 1e3:.2f} |", flush=True)
    del d, res, res_l
    torch.cuda.empty_cache()
