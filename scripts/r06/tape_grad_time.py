"""kf_filter_backward_tapes at the BASELINE shapes: time per call with all four tape cotangents (the LDS-resident adjoint, the only
kernel that carries them) beside kf_filter_backward on the same tapes (the shape's production adjoint)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


def ms(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run(name, w, ft="standard", **kw):
    dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    res = eng.forward(*dv, filter_type=ft, mvn_const="exact", **kw)
    B, n, m = w["B"], w["n"], w["m"]
    g = torch.Generator(device="cuda").manual_seed(1)
    cot = dict(g_filtered_states=torch.randn(B, n, m, device="cuda", dtype=torch.float64, generator=g),
               g_predicted_states=torch.randn(B, n + 1, m, device="cuda", dtype=torch.float64, generator=g),
               g_filtered_covs=torch.randn(B, n, m, m, device="cuda", dtype=torch.float64, generator=g),
               g_predicted_covs=torch.randn(B, n + 1, m, m, device="cuda", dtype=torch.float64, generator=g))
    ones = torch.ones(B, device="cuda", dtype=torch.float64)
    common = dict(predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, filter_type=ft, mvn_const="exact", **kw)
    t_t = ms(lambda: eng.backward(*dv, g_logp=ones, **cot, **common))
    t_l = ms(lambda: eng.backward(*dv, g_logp=ones, **common))
    gb = 8.0 * B * (2 * n * (m + m * m) * 2) / 1e9   # predicted tapes read + four cotangent tapes read
    print(f"{name:34s} tapes + logp {t_t:8.2f} ms ({gb / t_t:.2f} TB/s of tape reads)   logp only (production adjoint) {t_l:7.2f} ms", flush=True)


run("headline 4096 x 1000 m=10", W.structural(B=4096, n=1000))
run("C2 arma11 4096 x 1000 m=2", W.arma11(B=4096, n=1000))
run("C4 varmax(2,0) 2048 x 500 m=10 p=5", W.varmax(B=2048, n=500, order=(2, 0)), ft="univariate")
run("dense 1024 x 300 m=24 p=1", W.random_dense(1024, 300, 1, 24, 2, seed=3))
