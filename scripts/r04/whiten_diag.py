import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
for shape in [(2, 4, 2), (5, 15, 5), (4, 20, 6)]:
    p, m, r = shape
    w = W.random_dense(B=24, n=90, p=p, m=m, r=r, seed=100 + p + m, h_diag=False)
    rng = np.random.default_rng(p * m)
    c, d = 0.05 * rng.standard_normal((24, m)), 0.1 * rng.standard_normal((24, p))
    for use_d in (False, True):
        for use_c in (False, True):
            for nan in (False, True):
                for ft in ("standard", "cholesky"):
                    y = w["y"].copy()
                    if nan: y[[4, 30, 31]] = np.nan
                    args = (y,) + tuple(w[k] for k in KEYS)
                    kw = dict(c=c if use_c else None, d=d if use_d else None)
                    got = eng.forward(*args, kw["c"], kw["d"], filter_type=ft, mvn_const="exact")
                    ref = kn.kalman_filter(*args, filter_type=ft, mvn_const="exact", **kw)
                    err = np.abs(got.logp.cpu().numpy() - ref["logp"]).max()
                    print(shape, "d" if use_d else "-", "c" if use_c else "-", "nan" if nan else "-", ft, eng.forward_kernel_name(24, 90, p, m, r, filter_type=ft), f"{err:.2e}")
