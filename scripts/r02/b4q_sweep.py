"""AUTO rule for kf_forward_b4q.hip: multi-observation block kernel against what AUTO picked before (KF_FORWARD_B4Q=0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
def args(w):
    return (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
n = 500
for p, m, r in [(2, 2, 1), (2, 4, 2), (3, 6, 3), (2, 8, 3), (4, 8, 4), (5, 10, 5), (2, 11, 4), (3, 13, 4), (8, 12, 6), (5, 15, 5)]:
    for B in (512, 4096, 16384):
        w = W.random_dense(B, n, p, m, r, seed=3)
        dev = [eng.to_device(x) for x in args(w)]
        row = []
        for env in ("0", "1"):
            os.environ["KF_FORWARD_B4Q"] = env
            name = eng.forward_kernel_name(B, n, p, m, r, "standard", True)
            full = eng.forward(*dev, mvn_const="exact", h_diagonal=True)
            t_full = timeit(lambda: eng.forward(*dev, mvn_const="exact", h_diagonal=True, out=full))
            lo = eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False, store_predicted=False)
            t_lo = timeit(lambda: eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
            row.append((name, t_full, t_lo, full.logp.clone()))
        dl = (row[0][3] - row[1][3]).abs().max().item()
        print(f"p={p} m={m:2d} B={B:6d}  {row[0][0]:26s} full {row[0][1]:6.2f} logp {row[0][2]:6.2f} | b4q full {row[1][1]:6.2f} logp {row[1][2]:6.2f}  ratio {row[0][1]/row[1][1]:.2f} / {row[0][2]/row[1][2]:.2f}  dlogp {dl:.1e}", flush=True)
os.environ.pop("KF_FORWARD_B4Q")
