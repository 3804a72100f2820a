"""auto vs b4 forward kernel, p = 1, m = 1..14, 4096 draws x 1000 steps (and 512 draws: one GPU's share of a strong-scaling run)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for B in (4096, 512, 32768):
    for m in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14):
        if B == 32768 and m not in (2, 4, 6, 8, 10): continue
        w = W.random_dense(B, 1000 if B <= 4096 else 200, 1, m, min(m, 3), seed=m)
        dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
        row = f"B={B:5d} m={m:2d}"
        for kern in ("auto", "b4"):
            eng.set_forward_kernel(kern)
            full = eng.forward(*dev, mvn_const="exact")
            lo = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
            tf = timeit(lambda: eng.forward(*dev, mvn_const="exact", out=full))
            tl = timeit(lambda: eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
            name = eng.forward_kernel_name(B, w["n"], 1, m, w["r"])
            row += f" | {kern:4s} {name[7:-7]:12s} full {tf:7.3f} logp {tl:7.3f}"
            del full, lo
        eng.set_forward_kernel("auto")
        print(row, flush=True)
        del dev
