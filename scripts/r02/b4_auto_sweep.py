"""p = 1, small m, large batches: what AUTO picks (row / lane kernels) against the block kernel with its steady-state phase."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
def args(w): return (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
n = 500
import os
EX = os.environ.get("EXACT") == "1"
for m in (5, 6, 7, 8, 9):
    for B in (16384, 65536):
        w = W.random_dense(B, n, 1, m, min(m, 3), seed=m)
        dev = [eng.to_device(x) for x in args(w)]
        row = f"m={m} B={B:6d} T={n}:"
        for kern in ("auto", "b4"):
            eng.set_forward_kernel(kern)
            name = eng.forward_kernel_name(B, n, 1, m, min(m, 3))
            full = eng.forward(*dev, mvn_const="exact", exact_recursion=EX)
            t_full = timeit(lambda: eng.forward(*dev, mvn_const="exact", exact_recursion=EX, out=full))
            lo = eng.forward(*dev, mvn_const="exact", exact_recursion=EX, store_ll_obs=False, store_filtered=False, store_predicted=False)
            t_lo = timeit(lambda: eng.forward(*dev, mvn_const="exact", exact_recursion=EX, store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
            row += f"  {name:22s} full {t_full:6.2f} logp {t_lo:6.2f}"
            del full, lo
        eng.set_forward_kernel("auto")
        print(row, flush=True)
        del dev
