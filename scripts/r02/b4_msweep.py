import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
eng.set_forward_kernel(os.environ.get("KERN", "b4"))
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for m in (6, 8, 9, 10, 11, 12, 14):
    w = W.random_dense(4096, 1000, 1, m, 3, seed=m)
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    full = eng.forward(*dev, mvn_const="exact")
    lo = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
    tf = timeit(lambda: eng.forward(*dev, mvn_const="exact", out=full))
    tl = timeit(lambda: eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
    nbytes = 4096 * 8 * (1000 * m + 1001 * m + 1000 * m * m + 1001 * m * m + 1000)
    print(f"m={m:2d} full {tf:6.3f} ms  logp {tl:6.3f} ms  bytes {nbytes/1e9:5.2f} GB  -> {nbytes/tf/1e9:6.0f} GB/s   (full - logp = {tf-tl:5.3f} ms; bytes at 5.3 TB/s = {nbytes/5.3e9:5.3f} ms)")
    del full, lo, dev
