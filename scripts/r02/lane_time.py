"""lane kernel (kf_forward_lane.hip) vs b4 for m <= 4, p = 1: full tapes / logp only, ms per launch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def bench(tag, w):
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    row = tag
    for kern in ("lane", "b4"):
        eng.set_forward_kernel(kern)
        full = eng.forward(*dev, mvn_const="exact")
        lo = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
        tf = timeit(lambda: eng.forward(*dev, mvn_const="exact", out=full))
        tl = timeit(lambda: eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
        row += f" | {kern:4s} full {tf:7.3f} logp {tl:7.3f}"
    eng.set_forward_kernel("auto")
    print(row, flush=True)
bench("c2 (arma11 m=2 B=4096 T=1000)", W.WORKLOADS["c2"]())
for B in (512, 2048, 4096, 16384, 65536):
    for m in (1, 2, 3, 4):
        bench(f"B={B:6d} m={m}", W.random_dense(B, 1000 if B <= 4096 else 250, 1, m, min(m, 2), seed=m))
