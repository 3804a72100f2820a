import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for B in (16384, 6144):
    w = W.varmax(B=B, n=1000)
    d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*d, mvn_const="exact", h_diagonal=True, exact_recursion=(os.environ.get("EXACT") == "1"))
    out = {}
    for env in ("0", "1"):
        os.environ["KF_SMOOTHER_B4"] = env
        sa, sP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
        t = timeit(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs))
        out[env] = (sa, sP)
        print(f"C4 B={B} KF_SMOOTHER_B4={env}: {t:.2f} ms", flush=True)
    print("   max diff states %.1e covs %.1e" % ((out["0"][0] - out["1"][0]).abs().max().item(), (out["0"][1] - out["1"][1]).abs().max().item()))
    os.environ.pop("KF_SMOOTHER_B4")
    del d, res, out, sa, sP
    torch.cuda.empty_cache()
