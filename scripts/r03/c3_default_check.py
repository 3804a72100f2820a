import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS["c3"]()
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = eng.forward(*dev, filter_type="cholesky", mvn_const="exact")
for rep in range(3):
    for name, kw in (("default", {}), ("steady flag", dict(exact_recursion=False)), ("exact", dict(exact_recursion=True))):
        print(rep, name, round(t(lambda: eng.forward(*dev, filter_type="cholesky", mvn_const="exact", out=out, **kw)), 3), flush=True)
