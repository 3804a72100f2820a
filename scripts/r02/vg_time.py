"""logp + gradient in one call (kf_filter_value_and_grad) at the headline shape; KF_VG_TILE_TAPE=0 -> compact tapes."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name in ("headline", "c3"):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    ft = "cholesky" if name == "c3" else "standard"
    t = timeit(lambda: eng.value_and_grad(*dev, filter_type=ft))
    print(f"{name}: value_and_grad {t:.3f} ms   KF_VG_TILE_TAPE={os.environ.get('KF_VG_TILE_TAPE')}")
    del dev
    eng.release_workspace(); torch.cuda.empty_cache()
