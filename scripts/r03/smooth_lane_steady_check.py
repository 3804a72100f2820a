"""kf_smoother_lane.hip (m <= 2) with the wave-uniform reuse of the last gain against every step in full and the oracle; C2 timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, m in (("c2", 2), ("dense2", 2), ("dense1", 1)):
    w = W.WORKLOADS["c2"]() if name == "c2" else W.random_dense(4096, 1000, 1, m, m, seed=7)
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    res = eng.forward(*dev, store_ll_obs=False, store_predicted=False)
    args = (dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
    sa, sP = eng.smooth(*args)
    ea, eP = eng.smooth(*args, exact_recursion=True)
    idx = [0, 1000, 4095]
    ra, rP = kn.kalman_smoother(w["T"][idx], w["R"][idx], w["Q"][idx], res.filtered_states[idx].cpu().numpy(), res.filtered_covs[idx].cpu().numpy())
    sc = 1 + max(np.abs(ra).max(), np.abs(rP).max())
    eo = max(np.abs(sa[idx].cpu().numpy() - ra).max(), np.abs(sP[idx].cpu().numpy() - rP).max()) / sc
    df = max((sa - ea).abs().max().item(), (sP - eP).abs().max().item()) / sc
    t1 = timeit(lambda: eng.smooth(*args)); t0 = timeit(lambda: eng.smooth(*args, exact_recursion=True))
    print(f"{name}: smoother {t1:.3f} ms (every step in full {t0:.3f} ms); vs oracle (3 draws) {eo:.1e}; vs full steps {df:.1e}", flush=True)
