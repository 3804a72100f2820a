"""b4 smoother (kf_smoother_b4.hip) vs the numpy oracle + timing at the headline shape."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
worst = 0.0
for m in (5, 6, 7, 8, 9, 10, 11, 12, 13, 15):
    w = W.random_dense(4, 90, 1, m, m, seed=60 + m)
    flt = kn.kalman_filter(w["y"], *[w[k] for k in KEYS])
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    ea = np.abs(sa.cpu().numpy() - ra).max() / (1 + np.abs(ra).max()); eP = np.abs(sP.cpu().numpy() - rP).max() / (1 + np.abs(rP).max())
    worst = max(worst, ea, eP)
    print(f"m={m:2d}: err a {ea:.1e} P {eP:.1e}", flush=True)
print("worst", worst)
for name in ("headline", "c3"):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    res = eng.forward(*dev, store_ll_obs=False, store_predicted=False, filter_type="cholesky" if name == "c3" else "standard")
    def timeit(fn, n=4):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    ts = timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs))
    sa, sP = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
    idx = [0, 1000, 4095]
    fa, fP = res.filtered_states[idx].cpu().numpy(), res.filtered_covs[idx].cpu().numpy()
    ra, rP = kn.kalman_smoother(w["T"][idx], w["R"][idx], w["Q"][idx], fa, fP)
    e = max(np.abs(sa[idx].cpu().numpy() - ra).max(), np.abs(sP[idx].cpu().numpy() - rP).max())
    print(f"{name}: smoother {ts:.3f} ms  max err (3 draws) {e:.1e}  KF_SMOOTHER_B4={os.environ.get('KF_SMOOTHER_B4')}")
    del dev, res, sa, sP
    torch.cuda.empty_cache()
