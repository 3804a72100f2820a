"""kf_smoother_b4.hip with the converged-filter shortcut (default) against every step in full (KF_SMOOTHER_STEADY=0) and the numpy
oracle: long dense series, a missing stretch in the middle, tapes of both forward modes; timing at the headline shape and C3."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def both(*args):
    out = {}
    for s in ("1", "0"):
        os.environ["KF_SMOOTHER_STEADY"] = s
        out[s] = eng.smooth(*args)
    os.environ["KF_SMOOTHER_STEADY"] = "1"
    return out
for m in (5, 8, 10, 11, 13, 15):
    w = W.random_dense(6, 500, 1, m, m, seed=160 + m)
    y = w["y"].copy(); y[240:247] = np.nan; y[400] = np.nan
    for exact in (True, False):
        res = eng.forward(y, *[w[k] for k in KEYS], exact_recursion=exact, store_ll_obs=False, store_predicted=False)
        fa, fP = res.filtered_states, res.filtered_covs
        ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], fa.cpu().numpy(), fP.cpu().numpy())
        o = both(w["T"], w["R"], w["Q"], fa, fP)
        sa, sP = o["1"]; ea_, eP_ = o["0"]
        ea = np.abs(sa.cpu().numpy() - ra).max() / (1 + np.abs(ra).max()); eP = np.abs(sP.cpu().numpy() - rP).max() / (1 + np.abs(rP).max())
        da = (sa - ea_).abs().max().item() / (1 + np.abs(ra).max()); dP = (sP - eP_).abs().max().item() / (1 + np.abs(rP).max())
        print(f"m={m:2d} exact_fwd={exact}: vs oracle a {ea:.1e} P {eP:.1e} | vs full steps a {da:.1e} P {dP:.1e}", flush=True)
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name in ("headline", "c3"):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    res = eng.forward(*dev, store_ll_obs=False, store_predicted=False, filter_type="cholesky" if name == "c3" else "standard")
    args = (dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
    o = both(*args)
    d = max((o["1"][0] - o["0"][0]).abs().max().item(), (o["1"][1] - o["0"][1]).abs().max().item())
    rel = ((o["1"][1] - o["0"][1]).abs() / (o["0"][1].abs() + 1e-300)).max().item()
    tt = {}
    for s in ("1", "0"):
        os.environ["KF_SMOOTHER_STEADY"] = s
        tt[s] = timeit(lambda: eng.smooth(*args))
    print(f"{name}: smoother {tt['1']:.3f} ms (every step in full: {tt['0']:.3f} ms); max abs diff {d:.2e}, max entrywise rel diff of P {rel:.2e}", flush=True)
    del dev, res, o, args
    torch.cuda.empty_cache()
