"""C5 share gradient (kf_backward_big.hip), 128 draws x 40 steps: ms per step of the adjoint sweep, both product kernels."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.daily_seasonal(B=128, n=40)
KEYS = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
dev = [eng.to_device(w[k]) for k in KEYS]
res = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False)
g0 = None
for flag in ("1", "0"):
    os.environ["KF_SMOOTHER_BIG_GEMM"] = flag
    kw = dict(predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=np.ones(128), mvn_const="exact")
    g = eng.backward(*dev, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); g = eng.backward(*dev, **kw); torch.cuda.synchronize()
    print(f"KF_SMOOTHER_BIG_GEMM={flag}: gradient sweep {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per step", flush=True)
    if g0 is None: g0 = g
    else: print("max rel diff of g_T between the two:", float(((g["T"] - g0["T"]).abs().max() / g0["T"].abs().max())))
