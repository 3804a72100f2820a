"""Where does the steady-state phase start (per draw)?  From the forward kernel's own output: the first step from which the
filtered covariance tape is an exact constant."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for tol in sys.argv[1:] or ["1e-13"]:
  os.environ["KF_SS_ENTRY_TOL"] = tol
  print("entrywise tolerance", tol)
  for name, B, kw in (("c4", 4096, dict(h_diagonal=True)), ("c2", 4096, {}), ("c3", 2048, {})):
      w = W.WORKLOADS[name](B=B)
      res = eng.forward(*[w[k] for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")], mvn_const="exact", exact_recursion=False,
                        store_ll_obs=False, store_predicted=False, **kw)
      fP = res.filtered_covs
      n = fP.shape[1]
      same = (fP == fP[:, -1:]).flatten(2).all(2)            # (B, n): equal to the last slot
      # first index from which all later slots are equal
      rev = torch.flip(same, dims=[1]).int().cumprod(1).sum(1)  # trailing run length
      ts = (n - rev).cpu().numpy()
      print(f"{name}: n = {n}; steady from step: median {np.median(ts):.0f} p90 {np.percentile(ts, 90):.0f} p99 {np.percentile(ts, 99):.0f} "
            f"max {ts.max()}; draws with ts > n/2: {(ts > n // 2).sum()}, never (ts >= n - 16): {(ts >= n - 16).sum()} of {B}")
      del res, fP
      torch.cuda.empty_cache()
