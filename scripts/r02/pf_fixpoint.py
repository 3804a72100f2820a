"""How often do consecutive filtered covariances agree BIT FOR BIT (headline / C3 tapes from the forward kernel)?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for name, ft in (("headline", "standard"), ("c3", "cholesky")):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev, filter_type=ft, store_ll_obs=False, store_predicted=False)
    fP = res.filtered_covs
    same = (fP[:, 1:] == fP[:, :-1]).flatten(2).all(dim=2)          # (B, n-1)
    n = same.shape[1]
    # first step from which the covariance never changes again
    chg = (~same).int()
    last_change = torch.where(chg.any(1), n - 1 - torch.flip(chg, dims=[1]).argmax(1), torch.zeros_like(chg[:, 0]))
    lc = last_change.cpu().numpy()
    print(f"{name}: steps with P_f[t+1] == P_f[t] bitwise: mean {same.float().mean().item():.3f} of {n}; "
          f"last change at step: median {np.median(lc):.0f} p90 {np.percentile(lc, 90):.0f} p99 {np.percentile(lc, 99):.0f} max {lc.max()}; "
          f"draws that still change in the last 10 steps: {(lc >= n - 10).sum()}")
    del dev, res, fP, same
    torch.cuda.empty_cache()
