"""How soon does the predicted covariance reach a bitwise fixpoint? (per workload)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for name, kw in (("headline", dict(B=512)), ("c2", dict(B=512)), ("c4", dict(B=256)), ("hourly", dict(B=128))):
    w = W.WORKLOADS[name](**kw)
    d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*d, h_diagonal=True)
    P = res.predicted_covs  # (B, n+1, m, m)
    same = (P[:, 1:] == P[:, :-1]).flatten(2).all(dim=2)  # (B, n): P[t+1] == P[t]
    n = same.shape[1]
    # first t after which it stays equal to the end
    rev = torch.flip(same, dims=[1]).int().cumprod(dim=1).sum(dim=1)  # length of the trailing run
    first = n - rev
    # also a 2-cycle check
    same2 = (P[:, 2:] == P[:, :-2]).flatten(2).all(dim=2)
    rev2 = torch.flip(same2, dims=[1]).int().cumprod(dim=1).sum(dim=1)
    nan_rows = int(np.isnan(w["y"]).any(axis=-1).sum())
    q = torch.quantile(first.double(), torch.tensor([0.0, 0.5, 0.9, 1.0], dtype=torch.float64, device=first.device))
    print(f"{name}: n={n} m={w['m']} missing rows {nan_rows}; step from which P_pred is bitwise constant: min/median/p90/max = {q.cpu().numpy()}; "
          f"draws never constant: {int((rev == 0).sum())}/{len(rev)}; 2-cycle tail median {float(torch.median((n - 1 - rev2).double())):.0f}")
