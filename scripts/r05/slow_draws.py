"""Development aid (CPU): which draws of the headline batch never converge, and what the Newton-Schulz smoother does on them."""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
sys.path.insert(0, "scripts/r05")
from ns_extrap_proto import run

w = W.WORKLOADS["headline"](B=4096)
flt = kn.kalman_filter(w["y"], *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
fP = flt["filtered_covs"]
d = np.abs(fP[:, 1:] - fP[:, :-1]).reshape(4096, 999, -1).max(-1) / np.abs(fP[:, 1:]).reshape(4096, 999, -1).max(-1)
ts = np.array([np.max(np.nonzero(d[b] > 1e-15)[0], initial=0) for b in range(4096)])
print("ts quantiles", np.quantile(ts, [0.1, 0.5, 0.9, 0.95, 0.99]), "never (>=990):", (ts >= 990).sum())
slow = np.nonzero(ts >= 990)[0][:8]
np.save("/tmp/slow_idx.npy", np.nonzero(ts >= 990)[0])
for b in slow:
    print(f"draw {b}: rel change per step at t=100,500,900: {d[b,100]:.1e} {d[b,500]:.1e} {d[b,900]:.1e}  diagQ {np.diag(w['Q'][b])}  H {w['H'][b].ravel()}")
    line = ""
    for order in (0, 1, 2):
        its, pr = run(w, b, order)
        line += f"  order {order}: it/step {its.mean():.2f} prod/step {pr.mean():.2f} hist {np.bincount(its)[:6]}"
    print(line, flush=True)
