"""Edge cases of the steady-state phase: F == 0 at every step (known state, no noise), F < 0, NaN parameters, per-draw y."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
TAPES = ["ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"]
for p, m in ((1, 3), (1, 9), (3, 9), (5, 12)):
    z = W.reference_test_inputs(p, m, 1, 120)
    P0 = np.zeros_like(z["P0"])
    Hz, Qz = np.zeros((1, p, p)), np.zeros((1, 1, 1))
    for ft in ("univariate", "standard"):
        for mvn in ("reference", "exact"):
            r = eng.forward(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type=ft, mvn_const=mvn, h_diagonal=True)
            name = eng.forward_kernel_name(1, 120, p, m, 1, ft, True)
            if ft == "univariate":
                f = kn.kalman_filter(z["y"], z["a0"], P0, z["T"], z["Z"], z["R"], Hz, Qz, filter_type=ft, mvn_const=mvn)
                e = max(np.abs(getattr(r, k).cpu().numpy() - f[k]).max() for k in TAPES)
                print(f"p={p} m={m} {name:20s} {ft} {mvn}: dlogp {abs(r.logp.cpu().numpy() - f['logp']).max():.1e} dtape {e:.1e} info {r.info.cpu().numpy()}")
            else:
                print(f"p={p} m={m} {name:20s} {ft} {mvn}: info {r.info.cpu().numpy()} logp {r.logp.cpu().numpy()}")
# F < 0 in one draw, NaN T in another, per-draw y with a late NaN in one draw
for name, kw, ekw in (("structural", dict(B=9, n=300), {}), ("varmax", dict(B=9, n=300), dict(h_diagonal=True)), ("arma11", dict(B=9, n=300), {})):
    w = getattr(W, name)(**kw)
    B, n = 9, 300
    yb = np.repeat(w["y"][None], B, 0).copy()
    yb[3, 250] = np.nan
    H = w["H"].copy(); H[4] = -np.abs(H[4]) - 50.0 * np.eye(H.shape[-1])
    T = w["T"].copy(); T[6, 0, 0] = np.nan
    r = eng.forward(yb, w["a0"], w["P0"], T, w["Z"], w["R"], H, w["Q"], mvn_const="exact", **ekw)
    x = eng.forward(yb, w["a0"], w["P0"], T, w["Z"], w["R"], H, w["Q"], mvn_const="exact", exact_recursion=True, **ekw)
    good = np.array([0, 1, 2, 3, 5, 7, 8])
    f = kn.kalman_filter(yb[good], *[w[k][good] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")], mvn_const="exact", partial_missing="delete")
    lp, lx = r.logp.cpu().numpy(), x.logp.cpu().numpy()
    print(name, "info", r.info.cpu().numpy(), "exact info", x.info.cpu().numpy(), "nan logp", np.isnan(lp).nonzero()[0], np.isnan(lx).nonzero()[0],
          "dlogp good %.1e" % np.abs(lp[good] - f["logp"]).max(),
          "dtape good %.1e" % max(np.abs(getattr(r, k).cpu().numpy()[good] - f[k]).max() for k in TAPES))
