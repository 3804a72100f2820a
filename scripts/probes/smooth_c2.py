import sys
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.arma11(B=64, n=200)
flt = kn.kalman_filter(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
print("oracle max |P_s|", np.abs(rP).max(), "max |a_s|", np.abs(ra).max())
for kern in ("lds", "auto"):
    eng.set_forward_kernel(kern)
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa, sP = sa.cpu().numpy(), sP.cpu().numpy()
    eP = np.abs(sP - rP).max(axis=(1, 2, 3)); ea = np.abs(sa - ra).max(axis=(1, 2))
    print(kern, "max err P", eP.max(), "a", ea.max(), "worst draws", np.argsort(eP)[-3:], eP[np.argsort(eP)[-3:]])
    if kern == "auto":
        ws = eng._ws
        flags = ws[64 - (ws.data_ptr() % 64) if ws.data_ptr() % 64 else 0:][:64 * 4].view(torch.int32).cpu().numpy()
        print("flagged draws:", int(flags.sum()), "of 64")
        bi = int(np.argmax(eP))
        T = w["T"][bi]; R = w["R"][bi]; Q = w["Q"][bi]
        for t in (198, 100, 0):
            Ph = T @ flt["filtered_covs"][bi, t] @ T.T + R @ Q @ R.T
            print("  draw", bi, "t", t, "P_hat", Ph.ravel(), "eig", np.linalg.eigvalsh(0.5 * (Ph + Ph.T)), "flag", flags[bi])
eng.set_forward_kernel("auto")
