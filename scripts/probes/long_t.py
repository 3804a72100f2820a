import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import c_oracle as co
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.structural(B=3, n=70000)
res = eng.forward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], mvn_const="exact")
ref = co.kalman_filter(w["y"], *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")], mvn_const="exact", full=True)
print("kernel", eng.forward_kernel_name(3, 70000, 1, 10, 4), "max|dlogp|", np.abs(res.logp.cpu().numpy() - ref["logp"]).max(), "|logp|", np.abs(ref["logp"]).max())
print("last pred cov diff", np.abs(res.predicted_covs[:, -1].cpu().numpy() - ref["predicted_covs"][:, -1]).max(),
      "ll_obs diff", np.abs(res.ll_obs.cpu().numpy() - ref["ll_obs"]).max())
g = eng.backward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"], predicted_states=res.predicted_states,
                 predicted_covs=res.predicted_covs, g_logp=np.ones(3))
print("grad finite", all(bool(np.isfinite(v.cpu().numpy()).all()) for v in g.values()))
sa, sP = eng.smooth(w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
print("smooth finite", bool(np.isfinite(sa.cpu().numpy()).all()), "last equal", bool((sa[:, -1] == res.filtered_states[:, -1]).all().item()))
