import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for m in (21, 37, 50):
    n, B, p, r = 1, 3, 1, m
    w = W.random_dense(B, n, p, m, r, seed=1900 + n + m)
    rng = np.random.default_rng(n + m)
    yb = np.repeat(w["y"][None], B, axis=0); yb[:] = 0.3
    yb = yb + 0.1 * rng.standard_normal(yb.shape)
    args = (yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    ref = kn.kalman_filter(*args, mvn_const="exact")
    out = eng.forward(*args, mvn_const="exact")
    eng.set_forward_kernel("lds"); lds = eng.forward(*args, mvn_const="exact"); eng.set_forward_kernel("auto")
    F = np.einsum("bpi,bij,bqj->bpq", w["Z"], w["P0"], w["Z"])[:, 0, 0] + w["H"][:, 0, 0]
    v = yb[:, 0, 0] - np.einsum("bpi,bi->bp", w["Z"], w["a0"])[:, 0]
    print(m, "kernel", eng.forward_kernel_name(B, n, p, m, r), "ours", out.logp.cpu().numpy(), "lds", lds.logp.cpu().numpy(), "oracle", ref["logp"],
          "direct", -0.5 * (np.log(2 * np.pi) + np.log(F) + v * v / F), "P0 asym", np.abs(w["P0"] - np.swapaxes(w["P0"], -1, -2)).max())
