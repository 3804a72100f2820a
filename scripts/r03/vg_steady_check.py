"""kf_filter_value_and_grad with the steady-state adjoint (C4 shapes) against torch autograd of the recursion at every step
(oracle/kalman_torch.py) and against the exact adjoint (KF_VG_STEADY=0); timing at the C4 size."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
KEYS = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
NAMES = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")


def ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, w, ft, k=6, cd=True):
    B, m, p = w["B"], w["m"], w["p"]
    rng = np.random.default_rng(5)
    c = 0.05 * rng.standard_normal((B, m)) if cd else None
    d = 0.1 * rng.standard_normal((B, p)) if cd else None
    gl = 1.0 + 0.3 * rng.standard_normal(B)
    dev = [eng.to_device(w[x]) for x in KEYS]
    kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact", h_diagonal=True)
    lp, info, g = eng.value_and_grad(*dev, **kw)
    os.environ["KF_VG_STEADY"] = "0"
    lpx, _, gx = eng.value_and_grad(*dev, **kw)
    os.environ.pop("KF_VG_STEADY")
    rlp, _, rg = kt.logp_and_grads(w["y"], *[w[x][:k] for x in KEYS[1:]], c=None if c is None else c[:k],
                                   d=None if d is None else d[:k], filter_type=ft, mvn_const="exact", g_logp=gl[:k])
    out = [f"{name} {ft}: dlogp vs oracle {np.abs(lp[:k].cpu().numpy() - rlp).max():.1e}"]
    for nm in NAMES:
        ref = rg[nm]
        if nm == "H":
            ref = ref * np.eye(p)
        sc = max(np.abs(ref).max(), 1e-300)
        e1 = np.abs(g[nm][:k].cpu().numpy().reshape(ref.shape) - ref).max() / sc
        e2 = (g[nm] - gx[nm]).abs().max().item() / max(gx[nm].abs().max().item(), 1e-300)
        out.append(f"{nm} {e1:.0e}/{e2:.0e}")
    print(" ".join(out) + "   (steady vs autograd / steady vs exact adjoint, relative to the largest entry)", flush=True)


if __name__ == "__main__":
    for ft in ("standard", "univariate", "cholesky"):
        check("varmax m=15 p=5", W.varmax(B=24, n=300), ft)
    check("varmax m=10 p=5 (2,0)", W.varmax(B=24, n=300, order=(2, 0)), "standard")
    check("varmax k=3 m=9", W.varmax(B=24, n=250, k=3, order=(2, 1)), "standard", cd=False)
    w = W.varmax(B=12, n=260)
    y = w["y"].copy(); y[[3, 100]] = np.nan; y[150, 2] = np.nan
    check("varmax + missing rows / cells", dict(w, y=y), "standard")
    w = W.WORKLOADS["c4"]()
    dev = [eng.to_device(w[x]) for x in KEYS]
    kw = dict(filter_type="standard", mvn_const="exact", h_diagonal=True)
    t_s = ms(lambda: eng.value_and_grad(*dev, **kw))
    os.environ["KF_VG_STEADY"] = "0"
    t_x = ms(lambda: eng.value_and_grad(*dev, **kw), 3)
    os.environ.pop("KF_VG_STEADY")
    print(f"c4 (16384 x 1000, m = 15, p = 5) logp + gradient: steady-state adjoint {t_s:.2f} ms, exact adjoint at every step {t_x:.2f} ms")
    w = W.WORKLOADS["c4"](B=2048)
    dev = [eng.to_device(w[x]) for x in KEYS]
    t_s = ms(lambda: eng.value_and_grad(*dev, **kw))
    os.environ["KF_VG_STEADY"] = "0"
    t_x = ms(lambda: eng.value_and_grad(*dev, **kw), 3)
    print(f"c4 shard (2048 draws): {t_s:.2f} ms against {t_x:.2f} ms")
