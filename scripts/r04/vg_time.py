"""logp + gradient in one call (kf_filter_value_and_grad) at the headline / C3 shapes: the steady-state adjoint behind the block
kernel (default) against the exact adjoint at every step, with the per-draw convergence points the forward kernel found."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=10):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cases = [("headline", lambda: W.structural(B=4096, n=1000), "standard"),
         ("headline/8", lambda: W.structural(B=512, n=1000), "standard"),
         ("c3", lambda: W.structural(B=8192, n=2000), "cholesky")]
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    cases = [(f"B={B} T={n}", (lambda B=B, n=n: W.structural(B=B, n=n)), "standard")
             for n in (500, 1000, 2000) for B in (512, 1024, 2048, 4096, 8192)]
if len(sys.argv) > 1 and sys.argv[1] == "hint":
    # placement of the forward launch from the previous call's convergence points (kf_vg_order_kernel): the same inputs again (the hint
    # is exact), a sampler-like sequence (every call's variances 5 % / 20 % away from the previous call's), no hint
    for name, mk, ft in cases:
        w = mk()
        rng = np.random.default_rng(5)
        sets = {}
        for tag, sd in (("5 %", 0.05), ("20 %", 0.2)):
            w2 = dict(w)
            w2["Q"] = w["Q"] * np.exp(sd * rng.standard_normal(w["B"]))[:, None, None]
            w2["H"] = w["H"] * np.exp(sd * rng.standard_normal(w["B"]))[:, None, None]
            sets[tag] = [eng.to_device(w2["y"])] + [eng.to_device(w2[k]) for k in KEYS]
        d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
        kw = dict(filter_type=ft, mvn_const="exact", exact_recursion=False)
        t_same = ms(lambda: eng.value_and_grad(*d, **kw))
        out = [f"same inputs {t_same:7.3f} ms"]
        for tag, d2 in sets.items():
            def alt():
                eng.value_and_grad(*d, **kw)
                eng.value_and_grad(*d2, **kw)
            out.append(f"alternating with variances {tag} away {ms(alt) / 2:7.3f} ms")
        os.environ["KF_VG_SCHED_HINT"] = "0"
        out.append(f"no hint {ms(lambda: eng.value_and_grad(*d, **kw)):7.3f} ms")
        del os.environ["KF_VG_SCHED_HINT"]
        print(f"{name:11s} {ft:9s} " + "   ".join(out), flush=True)
    sys.exit(0)
for name, mk, ft in cases:
    w = mk()
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(filter_type=ft, mvn_const="exact")
    t_a = ms(lambda: eng.value_and_grad(*d, **kw))
    t_s = ms(lambda: eng.value_and_grad(*d, exact_recursion=False, **kw))
    t_x = ms(lambda: eng.value_and_grad(*d, exact_recursion=True, **kw))
    lp, info, g = eng.value_and_grad(*d, exact_recursion=False, **kw)
    lpx, _, gx = eng.value_and_grad(*d, exact_recursion=True, **kw)
    torch.cuda.synchronize()
    worst = 0.0
    for k in g:
        B = g[k].shape[0]
        sc = 1.0 + gx[k].reshape(B, -1).abs().max(dim=1).values
        worst = max(worst, float(((g[k] - gx[k]).reshape(B, -1).abs().max(dim=1).values / sc).max()))
    print(f"{name:11s} {ft:9s} default {t_a:7.3f} ms   steady-state adjoint {t_s:7.3f} ms   exact adjoint at every step {t_x:7.3f} ms   "
          f"max rel dlogp {float(((lp - lpx).abs() / lpx.abs()).max()):.2e}   worst gradient entry (rel. to the draw's largest) {worst:.2e}",
          flush=True)
