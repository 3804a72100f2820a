"""Forward filter with p > 1 scalar-observation updates (diagonal H): AUTO against the kernel families forced, full tapes ; logp only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = 300
for p, m in ((2, 4), (2, 8), (2, 12), (3, 6), (3, 8), (3, 12), (5, 8), (5, 10), (5, 15), (8, 12)):
    for B in (2048, 8192, 32768):
        w = W.random_dense(B, n, p, m, 2, seed=m + p)
        d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
        line = f"p={p} m={m:2d} B={B:5d} x {n} ({eng.forward_kernel_name(B, n, p, m, 2, h_diagonal=True):26s})"
        for tag, kw in (("full", dict()), ("logp", dict(store_ll_obs=False, store_filtered=False, store_predicted=False))):
            res = eng.forward(*d, mvn_const="exact", h_diagonal=True, **kw)
            t = {"auto": ms(lambda: eng.forward(*d, mvn_const="exact", h_diagonal=True, out=res, **kw))}
            for forced in ("b4", "mfma", "row"):
                eng.set_forward_kernel(forced)
                try:
                    t[forced] = ms(lambda: eng.forward(*d, mvn_const="exact", h_diagonal=True, out=res, **kw))
                except RuntimeError:
                    pass
                finally:
                    eng.set_forward_kernel("auto")
            best = min(t, key=t.get)
            flag = "" if t["auto"] <= 1.08 * t[best] else f" <-- {best} {t['auto'] / t[best]:.2f}x"
            line += f"   {tag}: " + " ".join(f"{k} {v:.3f}" for k, v in t.items()) + flag
            del res
        print(line, flush=True)
