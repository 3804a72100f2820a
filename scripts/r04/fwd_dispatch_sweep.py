"""Forward filter (full tapes ; log-likelihood only): AUTO against the block / row / lane kernels forced, around the batch sizes where
kf_b4_auto hands over to the packed kernels."""
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


def ms(fn, reps=4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
MS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (3, 4, 5, 6, 7, 8)
BS = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2048, 8192, 16384, 32768, 65536)
for m in MS:
    for B in BS:
        w = W.random_dense(B, n, 1, m, 2, seed=m)
        d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
        line = f"m={m} B={B:5d} x {n} ({eng.forward_kernel_name(B, n, 1, m, 2):22s})"
        for tag, kw in (("full", dict()), ("logp", dict(store_ll_obs=False, store_filtered=False, store_predicted=False))):
            res = eng.forward(*d, mvn_const="exact", **kw)
            t = {"auto": ms(lambda: eng.forward(*d, mvn_const="exact", out=res, **kw))}
            for forced in ("b4", "row", "lane"):
                if forced == "lane" and m > 4:
                    continue
                eng.set_forward_kernel(forced)
                try:
                    t[forced] = ms(lambda: eng.forward(*d, mvn_const="exact", out=res, **kw))
                except RuntimeError:
                    pass
                finally:
                    eng.set_forward_kernel("auto")
            best = min(t, key=t.get)
            flag = "" if t["auto"] <= 1.07 * t[best] else f" <-- {best} {t['auto'] / t[best]:.2f}x"
            line += f"   {tag}: " + " ".join(f"{k} {v:.3f}" for k, v in t.items()) + flag
            del res
        print(line, flush=True)
