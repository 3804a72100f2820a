"""logp + gradient in one call: AUTO against each forward-kernel family forced, over a grid of shapes -- looks for shapes where AUTO is
not the fastest choice (the table that found the row-kernel shapes of kf_vg_tile_path)."""
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


n = 300
grid = [(1, m, B) for m in (5, 7, 9, 11, 12, 14) for B in (512, 2048, 8192, 32768)]
grid += [(p, m, B) for (p, m) in ((2, 6), (3, 8), (5, 10), (3, 15), (5, 15)) for B in (2048, 16384)]
for p, m, B in grid:
    w = W.random_dense(B, n, p, m, 2, seed=m + p)
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(filter_type="standard", mvn_const="exact", h_diagonal=True)
    out = []
    t_auto = ms(lambda: eng.value_and_grad(*d, **kw))
    best = ("auto", t_auto)
    for forced in ("b4", "mfma", "row"):
        eng.set_forward_kernel(forced)
        try:
            t = ms(lambda: eng.value_and_grad(*d, **kw))
        except RuntimeError:
            t = float("nan")
        finally:
            eng.set_forward_kernel("auto")
        out.append(f"{forced} {t:7.3f}")
        if t == t and t < best[1]:
            best = (forced, t)
    flag = "" if best[0] == "auto" or best[1] > 0.9 * t_auto else f"   <-- {best[0]} is {t_auto / best[1]:.2f}x faster"
    print(f"p={p} m={m:2d} B={B:5d} x {n} ({eng.forward_kernel_name(B, n, p, m, 2, h_diagonal=True):26s}): auto {t_auto:7.3f}   " + "   ".join(out) + flag, flush=True)
