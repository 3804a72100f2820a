"""Forward-kernel selection sweep: row / multivariate-row / mfma over (m, p, B)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
n = 200


def run(p, m, B, kern, mode):
    w = W.random_dense(64, n, p, m, max(1, m // 2), seed=7)
    rep = (B + 63) // 64
    dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))[:B])
                                     for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    eng.set_forward_kernel(kern)
    kw = dict(h_diagonal=True)
    if mode == "logp":
        kw.update(store_ll_obs=False, store_filtered=False, store_predicted=False)
    try:
        out = eng.forward(*dev, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = eng.forward(*dev, out=out, **kw)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3 * (1000 / n)
    finally:
        eng.set_forward_kernel("auto")


for p, ms in ((1, (5, 6, 7, 8, 9, 10, 12, 16)), (3, (4, 6, 8, 10, 12))):
    for m in ms:
        for B in (4096, 32768):
            row = []
            for kern in ("row", "mfma"):
                for mode in ("logp", "tapes"):
                    try:
                        row.append(f"{kern}/{mode} {run(p, m, B, kern, mode):8.2f}")
                    except Exception as e:  # noqa
                        row.append(f"{kern}/{mode}   n/a   ")
            print(f"p={p} m={m:2d} B={B:6d} (ms per 1000 steps): " + " | ".join(row), flush=True)
