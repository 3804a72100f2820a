"""Forward kernel timing for 16 < m <= 48 (block-tile kernel vs LDS kernel)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pymc_statespace_amd import KalmanEngine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import workloads as W  # noqa

def main():
    eng = KalmanEngine()
    B, n = 4096, 1000
    for (p, m, r) in [(1, 17, 5), (1, 20, 5), (1, 24, 8), (1, 32, 8), (1, 36, 8), (1, 40, 8), (1, 48, 8), (3, 32, 8)]:
        w = W.random_dense(B, n, p, m, r, seed=m)
        args = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
        for kern in ("auto", "lds"):
            if kern == "lds" and m > 24:
                continue
            eng.set_forward_kernel(kern)
            for full in (True, False):
                kw = dict(h_diagonal=True) if full else dict(h_diagonal=True, store_filtered=False, store_predicted=False, store_ll_obs=False)
                eng.forward(*args, **kw); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    eng.forward(*args, **kw)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                print(f"p={p} m={m} r={r} {kern:5s} {'tapes' if full else 'logp '} {dt*1e3:8.2f} ms", flush=True)
        eng.set_forward_kernel("auto")
main()
