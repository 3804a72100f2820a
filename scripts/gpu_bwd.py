"""Backward timing: mfma adjoint vs the current automatic choice."""
import sys, time
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
for name, Bsub in (("headline", None), ("c3", 8192), ("c4", 2048), ("c4", 16384), ("c4_m10", 2048), ("c2", None)):
    w = W.WORKLOADS[name]()
    B = w["B"] if Bsub is None else Bsub
    n, p, m, r = w["n"], w["p"], w["m"], w["r"]
    dev = [eng.to_device(w["y"])] + [eng.to_device(w[k][:B]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev, store_filtered=False, store_ll_obs=False, h_diagonal=True)
    ones = torch.ones(B, device="cuda", dtype=torch.float64)
    base = None
    for kern, hd in (("auto", False), ("mfma", True)):
        eng.set_forward_kernel(kern)
        try:
            fn = lambda: eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                                      g_logp=ones, h_diagonal=hd)
            g = fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                g = fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            if base is None:
                base = g
            err = max(((g[k] - base[k]).abs().max() / (1 + base[k].abs().max())).item() for k in ("a0", "T", "Z", "R", "Q"))
            print(f"{name} B={B} m={m} p={p} bwd {kern:5s} {min(ts)*1e3:8.3f} ms  max rel diff vs auto {err:.1e}", flush=True)
        finally:
            eng.set_forward_kernel("auto")
    del dev, res
    torch.cuda.empty_cache()
