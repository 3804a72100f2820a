"""Forward timing: mfma kernel vs the current automatic choice (headline, c3 shard, c4, c4_m10)."""
import sys, time
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
for name, Bsub in (("headline", None), ("c4", None), ("c4_m10", None), ("c3", 16384), ("c2", None)):
    w = W.WORKLOADS[name]()
    B = w["B"] if Bsub is None else Bsub
    n, p, m, r = w["n"], w["p"], w["m"], w["r"]
    sl = slice(0, B)
    dev = [eng.to_device(w["y"])] + [eng.to_device(w[k][sl]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    base = None
    for kern in ("auto", "mfma"):
        eng.set_forward_kernel(kern)
        for mode in ("logp", "tapes"):
            kw = dict(h_diagonal=True)
            if mode == "logp":
                kw.update(store_ll_obs=False, store_filtered=False, store_predicted=False)
            out = eng.forward(*dev, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = eng.forward(*dev, out=out, **kw)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            lp = out.logp.clone()
            if base is None:
                base = lp
            print(f"{name} B={B} m={m} p={p} {kern:5s} {mode:5s} {min(ts)*1e3:8.3f} ms  "
                  f"max|dlogp|={(lp-base).abs().max().item():.2e}", flush=True)
            del out
    eng.set_forward_kernel("auto")
