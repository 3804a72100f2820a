"""C4 (VARMAX) forward timing: generic LDS kernel vs the multivariate row kernel."""
import sys, time
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
for name in ("c4", "c4_m10"):
    w = W.WORKLOADS[name]()
    B, n, p, m, r = w["B"], w["n"], w["p"], w["m"], w["r"]
    dev = [eng.to_device(x) for x in (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])]
    M = m + (m & 1)
    base = None
    for ft in ("standard", "univariate"):
        for kern, lanes in (("lds", 0), ("row", M * 2), ("row", M * (4 if M >= 14 else 3))):
            eng.set_forward_kernel(kern)
            eng.set_lanes_per_draw(lanes)
            for mode in ("logp", "tapes"):
                kw = dict(filter_type=ft, h_diagonal=True)
                if mode == "logp":
                    kw.update(store_ll_obs=False, store_filtered=False, store_predicted=False)
                out = eng.forward(*dev, **kw)
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    out = eng.forward(*dev, out=out, **kw)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                lp = out.logp.clone()
                if base is None:
                    base = lp
                print(f"{name} B={B} m={m} p={p} {ft:10s} {kern}:{lanes:3d} {mode:5s} "
                      f"{min(ts)*1e3:8.2f} ms  max|dlogp|={(lp-base).abs().max().item():.2e}", flush=True)
                del out
    base = None
