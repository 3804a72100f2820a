"""Cost of one log-density + gradient evaluation (what a NUTS leapfrog step consumes): forward with
the predicted tapes only, then the adjoint.  usage: python scripts/gpu_nuts_step.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
print("| workload | B | T | p,m,r | forward (predicted tapes only) ms | gradient ms | logp + gradient ms | evaluations/s |")
print("|---|---|---|---|---|---|---|---|")
CASES = (("headline", {}, "standard"), ("c3", {}, "cholesky"), ("c4", dict(B=16384), "standard"),
         ("hourly", {}, "standard"), ("weekly", {}, "standard"))
if len(sys.argv) > 1:  # few draws: the latency of one draw's T sequential steps (a few NUTS chains)
    CASES = tuple((n_, dict(B=int(sys.argv[1])), f_) for n_, _, f_ in CASES)
for name, kw, ft in CASES:
    w = W.WORKLOADS[name](**kw)
    d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    B = w["B"]
    gl = torch.ones(B, dtype=torch.float64, device=d[0].device)
    fkw = dict(filter_type=ft, store_ll_obs=False, store_filtered=False, h_diagonal=True)
    res = eng.forward(*d, **fkw)
    bkw = dict(predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl, filter_type=ft,
               h_diagonal=True)
    eng.backward(*d, **bkw)
    torch.cuda.synchronize()

    def tm(f, k=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    tf = tm(lambda: eng.forward(*d, out=res, **fkw))
    tb = tm(lambda: eng.backward(*d, **bkw))
    print(f"| {name} | {B} | {w['n']} | {w['p']},{w['m']},{w['r']} | {tf:.2f} | {tb:.2f} | {tf + tb:.2f} | {B / (tf + tb) * 1e3:.3g} |", flush=True)
