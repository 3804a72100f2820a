"""GPU check of the 4x4x4-block forward kernel (kf_forward_b4.hip) against the C oracle + timing vs the 16x16 tile kernel."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import c_oracle as co
from pymc_statespace_amd import KalmanEngine

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
TAPES = ["ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"]
worst = 0.0
for m in (1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14):
    for ft in ("standard", "cholesky", "univariate", "single"):
        for missing in (False, True):
            w = W.random_dense(7, 70, 1, m, min(m, 3), seed=100 + m)
            y = w["y"].copy()
            if missing:
                y[[3, 17, 42, 69]] = np.nan
            c = 0.1 * np.arange(m)[None].repeat(7, 0) / max(m, 1)
            d = 0.05 * np.ones((7, 1))
            ref = co.kalman_filter(y, *[w[k] for k in KEYS], c=c, d=d, filter_type=ft, mvn_const="exact")
            eng.set_forward_kernel("b4")
            out = None
            if missing:  # tapes that start 8 bytes off a 16-byte boundary: the streams' odd-parity path
                from pymc_statespace_amd.engine import FilterResult
                B_, n_ = 7, 70
                def odd(*shape):
                    flat = torch.full((int(np.prod(shape)) + 3,), float("nan"), device="cuda", dtype=torch.float64)
                    return flat[1:1 + int(np.prod(shape))].view(*shape)
                out = FilterResult(logp=odd(B_), ll_obs=odd(B_, n_), filtered_states=odd(B_, n_, m),
                                   predicted_states=odd(B_, n_ + 1, m), filtered_covs=odd(B_, n_, m, m),
                                   predicted_covs=odd(B_, n_ + 1, m, m),
                                   info=torch.empty(B_, device="cuda", dtype=torch.int32))
            res = eng.forward(y, *[w[k] for k in KEYS], c, d, filter_type=ft, mvn_const="exact", out=out)
            eng.set_forward_kernel("auto")
            e = np.abs(res.logp.cpu().numpy() - ref["logp"]).max()
            et = max(np.abs(getattr(res, k).cpu().numpy() - ref[k]).max() for k in TAPES)
            worst = max(worst, e, et)
            flag = "" if (e < 1e-8 and et < 1e-8) else "   <<<<<< FAIL"
            if flag or (ft == "standard" and not missing):
                print(f"m={m:2d} {ft:10s} missing={missing!s:5s} dlogp={e:.2e} dtape={et:.2e} info={int(res.info.abs().sum())}{flag}")
print("worst", worst)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for name in ("headline", "c3"):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    ft = "standard" if name == "headline" else "cholesky"
    for kern in ("mfma", "b4"):
        eng.set_forward_kernel(kern)
        full = eng.forward(*dev, filter_type=ft, mvn_const="exact")
        lo = eng.forward(*dev, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
        t_full = timeit(lambda: eng.forward(*dev, filter_type=ft, mvn_const="exact", out=full))
        t_lo = timeit(lambda: eng.forward(*dev, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
        idx = np.linspace(0, w["B"] - 1, 64).astype(int)
        ref = co.kalman_filter(w["y"], *[w[k][idx] for k in KEYS], filter_type=ft, mvn_const="exact", full=False)
        e = np.abs(full.logp.cpu().numpy()[idx] - ref["logp"]).max()
        print(f"{name:9s} {kern:5s} full {t_full:7.3f} ms   logp-only {t_lo:7.3f} ms   max|dlogp| {e:.2e}")
    eng.set_forward_kernel("auto")
