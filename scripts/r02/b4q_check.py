"""GPU check of the multi-observation block kernel (kf_forward_b4q.hip) against the numpy oracle + timing against the
16x16 tile kernel / the multivariate row kernel at the C4 shapes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
TAPES = ["ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"]
def args(w, y=None):
    return (w["y"] if y is None else y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
worst = 0.0
SHAPES = [(5, 15, 5), (5, 10, 5), (2, 15, 3), (3, 13, 4), (8, 12, 6), (4, 7, 3), (3, 3, 2), (2, 1, 1), (2, 11, 4), (6, 2, 2)]
for p, m, r in SHAPES:
    for ft in ("standard", "cholesky", "univariate"):
        for mvn in ("reference", "exact"):
            B = 19
            w = W.random_dense(B, 139, p, m, r, seed=500 + 17 * p + m)
            y = w["y"].copy()
            y[[3, 17, 64, 138]] = np.nan
            y[5, 0] = np.nan
            y[70, 1:] = np.nan
            rng = np.random.default_rng(m)
            c = 0.1 * rng.standard_normal((B, m))
            d = 0.1 * rng.standard_normal((B, p))
            ref = kn.kalman_filter(*args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn,
                                   cholesky_triangle="correct", partial_missing="delete")
            eng.set_forward_kernel("b4")
            assert eng.forward_kernel_name(B, 139, p, m, r, ft, True) == "kf_fwd_b4q_kernel"
            res = eng.forward(*args(w, y), c=c, d=d, filter_type=ft, mvn_const=mvn)
            eng.set_forward_kernel("auto")
            e = np.abs(res.logp.cpu().numpy() - ref["logp"]).max()
            et = max(np.abs(getattr(res, k).cpu().numpy() - ref[k]).max() for k in TAPES)
            worst = max(worst, e, et)
            pc = res.predicted_covs.cpu().numpy()[:, 1:]
            sym = np.array_equal(pc, np.swapaxes(pc, -1, -2))
            flag = "" if (e < 1e-8 and et < 1e-8 and sym and not res.info.cpu().numpy().any()) else "   <<<<<< FAIL"
            if flag or (ft == "standard" and mvn == "exact"):
                print(f"p={p} m={m:2d} {ft:10s} {mvn:9s} dlogp={e:.2e} dtape={et:.2e} sym={sym} info={int(res.info.abs().sum())}{flag}")
print("worst", worst)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

import os
for B in (16384, 2048, 1024):
    w = W.varmax(B=B, n=1000)
    dev = [eng.to_device(x) for x in args(w)]
    ref = None
    for env in ("0", "1"):
        os.environ["KF_FORWARD_B4Q"] = env
        name = eng.forward_kernel_name(B, 1000, w["p"], w["m"], w["r"], "standard", True)
        full = eng.forward(*dev, mvn_const="exact", h_diagonal=True)
        t_full = timeit(lambda: eng.forward(*dev, mvn_const="exact", h_diagonal=True, out=full))
        lo = eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False, store_predicted=False)
        t_lo = timeit(lambda: eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo))
        po = eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False)
        t_po = timeit(lambda: eng.forward(*dev, mvn_const="exact", h_diagonal=True, store_ll_obs=False, store_filtered=False, out=po))
        if ref is None:
            ref = (full.logp.clone(), full.filtered_covs[::97, -1].clone())
            dl = dc = 0.0
        else:
            dl = (full.logp - ref[0]).abs().max().item()
            dc = (full.filtered_covs[::97, -1] - ref[1]).abs().max().item()
        print(f"C4 B={B} p={w['p']} m={w['m']} {name:28s} full {t_full:7.2f} ms  logp-only {t_lo:7.2f} ms  pred-only {t_po:7.2f} ms  dlogp {dl:.1e} dcov {dc:.1e}")
    os.environ.pop("KF_FORWARD_B4Q")
