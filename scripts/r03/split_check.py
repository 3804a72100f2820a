"""Split forward pipeline (recursion chunks + time-parallel steady kernels) against the step-by-step kernel and the C oracle,
and its timing next to the in-kernel steady phase (KF_B4_SPLIT=0) and KF_FLAG_EXACT_RECURSION."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import workloads as W
from oracle import c_oracle as co
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
KEYS = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")


def ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, w, ft="standard", oracle_draws=64, full=True, time_it=True, **kw):
    dev = [eng.to_device(w[k]) for k in KEYS]
    fkw = dict(filter_type=ft, mvn_const="exact", store_ll_obs=full, store_filtered=full, store_predicted=full, **kw)
    os.environ["KF_B4_SPLIT"] = "1"
    rs = eng.forward(*dev, **fkw)
    torch.cuda.synchronize()
    rx = eng.forward(*dev, exact_recursion=True, **fkw)
    torch.cuda.synchronize()
    out = [f"{name}: B={w['B']} n={w['n']} m={w['m']}"]
    rel = ((rs.logp - rx.logp).abs() / rx.logp.abs().clamp_min(1.0)).max().item()
    out.append(f"split vs exact: rel dlogp {rel:.2e}")
    if full:
        for nm in ("ll_obs", "filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
            a, b = getattr(rs, nm), getattr(rx, nm)
            out.append(f"{nm} {(a - b).abs().max().item():.1e}")
    assert int(rs.info.abs().sum().item()) == int(rx.info.abs().sum().item()), "info differs"
    if oracle_draws:
        k = min(oracle_draws, w["B"])
        ref = co.kalman_filter(w["y"], *[w[x][:k] for x in KEYS[1:]], filter_type=ft, mvn_const="exact", full=full)
        dl = np.abs(rs.logp[:k].cpu().numpy() - ref["logp"])
        out.append(f"vs oracle: max abs dlogp {dl.max():.2e} (|logp| <= {np.abs(ref['logp']).max():.0f})")
        if full:
            out.append(f"pred_P {np.abs(rs.predicted_covs[:k].cpu().numpy() - ref['predicted_covs']).max():.1e} "
                       f"filt_a {np.abs(rs.filtered_states[:k].cpu().numpy() - ref['filtered_states']).max():.1e}")
    print(" | ".join(out), flush=True)
    if time_it:
        t_split = ms(lambda: eng.forward(*dev, out=rs, **fkw))
        os.environ["KF_B4_SPLIT"] = "0"
        t_in = ms(lambda: eng.forward(*dev, out=rs, **fkw))
        os.environ["KF_B4_SPLIT"] = "1"
        t_ex = ms(lambda: eng.forward(*dev, out=rx, exact_recursion=True, **fkw))
        print(f"    time: split {t_split:.3f} ms | in-kernel steady {t_in:.3f} ms | exact {t_ex:.3f} ms", flush=True)
    del dev, rs, rx
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "small"):
        for m in (1, 2, 3, 5, 6, 7, 9, 10):
            for n in (256, 300, 1000):
                w = W.random_dense(37, n, 1, m, min(m, 3), seed=100 + m)
                check(f"dense m={m}", w, time_it=False, oracle_draws=37)
        w = W.structural(B=333, n=700)
        y = w["y"].copy()
        y[[5, 130, 131, 400]] = np.nan
        w = dict(w, y=y)
        check("structural + NaN rows", w, time_it=False)
        check("structural logp only", W.structural(B=333, n=700), full=False, time_it=False)
        check("structural cholesky", W.structural(B=100, n=2000), ft="cholesky", time_it=False)
        check("structural univariate", W.structural(B=100, n=600), ft="univariate", time_it=False)
    if which in ("all", "headline"):
        check("headline", W.WORKLOADS["headline"](), oracle_draws=256)
        check("headline logp-only", W.WORKLOADS["headline"](), oracle_draws=256, full=False)
        check("headline B=512", W.WORKLOADS["headline"](B=512), oracle_draws=64)
    if which in ("all", "c3"):
        check("c3", W.WORKLOADS["c3"](), ft="cholesky", oracle_draws=64)
