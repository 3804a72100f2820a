"""Random-shape parity sweep: forward tapes, gradient and smoother of whatever kernel AUTO selects
against the CPU oracles.  usage: python scripts/gpu_fuzz.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads as W
from oracle import kalman_numpy as kn
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine

TAPES = ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "ll_obs")


def run(n_cases=120, seed=2026, eng=None, verbose=True):
    rng = np.random.default_rng(seed)
    eng = eng or KalmanEngine()
    flagged = 0
    worst = dict(logp=0.0, tapes=0.0, grad=0.0, smooth=0.0)
    kernels = {}
    t0 = time.time()
    for case in range(n_cases):
        m = int(rng.choice([rng.integers(1, 17), rng.integers(17, 49), rng.integers(49, 65), rng.integers(65, 72)],
                           p=[0.35, 0.35, 0.25, 0.05]))
        p = int(min(m, rng.choice([1, 1, 1, 2, 3, 5, 8])))
        r = int(rng.integers(1, m + 1))
        n = int(rng.choice([1, 2, 7, 33, 64, 65, 90]))
        ft = str(rng.choice(["standard", "cholesky", "univariate"] + (["single"] if p == 1 else [])))
        mvn = str(rng.choice(["reference", "exact"]))
        B = int(rng.integers(1, 5))
        dense_h = p > 1 and m <= 40 and rng.random() < 0.2  # joint update with a full H: generic LDS kernels
        w = W.random_dense(B, n, p, m, r, seed=int(rng.integers(1 << 30)), h_diag=not dense_h)
        y = w["y"].copy()
        if n == 1:
            y[:] = 0.2
        if n > 4:
            y[rng.integers(0, n)] = np.nan
            if p > 1 and not dense_h:
                y[rng.integers(0, n), rng.integers(0, p)] = np.nan
        per_draw = n > 1 and rng.random() < 0.25
        if per_draw:
            y = y[None] + 0.1 * rng.standard_normal((B,) + y.shape)
        c = 0.1 * rng.standard_normal((B, m)) if rng.random() < 0.5 else None
        d = 0.1 * rng.standard_normal((B, p)) if rng.random() < 0.5 else None
        args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
        name = eng.forward_kernel_name(B, n, p, m, r, ft, not dense_h)
        kernels[name] = kernels.get(name, 0) + 1
        ref = kn.kalman_filter(*args, c=c, d=d, filter_type=ft, mvn_const=mvn, cholesky_triangle="correct",
                               partial_missing="delete")
        res = eng.forward(*args, c=c, d=d, filter_type=ft, mvn_const=mvn)
        e_lp = float(np.max(np.abs(res.logp.cpu().numpy() - ref["logp"])))
        e_tp = 0.0
        for k in TAPES:
            a, b_ = getattr(res, k).cpu().numpy(), ref[k]
            e_tp = max(e_tp, float(np.max(np.abs(a - b_) / (1.0 + np.abs(b_)))))
        e_g = 0.0
        if n <= 40 and m <= 64:
            g_logp = rng.standard_normal(B)
            g_ll = rng.standard_normal((B, n)) if rng.random() < 0.3 else None
            fto = "standard" if ft == "cholesky" else ft
            if per_draw:
                gts = [kt.logp_and_grads(y[i], *[x[i:i + 1] for x in args[1:]], c=None if c is None else c[i:i + 1],
                                         d=None if d is None else d[i:i + 1], g_logp=g_logp[i:i + 1],
                                         g_ll_obs=None if g_ll is None else g_ll[i:i + 1], filter_type=fto,
                                         mvn_const=mvn)[2] for i in range(B)]
                gt = {k: np.concatenate([gi[k] for gi in gts], axis=0) for k in kt.NAMES}
            else:
                _, _, gt = kt.logp_and_grads(*args, c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=fto,
                                             mvn_const=mvn)
            g = eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                             g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft, mvn_const=mvn, want_cd=True,
                             h_diagonal=not dense_h)
            for k in kt.NAMES:
                a, b_ = g[k].cpu().numpy(), gt[k]
                if dense_h and ft != "univariate":
                    if k in ("P0", "Q", "H"):  # entrywise partials of symmetric inputs: compare symmetric parts
                        a, b_ = a + np.swapaxes(a, -1, -2), b_ + np.swapaxes(b_, -1, -2)
                elif k == "H" and p > 1:
                    a, b_ = np.diagonal(a, axis1=-2, axis2=-1), np.diagonal(b_, axis1=-2, axis2=-1)
                elif k in ("P0", "Q") and (p > 1 or ft == "univariate"):
                    a, b_ = a + np.swapaxes(a, -1, -2), b_ + np.swapaxes(b_, -1, -2)
                e_g = max(e_g, float(np.max(np.abs(a - b_)) / (1.0 + np.max(np.abs(b_)))))
        e_s = 0.0
        if r == m or rng.random() < 0.3:
            cond = 1.0
            if r < m:  # low-rank state noise: the prediction covariance can be nearly singular
                Ph = np.einsum("bij,btjk,blk->btil", w["T"], ref["filtered_covs"], w["T"]) + \
                    np.einsum("bij,bjk,blk->bil", w["R"], w["Q"], w["R"])[:, None]
                ev = np.linalg.eigvalsh(0.5 * (Ph + np.swapaxes(Ph, -1, -2)))
                cond = float((ev[..., -1] / np.maximum(ev[..., 0], 1e-300)).max())
            if cond < 1e7:  # beyond that pinv (reference, fallbacks) and a plain solve legitimately differ
                #             (and m > 64 has no pinv fallback)
                ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
                sa, sP = eng.smooth(w["T"], w["R"], w["Q"], ref["filtered_states"], ref["filtered_covs"])
                e_s = float(np.max(np.abs(sP.cpu().numpy() - rP) / (1.0 + np.abs(rP)))) / max(1.0, cond * 1e-3)
                if not np.isfinite(e_s):
                    e_s = np.inf
        worst["logp"] = max(worst["logp"], e_lp); worst["tapes"] = max(worst["tapes"], e_tp)
        worst["grad"] = max(worst["grad"], e_g); worst["smooth"] = max(worst["smooth"], e_s)
        ok = e_lp < 1e-8 and e_tp < 1e-9 and e_g < 1e-9 and e_s < 1e-8
        flagged += 0 if ok else 1
        flag = "" if ok else "   <-- CHECK"
        if verbose or not ok:
            print(f"{case:3d} p={p} m={m:2d} r={r:2d} n={n:2d} B={B} {ft:10s} {mvn:9s} {'H' if dense_h else '-'}{'y' if per_draw else '-'} {name:26s} "
                  f"dlogp {e_lp:.1e} tapes {e_tp:.1e} grad {e_g:.1e} smooth {e_s:.1e}{flag}", flush=True)
    print("kernels:", kernels)
    print("worst:", {k: f"{v:.2e}" for k, v in worst.items()}, f"flagged: {flagged}", f"({time.time() - t0:.0f} s)")
    return flagged, worst, kernels


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
