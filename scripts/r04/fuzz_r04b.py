"""Random-shape parity sweep of the paths of round 4's second half.  usage: python scripts/r04/fuzz_r04b.py [n_cases] [seed]
  vgs     logp + gradient with the steady-state path forced (time-parallel forward means + adjoint, forward launch placed by whatever
          the reused workspace holds: the cases change shape all the time) against the adjoint at every step; 1 <= m <= 10 (the block
          kernels forced for m <= 4), series lengths around the 63 / 64-step block boundaries, missing observations, intercepts,
          never-converging draws, per-draw series
  smooth  RTS smoother with the time-parallel kernels behind the filter's convergence point forced (KF_SMOOTHER_SPLIT=1) against every
          step in kf_smooth_b4_kernel and the numpy oracle; tapes of an every-step and of a steady-state forward, missing observations
          anywhere (also the last one), slow draws, 5 <= m <= 10
  lane    logp + gradient behind the lane kernel (m <= 3; m = 4 with the lane kernel forced takes the full sweep): the default -- steady-
          state adjoint behind each draw's convergence point, exact sweep over the compact tapes before it -- against the full sweep"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
GRADS = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 41)
eng = KalmanEngine()
worst = dict(vgs_vs_exact=0.0, smooth_vs_every_step=0.0, smooth_vs_oracle=0.0, lane_vs_exact=0.0)
counts = dict(vgs=0, smooth=0, smooth_split_taken=0, lane=0, lane_steady_taken=0)
flagged = 0
t0 = time.time()


def rel_rows(a_, x_):
    B = a_.shape[0]
    return float(np.max(np.max(np.abs(a_ - x_).reshape(B, -1), axis=1) / (1.0 + np.max(np.abs(x_).reshape(B, -1), axis=1))))


def slow(w, idx):
    for b in idx:
        if w["m"] >= 2:
            w["T"][b, 0, 0] = 1.0
            w["T"][b, 0, 1] = 1.0
            w["T"][b, 1, 1] = 1.0
            w["P0"][b] = 1e3 * np.eye(w["m"])
            w["Q"][b] = w["Q"][b] * 1e-8


for case in range(n_cases):
    kind = int(rng.integers(0, 3))
    seed = int(rng.integers(1 << 30))
    n = int(rng.choice([64, 65, 100, 127, 128, 129, 191, 200, 320, 700]))
    B = int(rng.choice([1, 3, 20, 70]))
    if kind == 0:   # ---- value and gradient, steady-state path forced -------------------------------------------------------------------
        m = int(rng.integers(1, 11))
        r = int(rng.integers(1, 5))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        w = W.structural(B=B, n=n, seed=seed) if rng.integers(0, 3) == 0 else W.random_dense(B, n, 1, m, r, seed=seed)
        m = w["m"]
        if rng.integers(0, 3) == 0 and w["name"].startswith("structural"):   # (a unit root put into a random dense T explodes)
            slow(w, rng.integers(0, B, size=max(1, B // 8)))
        y = w["y"].copy()
        if rng.integers(0, 4) == 0:
            y = y[None] + 0.1 * rng.standard_normal((B,) + y.shape)   # per-draw series
        for _ in range(int(rng.integers(0, 3))):
            y[..., rng.integers(0, n), :] = np.nan
        c = 0.05 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
        d = 0.1 * rng.standard_normal((B, 1)) if rng.integers(0, 2) else None
        gl = 1.0 + 0.3 * rng.standard_normal(B) if rng.integers(0, 2) else None
        args = (y,) + tuple(w[x] for x in KEYS)
        kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
        if m <= 4:
            eng.set_forward_kernel("b4")
        try:
            lp, info, g = eng.value_and_grad(*args, exact_recursion=False, **kw)
            lpx, infox, gx = eng.value_and_grad(*args, exact_recursion=True, **kw)
        finally:
            eng.set_forward_kernel("auto")
        e = rel_rows(lp.cpu().numpy()[:, None], lpx.cpu().numpy()[:, None])
        ewhere = "logp"
        for nm in GRADS:
            en = rel_rows(g[nm].cpu().numpy(), gx[nm].cpu().numpy())
            if not (en <= e):
                e, ewhere = en, nm
        worst["vgs_vs_exact"] = max(worst["vgs_vs_exact"], e)
        counts["vgs"] += 1
        if not (e <= 1e-10) or not np.array_equal(info.cpu().numpy(), infox.cpu().numpy()):
            flagged += 1
            a_, x_ = (lp, lpx) if ewhere == "logp" else (g[ewhere], gx[ewhere])
            a_, x_ = a_.cpu().numpy().reshape(B, -1), x_.cpu().numpy().reshape(B, -1)
            bb = int(np.argmax(np.max(np.abs(a_ - x_), axis=1) / (1.0 + np.max(np.abs(x_), axis=1))))
            print(f"FLAG vgs case {case} seed {seed}: {w['name']} m={m} r={w['r']} n={n} B={B} {ft} y{y.shape} c={c is not None} d={d is not None} "
                  f"gl={gl is not None}: vs exact {e:.2e} in {ewhere} draw {bb}: {a_[bb][:4]} vs {x_[bb][:4]}; logp {lp[bb].item():.6g} "
                  f"{lpx[bb].item():.6g} nan rows {np.argwhere(np.isnan(y.reshape(-1, n)[0] if y.ndim == 3 else y[:, 0])).ravel()[:6]}", flush=True)
    elif kind == 2:   # ---- value and gradient behind the lane kernel ---------------------------------------------------------------------
        m = int(rng.integers(1, 5))
        r = int(rng.integers(1, 3))
        B = int(rng.choice([1, 3, 5, 20, 70, 300]))
        ft = str(rng.choice(["standard", "cholesky", "univariate", "single"]))
        w = W.arma11(B=B, n=n, seed=seed) if (m == 2 and rng.integers(0, 2)) else W.random_dense(B, n, 1, m, r, seed=seed)
        m = w["m"]
        y = w["y"].copy()
        if rng.integers(0, 4) == 0:
            y = y[None] + 0.1 * rng.standard_normal((B,) + y.shape)
        for _ in range(int(rng.integers(0, 3))):
            y[..., rng.integers(0, n), :] = np.nan
        c = 0.05 * rng.standard_normal((B, m)) if rng.integers(0, 2) else None
        d = 0.1 * rng.standard_normal((B, 1)) if rng.integers(0, 2) else None
        gl = 1.0 + 0.3 * rng.standard_normal(B) if rng.integers(0, 2) else None
        args = (y,) + tuple(w[x] for x in KEYS)
        kw = dict(c=c, d=d, g_logp=gl, filter_type=ft, mvn_const="exact")
        eng.set_forward_kernel("lane")
        try:
            lp, info, g = eng.value_and_grad(*args, **kw)
            lpx, infox, gx = eng.value_and_grad(*args, exact_recursion=True, **kw)
        finally:
            eng.set_forward_kernel("auto")
        e = rel_rows(lp.cpu().numpy()[:, None], lpx.cpu().numpy()[:, None])
        ewhere = "logp"
        taken = False
        for nm in GRADS:
            taken = taken or not np.array_equal(g[nm].cpu().numpy(), gx[nm].cpu().numpy())
            en = rel_rows(g[nm].cpu().numpy(), gx[nm].cpu().numpy())
            if not (en <= e):
                e, ewhere = en, nm
        worst["lane_vs_exact"] = max(worst["lane_vs_exact"], e)
        counts["lane"] += 1
        counts["lane_steady_taken"] += int(taken)
        if not (e <= 1e-10) or not np.array_equal(info.cpu().numpy(), infox.cpu().numpy()):
            flagged += 1
            print(f"FLAG lane case {case} seed {seed}: {w['name']} m={m} r={w['r']} n={n} B={B} {ft} y{y.shape} c={c is not None} d={d is not None}: "
                  f"vs the full sweep {e:.2e} in {ewhere}", flush=True)
    else:   # ---- smoother, split forced --------------------------------------------------------------------------------------------------
        m = int(rng.integers(5, 11))
        r = int(rng.integers(1, 5))
        w = W.structural(B=B, n=n, seed=seed) if rng.integers(0, 3) == 0 else W.random_dense(B, n, 1, m, r, seed=seed)
        m = w["m"]
        has_slow = rng.integers(0, 3) == 0 and w["name"].startswith("structural")
        if has_slow:
            slow(w, rng.integers(0, B, size=max(1, B // 8)))
        y = w["y"].copy()
        for _ in range(int(rng.integers(0, 3))):
            y[rng.integers(0, n)] = np.nan
        if rng.integers(0, 6) == 0:
            y[n - 1] = np.nan
        res = eng.forward(y, *[w[k] for k in KEYS], mvn_const="exact", store_predicted=False, store_ll_obs=False,
                          exact_recursion=bool(rng.integers(0, 2)))
        args = (w["T"], w["R"], w["Q"], res.filtered_states, res.filtered_covs)
        os.environ["KF_SMOOTHER_SPLIT"] = "1"
        sa, sP = eng.smooth(*args)
        os.environ["KF_SMOOTHER_SPLIT"] = "0"
        xa, xP = eng.smooth(*args)
        del os.environ["KF_SMOOTHER_SPLIT"]
        sa, sP, xa, xP = sa.cpu().numpy(), sP.cpu().numpy(), xa.cpu().numpy(), xP.cpu().numpy()
        e = max(rel_rows(sa, xa), rel_rows(sP, xP))
        counts["smooth_split_taken"] += int(not (np.array_equal(sa, xa) and np.array_equal(sP, xP)))
        na, nP = kn.kalman_smoother(w["T"], w["R"], w["Q"], res.filtered_states.cpu().numpy(), res.filtered_covs.cpu().numpy())
        e2 = max(rel_rows(sa, na), rel_rows(sP, nP))
        e2x = max(rel_rows(xa, na), rel_rows(xP, nP))
        worst["smooth_vs_every_step"] = max(worst["smooth_vs_every_step"], e)
        if not has_slow:
            worst["smooth_vs_oracle"] = max(worst["smooth_vs_oracle"], e2)
        counts["smooth"] += 1
        if e2x > 1e-2 and not has_slow:   # where the every-step kernel itself is far from the oracle: what kind of case is it
            bb = int(np.argmax(np.max(np.abs(xP - nP).reshape(B, -1), axis=1) / (1.0 + np.max(np.abs(nP).reshape(B, -1), axis=1))))
            Tb = w["T"][bb]
            Ph = Tb @ res.filtered_covs.cpu().numpy()[bb, -1] @ Tb.T + w["R"][bb] @ w["Q"][bb] @ w["R"][bb].T
            ev = np.linalg.eigvalsh(0.5 * (Ph + Ph.T))
            print(f"INFO smooth case {case} seed {seed}: {w['name']} n={n} B={B}: every step vs oracle {e2x:.2e} on draw {bb}: eig(P_hat) "
                  f"{ev[0]:.2e} .. {ev[-1]:.2e}; max |P_s| gpu {np.abs(xP[bb]).max():.3e} oracle {np.abs(nP[bb]).max():.3e}; forward info "
                  f"{res.info.cpu().numpy()[bb]}", flush=True)
        # (the oracle comparison is held to what the every-step kernel itself achieves on the case: diffuse slow draws sit 1e-7 away)
        # ... and the split to five times that distance: both invert the same ill-conditioned P_hat, one by Cholesky, one by iteration;
        # seen: 9.3e-10 where the every-step kernel sits 2.5e-10 from the oracle, m = 9, r = 3)
        # (floor: the tape tolerance 1e-14 times the conditioning the Cholesky pivot rule admits; measured 3e-13 with the rule at 1e-2)
        if not (e <= max(2e-10, 5.0 * e2x)) or not (e2 <= max(1e-9, 5.0 * e2x)) or not np.all(np.isfinite(sa)):
            flagged += 1
            print(f"FLAG smooth case {case}: {w['name']} m={m} n={n} B={B}: vs every step {e:.2e}, vs oracle {e2:.2e} "
                  f"(every step vs oracle {e2x:.2e})", flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s: {counts}; flagged {flagged}; worst relative errors {worst}")
