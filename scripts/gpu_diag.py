"""GPU-box diagnostic: HIP forward filter vs the numpy oracle, many shapes, error table.

    python scripts/gpu_diag.py [--out gpurun_out/diag.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) if a.size else 0.0


def run_case(eng, w, ft, lanes, mvn="exact", y=None):
    y = w["y"] if y is None else y
    eng.set_lanes_per_draw(lanes)
    res = eng.forward(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"],
                      filter_type=ft, mvn_const=mvn)
    torch.cuda.synchronize()
    o = kn.kalman_filter(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"],
                         filter_type=ft, mvn_const=mvn, cholesky_triangle="correct",
                         partial_missing="delete")
    errs = dict(
        logp=float(np.max(np.abs(res.logp.cpu().numpy() - o["logp"]))),
        ll=relerr(res.ll_obs.cpu().numpy(), o["ll_obs"]),
        fa=relerr(res.filtered_states.cpu().numpy(), o["filtered_states"]),
        pa=relerr(res.predicted_states.cpu().numpy(), o["predicted_states"]),
        fP=relerr(res.filtered_covs.cpu().numpy(), o["filtered_covs"]),
        pP=relerr(res.predicted_covs.cpu().numpy(), o["predicted_covs"]),
    )
    info = res.info.cpu().numpy()
    return errs, int(np.count_nonzero(info)), float(np.max(np.abs(o["logp"])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    eng = KalmanEngine()
    emit(f"device: {torch.cuda.get_device_name(0)}")
    cases = [
        ("c1", W.nile_local_level(B=5)),
        ("nile_llt", W.nile_local_linear_trend(B=3)),
        ("c2", W.arma11(B=70, n=150)),
        ("headline", W.structural(B=37, n=120)),
        ("c4_m10", W.varmax(B=9, n=60, order=(2, 0))),
        ("c4_m15", W.varmax(B=9, n=50, order=(2, 1))),
        ("dense_p3m4r2", W.random_dense(11, 50, 3, 4, 2, seed=3)),
        ("dense_p2m6r3", W.random_dense(7, 40, 2, 6, 3, seed=4)),
        ("dense_p9m12r3", W.random_dense(5, 30, 9, 12, 3, seed=5)),
        ("dense_p1m20r2", W.random_dense(5, 30, 1, 20, 2, seed=6)),
        ("dense_p1m33r2", W.random_dense(3, 20, 1, 33, 2, seed=7)),
    ]
    for name, w in cases:
        for ft in ("standard", "cholesky", "univariate") + (("single",) if w["p"] == 1 else ()):
            for lanes in (0, 4, 16, 64):
                try:
                    e, nbad, mx = run_case(eng, w, ft, lanes)
                    emit(f"{name:14s} p{w['p']} m{w['m']:<3d} {ft:10s} G={lanes:<2d} "
                         f"|dlogp| {e['logp']:.1e} (|logp|<={mx:.0f}) ll {e['ll']:.1e} fa {e['fa']:.1e} "
                         f"pa {e['pa']:.1e} fP {e['fP']:.1e} pP {e['pP']:.1e} info!=0: {nbad}")
                except Exception as ex:  # noqa: BLE001
                    emit(f"{name:14s} {ft:10s} G={lanes}: EXC {type(ex).__name__}: {ex}")
    # missing data: whole rows and single cells
    w = W.random_dense(6, 40, 3, 4, 2, seed=8)
    y = w["y"].copy()
    y[[3, 17]] = np.nan
    y[5, 0] = np.nan
    y[9, 1:] = np.nan
    for ft in ("standard", "cholesky", "univariate"):
        e, nbad, mx = run_case(eng, w, ft, 0, y=y)
        emit(f"missing p3m4   {ft:10s} |dlogp| {e['logp']:.1e} ll {e['ll']:.1e} fa {e['fa']:.1e} "
             f"fP {e['fP']:.1e} pP {e['pP']:.1e} info!=0: {nbad}")
    w = W.structural(B=8, n=60)
    y = w["y"].copy()
    y[[0, 7, 8, 59]] = np.nan
    for ft in ("standard", "cholesky", "univariate", "single"):
        for mvn in ("exact", "reference"):
            e, nbad, mx = run_case(eng, w, ft, 0, mvn=mvn, y=y)
            emit(f"missing m10    {ft:10s} {mvn:9s} |dlogp| {e['logp']:.1e} ll {e['ll']:.1e} "
                 f"fa {e['fa']:.1e} fP {e['fP']:.1e} info!=0: {nbad}")

    # quick timing of the headline shape
    w = W.structural(B=4096, n=1000)
    dev = {k: eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")}
    for lanes in (0, 16, 32, 64):
        eng.set_lanes_per_draw(lanes)
        for full in (False, True):
            kw = dict(store_ll_obs=full, store_filtered=full, store_predicted=full)
            res = eng.forward(*[dev[k] for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")], **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                eng.forward(*[dev[k] for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")], out=res, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            emit(f"headline B=4096 n=1000 m=10 G={lanes:<2d} full_tape={full}: {dt * 1e3:.3f} ms "
                 f"-> {4096 / dt:.3e} evals/s")
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
