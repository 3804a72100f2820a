"""Generate the golden vectors under tests/golden/*.npz from statsmodels.

Run ONCE in the build container (statsmodels lives only in the conda python):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py

statsmodels 0.12.2's low-level Cython ``KalmanSmoother`` is the same external
oracle the reference's parity test compares against
(reference tests/test_kalman_filter.py:226-241, tests/utilities/test_helpers.py:110-140).
``statsmodels.api`` itself does not import in this image (pandas Int64Index), the
low-level module does after the one-line ``np.MachAr`` shim below.

Each .npz holds INPUTS (reference layouts, leading draw axis B) and EXPECTED
outputs (time-major, draw axis first):
  y (n,p) a0 (B,m) P0 (B,m,m) T Z R H Q
  llf_obs (B,n) filtered_state (B,n,m) predicted_state (B,n+1,m)
  filtered_state_cov (B,n,m,m) predicted_state_cov (B,n+1,m,m)
  smoothed_state (B,n,m) smoothed_state_cov (B,n,m,m)
statsmodels uses the exact p*log(2pi) constant and proper row deletion for
partially missing observations, so for p > 1 these pin the build's
mvn_const="exact" / partial_missing="delete" modes.
"""
import os
import sys

import numpy as np

np.MachAr = type("M", (), {"eps": np.finfo(float).eps})  # statsmodels 0.12 vs numpy>=1.24

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from statsmodels.tsa.statespace.kalman_smoother import KalmanSmoother  # noqa: E402

import workloads as W  # noqa: E402


def run_statsmodels(y, a0, P0, T, Z, R, H, Q):
    n, p = y.shape
    m, r = T.shape[0], R.shape[1]
    ks = KalmanSmoother(k_endog=p, k_states=m, k_posdef=r)
    ks.bind(np.asfortranarray(y.T.copy()))
    ks["design"] = Z
    ks["obs_cov"] = H
    ks["transition"] = T
    ks["selection"] = R
    ks["state_cov"] = Q
    ks.initialize_known(a0, P0)
    ks.tolerance = 0.0  # never switch to the converged (steady-state) shortcut
    res = ks.smooth()
    mv = lambda x: np.ascontiguousarray(np.moveaxis(x, -1, 0))  # noqa: E731  time first
    return dict(
        llf_obs=np.array(res.llf_obs),
        filtered_state=mv(res.filtered_state),
        predicted_state=mv(res.predicted_state),
        filtered_state_cov=mv(res.filtered_state_cov),
        predicted_state_cov=mv(res.predicted_state_cov),
        smoothed_state=mv(res.smoothed_state),
        smoothed_state_cov=mv(res.smoothed_state_cov),
    )


def make_case(fname, w, nan_rows=(), nan_cells=()):
    y = w["y"].copy()
    for t in nan_rows:
        y[t, :] = np.nan
    for t, i in nan_cells:
        y[t, i] = np.nan
    outs = [run_statsmodels(y, w["a0"][b], w["P0"][b], w["T"][b], w["Z"][b], w["R"][b],
                            w["H"][b], w["Q"][b]) for b in range(w["B"])]
    stacked = {k: np.stack([o[k] for o in outs]) for k in outs[0]}
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, y=y, a0=w["a0"], P0=w["P0"], T=w["T"], Z=w["Z"], R=w["R"],
                        H=w["H"], Q=w["Q"], **stacked)
    print(f"{fname:32s} B={w['B']} n={w['n']} p={w['p']} m={w['m']} r={w['r']} "
          f"llf[0]={stacked['llf_obs'][0].sum()!r}  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    # the reference's only numeric parity case, with and without missing rows
    make_case("nile_llt.npz", W.nile_local_linear_trend(B=1))
    make_case("nile_llt_missing.npz", W.nile_local_linear_trend(B=1), nan_rows=(3, 17, 42, 60, 99))
    # BASELINE config 1
    make_case("nile_local_level.npz", W.nile_local_level(B=3))
    # config-shaped single-GPU families at reduced n / B
    make_case("arma11.npz", W.arma11(B=4, n=120))
    make_case("structural_m10.npz", W.structural(B=3, n=100))
    make_case("structural_m10_missing.npz", W.structural(B=2, n=80), nan_rows=(0, 7, 8, 40, 79))
    make_case("varmax_m10.npz", W.varmax(B=2, n=60, order=(2, 0)))
    make_case("varmax_m15.npz", W.varmax(B=2, n=50, order=(2, 1)))
    # p > 1 dense, full H, with whole-row and partial missing
    make_case("dense_p3m4r2.npz", W.random_dense(3, 60, 3, 4, 2, seed=11, h_diag=False))
    make_case("dense_p3m4r2_diagH_partial.npz", W.random_dense(3, 60, 3, 4, 2, seed=12),
              nan_rows=(5, 30), nan_cells=((2, 0), (9, 1), (9, 2), (41, 2), (59, 0)))
    make_case("dense_p3m4r2_fullH_partial.npz",
              W.random_dense(2, 40, 3, 4, 2, seed=13, h_diag=False),
              nan_rows=(4,), nan_cells=((1, 1), (8, 0), (8, 2), (39, 1)))
    make_case("dense_p2m6r3.npz", W.random_dense(2, 50, 2, 6, 3, seed=14))


if __name__ == "__main__":
    main()
