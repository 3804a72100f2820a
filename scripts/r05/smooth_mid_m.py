"""16 <= m <= 64: kf_smoother (on-chip RTS kernels) against kf_smoother_from_filter (HBM-workspace path) over the noise rank."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()

def ms(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = 300
MS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 20, 32, 48, 64]
RS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
for m in MS:
    for r in (RS or [1, 2, m]):
        w = W.random_dense(B, n, 1, m, r, seed=m + r)
        d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
        res = eng.forward(*d, mvn_const="exact", store_ll_obs=False)
        t_rts = ms(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs))
        t_sff = ms(lambda: eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs))
        sa, sP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
        sa2, sP2 = eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)
        dev = float((sP - sP2).abs().max() / sP.abs().max())
        print(f"m={m:2d} r={r:2d} B={B} x {n}: kf_smoother {t_rts:8.2f} ms   kf_smoother_from_filter {t_sff:8.2f} ms   max rel deviation of the covariances {dev:.1e}", flush=True)
        del res, sa, sP, sa2, sP2
        torch.cuda.empty_cache()
