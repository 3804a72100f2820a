import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS["headline"]()
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
res = eng.forward(*dev, store_ll_obs=False, store_predicted=False)
fP = res.filtered_covs
fPc = fP[:, -1:].expand(-1, fP.shape[1], -1, -1).contiguous()
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for s in ("1", "0"):
    os.environ["KF_SMOOTHER_STEADY"] = s
    print("steady", s, "real tape", timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, fP)),
          "constant tape", timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, fPc)), flush=True)
# how many steps of the real tape are within 1e-14 x max of the last one
d = (fP - fP[:, -1:]).abs().flatten(2).max(2).values / fP[:, -1].abs().flatten(1).max(1).values[:, None]
ok = (d <= 1e-14).int()
run = torch.flip(ok, dims=[1]).cumprod(1).sum(1).float()
print("trailing steps within 1e-14 of the last: mean", run.mean().item(), "median", run.median().item(), "min", run.min().item())
# latency or throughput?  the same tapes at 1, 2, 4 wavefronts per SIMD (1024 / 2048 / 4096 draws), every step in full
os.environ["KF_SMOOTHER_STEADY"] = "0"
for nb in (1024, 2048, 4096):
    a = [x[:nb].contiguous() for x in (dev[3], dev[5], dev[7], res.filtered_states, fP, fPc)]
    print(nb, "draws, every step in full: real tape", timeit(lambda: eng.smooth(*a[:5])), "constant tape", timeit(lambda: eng.smooth(a[0], a[1], a[2], a[3], a[5])), flush=True)
# the slowest draws alone: the 1024 draws with the shortest converged stretch
idx = torch.argsort(run)[:1024]
a = [x[idx].contiguous() for x in (dev[3], dev[5], dev[7], res.filtered_states, fP)]
print("1024 least converged draws, every step in full:", timeit(lambda: eng.smooth(*a)), flush=True)
os.environ["KF_SMOOTHER_STEADY"] = "1"
print("1024 least converged draws, shortcut:", timeit(lambda: eng.smooth(*a)), flush=True)
