"""C5-shaped run on one GPU's share (B=128, m=367, T=64 by default) for profiling."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W
from pymc_statespace_amd import KalmanEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = KalmanEngine()
w = W.daily_seasonal(B=B, n=n)
d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
kw = dict(store_ll_obs=True, store_filtered=False, store_predicted=False)
res = eng.forward(*d, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.forward(*d, out=res, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} n={n} m=367: {dt*1e3:.1f} ms, {dt/n*1e3:.3f} ms/step, GEMM rate {B*n*4.0*367**3/dt/1e12:.2f} TFLOP/s")
# smoother on the same share (filtered tapes needed): smaller n keeps the tapes in HBM comfortably
ns = min(n, 32)
ws = W.daily_seasonal(B=B, n=ns)
ds = [eng.to_device(ws[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
rs = eng.forward(*ds, store_ll_obs=False, store_predicted=False)
torch.cuda.synchronize()
eng.smooth(ds[3], ds[5], ds[7], rs.filtered_states, rs.filtered_covs)
torch.cuda.synchronize()
t0 = time.perf_counter()
sa, sP = eng.smooth(ds[3], ds[5], ds[7], rs.filtered_states, rs.filtered_covs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"smoother B={B} n={ns} m=367: {dt*1e3:.1f} ms, {dt/(ns-1)*1e3:.3f} ms/step, "
      f"~{B*(ns-1)*10.3*367**3/dt/1e12:.2f} TFLOP/s (10.3 m^3 flop per draw-step)")
# gradient on the same share (needs the predicted tapes)
rg = eng.forward(*ds, store_ll_obs=False, store_filtered=False)
ones = torch.ones(B, device="cuda", dtype=torch.float64)
fn = lambda: eng.backward(*ds, predicted_states=rg.predicted_states, predicted_covs=rg.predicted_covs, g_logp=ones)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
fn()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"gradient B={B} n={ns} m=367: {dt*1e3:.1f} ms, {dt/ns*1e3:.3f} ms/step, "
      f"~{B*ns*6.0*367**3/dt/1e12:.2f} TFLOP/s (3 GEMMs = 6 m^3 flop per draw-step)")
