"""C5 share (128 x 730, m = 367) forward: a_hat / mirror folded into the products (default) against the separate kernels (KF_BIG_FUSE=0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS["c5"](B=128)
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
kw = dict(filter_type="standard", mvn_const="exact", store_ll_obs=True, store_filtered=False, store_predicted=False)
res = {}
for rep in range(2):
    for fuse in ("1", "0"):
        os.environ["KF_BIG_FUSE"] = fuse
        r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        print(f"KF_BIG_FUSE={fuse}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
        res[fuse] = r.logp.clone()
print("max rel dlogp fused vs separate:", ((res["1"] - res["0"]).abs() / res["0"].abs()).max().item())
