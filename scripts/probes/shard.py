import sys, time
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS["headline"]()
for B in (512, 1024, 2048, 4096):
    dev = [eng.to_device(w["y"])] + [eng.to_device(w[k][:B]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    for mode in ("logp", "tapes"):
        kw = {} if mode == "tapes" else dict(store_ll_obs=False, store_filtered=False, store_predicted=False)
        out = eng.forward(*dev, **kw); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = eng.forward(*dev, out=out, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"headline shard B={B} {mode}: {min(ts)*1e3:.3f} ms", flush=True)
