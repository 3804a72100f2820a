import sys, time
import torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for name, B in (("headline", 4096), ("c4", 2048), ("c2", 4096)):
    w = W.WORKLOADS[name]()
    dev = [eng.to_device(w["y"])] + [eng.to_device(w[k][:B]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev, store_ll_obs=False, store_predicted=False, h_diagonal=True)
    out = {}
    for kern in ("lds", "auto"):
        eng.set_forward_kernel(kern)
        fn = lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
        sa, sP = fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); sa, sP = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[kern] = (sa, sP)
        print(f"{name} B={B} smoother {kern}: {min(ts)*1e3:.2f} ms", flush=True)
    eng.set_forward_kernel("auto")
    da = (out["lds"][0] - out["auto"][0]).abs().max().item(); dP = (out["lds"][1] - out["auto"][1]).abs().max().item()
    print(f"   max |d a_s| {da:.2e}  max |d P_s| {dP:.2e}", flush=True)
    del dev, res, out
    torch.cuda.empty_cache()
