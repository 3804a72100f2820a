"""C4 with a dense H: wall time of kf_filter_value_and_grad per call (back-to-back calls, one sync at the end) against the diagonal-H call."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.varmax()
rng = np.random.default_rng(0)
p = w["p"]
A = 0.1 * rng.standard_normal((w["B"], p, p))
Hd = w["H"] + A @ np.swapaxes(A, 1, 2)
for name, H, kw in (("diagonal H", w["H"], dict(h_diagonal=True)), ("dense H", Hd, {})):
    d = [eng.to_device(w["y"])] + [eng.to_device(x) for x in (w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])]
    for _ in range(3):
        out = eng.value_and_grad(*d, mvn_const="exact", **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = eng.value_and_grad(*d, mvn_const="exact", **kw)
    torch.cuda.synchronize()
    print(name, "value_and_grad", round((time.perf_counter() - t0) * 100, 3), "ms per call")
    t0 = time.perf_counter()
    for _ in range(10):
        out = eng.value_and_grad(*d, mvn_const="exact", **kw)
        torch.cuda.synchronize()
    print(name, "value_and_grad, sync after every call", round((time.perf_counter() - t0) * 100, 3), "ms per call")
