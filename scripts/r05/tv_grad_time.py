"""Time-varying models at the headline shape: forward, logp + gradient and smoother against the static calls."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = W.structural(B=B, n=1000)
n = w["n"]
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")

def ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for tv in ((), ("Z",), ("T",), ("T", "Z", "H")):
    args = [eng.to_device(w["y"])]
    for k in KEYS:
        x = torch.as_tensor(w[k], device=eng.device)
        if k in tv:
            x = x[:, None].expand(B, n, *x.shape[1:]).contiguous()
        args.append(x)
    kw = dict(mvn_const="exact", time_varying=tv)
    f = ms(lambda: eng.forward(*args, **kw))
    res = eng.forward(*args, **kw)
    gl = torch.ones(B, dtype=torch.float64, device=eng.device)
    try:
        if tv:   # (kf_filter_value_and_grad takes static models: forward with the predicted tapes, then kf_filter_backward)
            g = f + ms(lambda: eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl, **kw))
        else:
            g = ms(lambda: eng.value_and_grad(*args, mvn_const="exact"))
    except Exception as e:  # noqa: BLE001
        g = repr(e)[:160]
    try:
        s = ms(lambda: eng.smooth(args[3], args[5], args[7], res.filtered_states, res.filtered_covs, time_varying=tuple(k for k in tv if k in ("T", "R", "Q"))))
    except Exception as e:  # noqa: BLE001
        s = repr(e)[:120]
    print(f"B={B} time-varying {tv}: forward {f:.2f} ms, logp + gradient {g if isinstance(g, str) else round(g, 2)} ms, smoother {s if isinstance(s, str) else round(s, 2)} ms", flush=True)
    del args, res
    torch.cuda.empty_cache()
