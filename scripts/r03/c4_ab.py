"""C4 forward (full tapes / logp only): in-kernel steady phase (KF_B4Q_SPLIT=0) against the split phase, chunk sizes, prep stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS["c4"]()
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
kw = dict(filter_type="standard", mvn_const="exact", h_diagonal=True)
res = eng.forward(*dev, **kw)
lo = eng.forward(*dev, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw)
def ms(fn, reps=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def both(tag):
    f = ms(lambda: eng.forward(*dev, out=res, **kw))
    l = ms(lambda: eng.forward(*dev, out=lo, store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
    print(f"{tag}: full tapes {f:.2f} ms, logp only {l:.2f} ms", flush=True)
for rep in range(2):
    os.environ["KF_B4Q_SPLIT"] = "0"; both("in-kernel steady phase")
    os.environ["KF_B4Q_SPLIT"] = "1"
    for ch in (2048, 4096, 8192, 16384):
        for ps in ("3", "1"):
            os.environ["KF_B4Q_CHUNK"] = str(ch); os.environ["KF_B4Q_PREP_STREAM"] = ps
            both(f"split, chunk {ch}, prep on stream {ps}")
print("exact:", ms(lambda: eng.forward(*dev, out=res, exact_recursion=True, **kw)))
