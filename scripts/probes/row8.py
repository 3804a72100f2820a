import sys, subprocess
code = r'''
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
m, B, f, pr, ll = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1", sys.argv[4] == "1", sys.argv[5] == "1"
eng = KalmanEngine()
w = W.random_dense(64, 200, 1, m, m // 2, seed=7)
rep = B // 64
dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
eng.set_forward_kernel("mfma")
out = eng.forward(*dev, store_filtered=f, store_predicted=pr, store_ll_obs=ll)
torch.cuda.synchronize()
print("ok", out.logp[:1].tolist())
'''
for args in (("8", "32768", "1", "0", "0"), ("8", "32768", "0", "1", "0"), ("8", "32768", "0", "0", "1"),
             ("8", "16384", "1", "1", "1"), ("8", "24576", "1", "1", "1"), ("9", "32768", "1", "1", "1"), ("10", "32768", "1", "1", "1")):
    r = subprocess.run([sys.executable, "-c", code, *args], capture_output=True, text=True)
    print(args, (r.stdout.strip().splitlines() or ["-"])[-1], "|", (r.stderr.strip().splitlines() or ["-"])[-1][:100], flush=True)
