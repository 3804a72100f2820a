"""Headline-shape run of the three entry points (forward, gradient, smoother) for rocprofv3:
   cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 $ROOT/scripts/profile_ops.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W
from pymc_statespace_amd import KalmanEngine

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
eng = KalmanEngine()
w = W.WORKLOADS[name]() if name != "c4_shard" else W.WORKLOADS["c4"](B=2048)
d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
ones = torch.ones(w["B"], device="cuda", dtype=torch.float64)
res = None
for _ in range(5):
    res = eng.forward(*d, out=res, h_diagonal=True)
    g = eng.backward(*d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=ones,
                     h_diagonal=True)
    sa, sP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
    vg = eng.value_and_grad(*d, h_diagonal=True)
torch.cuda.synchronize()
print("done", name)
