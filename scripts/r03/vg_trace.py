"""A few kf_filter_value_and_grad calls on C4 (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
w = W.WORKLOADS["c4"](B=B)
dev = [eng.to_device(w[x]) for x in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
for _ in range(4):
    eng.value_and_grad(*dev, filter_type="standard", mvn_const="exact", h_diagonal=True)
torch.cuda.synchronize()
