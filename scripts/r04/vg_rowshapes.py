"""logp + gradient at shapes where AUTO keeps the packed row / lane kernels for the FORWARD (large batches of small models): the call as
AUTO runs it against the block kernels forced (tile tape + steady-state adjoint)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for m, B in ((4, 16384), (4, 32768), (6, 16384), (6, 32768), (8, 16384), (8, 32768), (3, 4096), (3, 16384), (2, 16384)):
    w = W.random_dense(B, 500, 1, m, 2, seed=m)
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    kw = dict(filter_type="standard", mvn_const="exact")
    name = eng.forward_kernel_name(B, 500, 1, m, 2)
    t_auto = ms(lambda: eng.value_and_grad(*d, **kw))
    eng.set_forward_kernel("b4")
    try:
        t_b4 = ms(lambda: eng.value_and_grad(*d, **kw))
        t_b4s = ms(lambda: eng.value_and_grad(*d, exact_recursion=False, **kw))
    finally:
        eng.set_forward_kernel("auto")
    print(f"m={m} B={B} x 500 (forward AUTO: {name}): logp + gradient AUTO {t_auto:.3f} ms   block kernels {t_b4:.3f} ms   block kernels, steady-state "
          f"adjoint forced {t_b4s:.3f} ms", flush=True)
