"""A/B timing of two builds of libkf_hip.so on one box (KF_LIB_PATH): the block forward kernel's instantiations at the shapes that use
them.  usage: ab_time.py <lib_a.so> <lib_b.so>   (each library is timed in its own process)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) == 3:
    for lib in sys.argv[1:]:
        env = dict(os.environ, KF_LIB_PATH=os.path.abspath(lib))
        print("====", lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
    sys.exit(0)

sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


def ms(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run(name, w, grad=False, **kw):
    dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    full = ms(lambda: eng.forward(*dv, mvn_const="exact", **kw))
    lp = ms(lambda: eng.forward(*dv, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, **kw))
    line = f"{name:34s} full {full:7.3f} ms   logp only {lp:7.3f} ms"
    if grad:
        g = ms(lambda: eng.value_and_grad(*dv, mvn_const="exact", **kw), reps=10)
        line += f"   value_and_grad {g:7.3f} ms"
    print(line, flush=True)


run("headline 4096 x 1000 m=10", W.structural(B=4096, n=1000), grad=True)
run("c3 8192 x 2000 m=10", W.structural(B=8192, n=2000))
run("512 x 1000 m=10 (chunked)", W.structural(B=512, n=1000))
run("2048 x 1000 m=10", W.structural(B=2048, n=1000))
run("dense 4096 x 500 m=6 r=2", W.random_dense(4096, 500, 1, 6, 2, seed=5))
run("dense 4096 x 500 m=8 r=2", W.random_dense(4096, 500, 1, 8, 2, seed=5))
run("dense 8192 x 500 m=2 r=1", W.random_dense(8192, 500, 1, 2, 1, seed=5))
run("dense 2048 x 500 m=13 r=2", W.random_dense(2048, 500, 1, 13, 2, seed=5))
