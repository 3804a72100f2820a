"""Where does the pair kernel's step go?  Instrumented builds of kf_forward_b4p.hip (scripts/r03/variants/*.so, same ABI;
results of the NOWAIT / FREE / NOSTORE builds are incomplete by construction):
  NOWAIT  the recursion wavefront alone (the writer returns at once, the recursion never waits for a slot)
  FREE    both wavefronts, no handshake (each runs at its own pace)
  NOSTORE the handshake without the writer's global stores"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for B in (256, 512, 1024):
    w = W.WORKLOADS["headline"](B=B)
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev, mvn_const="exact")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.forward(*dev, mvn_const="exact", out=res)
    torch.cuda.synchronize()
    print(f"   B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
''' % ROOT
for v in ("", "NOWAIT", "FREE", "NOSTORE"):
    env = dict(os.environ)
    if v:
        env["KF_LIB_PATH"] = os.path.join(ROOT, "scripts", "r03", "variants", f"libkf_b4p_{v}.so")
    print(v or "as shipped", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env)
