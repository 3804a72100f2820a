"""Clock ramp of an idle MI355X under the headline forward step: per-launch kernel time (HIP events) next to the shader clock
and socket power read from the amdgpu hwmon files, from a cold (deep-sleep) start.  Evidence for bench.py's clock spin-up."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

torch.cuda.set_device(0)
eng = KalmanEngine(torch.device("cuda", 0))
w = W.WORKLOADS["headline"]()
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in bench.KEYS]
kw = dict(filter_type="standard", mvn_const="exact")
res = eng.forward(*d, **kw)
torch.cuda.synchronize()
tel = bench.GpuTelemetry(0)
print(f"# hwmon: {tel.freq}  dpm max {tel.fmax_mhz} MHz")
for idle_s in (3.0, 0.5):
    time.sleep(idle_s)
    f0, p0 = tel.read()
    print(f"# after {idle_s} s idle: sclk {f0} MHz, {p0} W")
    N = 400
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    rows = []
    t0 = time.perf_counter()
    for k in range(N):
        ev[k][0].record()
        eng.forward(*d, out=res, **kw)
        ev[k][1].record()
        if k % 4 == 3:
            ev[k][1].synchronize()
            f, p = tel.read()
            rows.append((k, (time.perf_counter() - t0) * 1e3, f, p))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print("launch  t_ms   kernel_ms(avg of 4)  sclk_mhz  power_w  kernel_Mcycles")
    for k, t, f, p in rows:
        if k < 80 or k % 40 == 39:
            km = sum(ms[k - 3:k + 1]) / 4
            print(f"{k:5d} {t:7.1f} {km:8.4f} {f if f is None else round(f):>8} {p if p is None else round(p):>6} "
                  f"{'' if f is None else round(km * 1e-3 * f, 4)}")
    print(f"# mean of the last 100 launches: {sum(ms[-100:]) / 100:.4f} ms; of the first 20: {sum(ms[:20]) / 20:.4f} ms")
