#!/usr/bin/env python3
"""bench.py -- Kalman log-likelihood throughput on MI355X (contract: see the round brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
                    [--workload headline|c2|c3|c4|c4_m10|c5|c5_dense|weekly|hourly]

A "step" is ONE pass of the hot path over one batch of synthetic draws already resident in
HBM: the fused forward filter (reference pymc_statespace/filters/kalman_filter.py:126-252)
for every draw, writing the same six outputs the reference graph registers
(filtered/predicted states + covariances, logp, ll_obs -- core/statespace.py:164-174), i.e.
the "reference-equivalent op" of SURVEY.md section 8(d), which is HBM-write bound
(workload c5, m = 367: logp only -- its tapes are 0.8 GB per draw; its sparse T runs the CSR gather path,
HBM bound on the covariance it streams per step; a dense T of that size would be MFMA bound).

Multi-GPU (`--gpus N`, N > 1): one process per GPU over RCCL.  Started without a launcher
(no WORLD_SIZE in the environment) this script spawns `python -m torch.distributed.run
--nproc-per-node N` on itself BEFORE anything touches the GPU and exits with its code; under
a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  The draw axis is sharded through
`pymc_statespace_amd.distributed.ShardedKalman` (contiguous shards, no data-path exchange)
and the per-draw log-likelihoods are all-gathered inside the timed region -- the loop this
axis replaces is the per-draw Python loop of reference core/statespace.py:230-236.
  --scaling weak   (default) every GPU filters the workload's B draws: global batch N x B
  --scaling strong the workload's B draws in total, B / N per GPU
A weak run on N > 1 GPUs also reports a `strong_scaling` object: the same global batch as
the 1-GPU run, sharded over the N GPUs, next to rank 0 filtering all of it alone.

Prints ONE JSON line on rank 0.  `value` = draws filtered per second over all ranks
("evals/s": one evaluation = one draw's full T-step filter).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured copy)
PARITY_REL_FAIL = 1e-10   # the run exits non-zero when any draw's |dlogp| / max(1, |logp|) exceeds this
FP64_MFMA_PEAK_TF = 78.6  # dense fp64 matrix peak; v_mfma_f64_16x16x4_f64 measured at 45.2 TF with
#                           register operands only (profiles/r01/fp64_issue_rates.txt)

DEFAULT_FILTER = {"headline": "standard", "c2": "standard", "c3": "cholesky", "c4": "standard",
                  "c4_m10": "standard", "c1": "standard", "c5": "standard", "c5_dense": "standard"}
# draws per GPU when --batch is not given: the workload's own B, except C5 whose BASELINE batch
# (1024) is quoted over 8 GPUs -- one GPU's share is 128 draws (0.28 GB of inputs)
DEFAULT_BATCH = {"c5": 128, "c5_dense": 128}
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


def algorithmic_bytes(B, n, p, m, r, full_tape=True):
    """bytes one launch must move (fp64)."""
    inp = 8 * (p * m + 2 * m * m + m * r + r * r + p * p + m)
    out = 8 * (n * m + (n + 1) * m + n * m * m + (n + 1) * m * m + n + 1) if full_tape else 8
    return B * (inp + out) + 8 * n * p


def algorithmic_flops(B, n, p, m, r):
    """dense flop count of the reference's own operation sequence
    (Standard filter, Joseph form), per draw-step, times draws x steps."""
    per_step = (8 * m ** 3 + m * m * (6 * p + 2 * r + 7) + m * (6 * p * p + 2 * r * r + 4 * p + 2)
                + 3 * p ** 3 + 3 * p * p + 4 * p)
    return float(per_step) * B * n


def executed_flops_big(B, n, m, T=None):
    """What the workspace path actually issues on the matrix cores per launch: two m^3 products per
    draw-step -- W = T P_f, S = W T' + R Q R' on the 128 x 64 tiles that touch the upper triangle only
    (kf_forward_big.hip: a tile is skipped when its first row lies below its last column) -- and, round 5, of every
    product only the 32-deep K chunks whose 64-row block of the static factor T holds a non-zero (big_tmask_kernel:
    m <= 1024, KF_BIG_TMASK=0 switches it off).  ``T`` (B, m, m): the draws' transition matrices, to count those chunks."""
    import numpy as np

    tm, tn, kc = -(-m // 128), -(-m // 64), -(-m // 32)
    kept = [(i, j) for i in range(tm) for j in range(tn) if not (128 * i > 64 * j + 63)]
    fw, fs = 1.0, len(kept) / (tm * tn)
    if T is not None and m <= 1024 and os.environ.get("KF_BIG_TMASK", "1")[:1] != "0":
        Tn = np.asarray(T)
        pad = np.zeros((Tn.shape[0], 64 * tn, 32 * kc), dtype=bool)
        pad[:, :m, :m] = (Tn != 0) | np.isnan(Tn)
        nz = pad.reshape(Tn.shape[0], tn, 64, kc, 32).any(axis=(2, 4))          # (B, 64-row block, chunk)
        rows = np.stack([nz[:, 2 * i] | (nz[:, 2 * i + 1] if 2 * i + 1 < tn else False) for i in range(tm)], axis=1)
        fw = float(rows.sum(axis=2).mean()) / kc                                   # W: A = T, 128-row tiles
        fs = float(sum(nz[:, j].sum(axis=1).mean() for _, j in kept)) / (kc * tm * tn)   # S: op(B) = T', 64-column tiles
    return (2.0 * fw + 2.0 * fs) * m ** 3 * B * n


def usable_cores():
    """Cores this process may actually run on: the scheduler affinity mask, capped by the cgroup CPU
    quota when one is set (os.cpu_count() reports the machine, not the share a container gets)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


class GpuTelemetry:
    """Shader clock and socket power of the GPU this rank runs on, from the amdgpu hwmon files an ordinary user can read
    (`/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input` in Hz, `power1_input` in uW; what `rocm-smi --showclocks
    --showpower` prints).  The forward kernels are issue-bound, so their time follows the shader clock: an idle MI355X sits in
    deep sleep (~100 MHz) and takes tens of milliseconds of load to reach its sustained clock (profiles/r04/clock_ramp.txt),
    longer than the driver's 5 warm-up steps -- bench.py therefore spins the clocks up before the W warm-up steps (untimed,
    reported as `clock_spinup`) and samples sclk / power over the timed region (`roofline.sclk_mhz`)."""

    def __init__(self, torch_index=0):
        import glob
        self.freq = self.power = None
        self.fmax_mhz = None
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:  # noqa: BLE001
            want = None
        cands = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            dev = os.path.realpath(os.path.join(os.path.dirname(f), "..", ".."))
            cands.append((os.path.basename(dev), f))
        pick = [f for bdf, f in cands if want and bdf.startswith(want)] or [f for _, f in cands[:1]]
        if pick:
            self.freq = pick[0]
            pw = os.path.join(os.path.dirname(pick[0]), "power1_input")
            self.power = pw if os.path.exists(pw) else None
            try:
                lv = open(os.path.join(os.path.dirname(pick[0]), "..", "..", "pp_dpm_sclk")).read()
                self.fmax_mhz = max(float(x.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                                    for x in lv.strip().splitlines())
            except Exception:  # noqa: BLE001
                self.fmax_mhz = None
        self._stop = None
        self.samples = []

    def read(self):
        """(sclk MHz, socket W) now, or (None, None)."""
        try:
            f = float(open(self.freq).read()) / 1e6 if self.freq else None
            p = float(open(self.power).read()) / 1e6 if self.power else None
            return f, p
        except Exception:  # noqa: BLE001
            return None, None

    def start(self):
        import threading
        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self.read())
                time.sleep(0.002)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        """Summary of the samples taken since start()."""
        if self._stop is None:
            return {}
        self._stop.set()
        self._thr.join()
        fs = [f for f, _ in self.samples if f is not None]
        ps = [p for _, p in self.samples if p is not None]
        if not fs:
            return {"sclk_mhz": None, "note": "no readable amdgpu hwmon freq1_input on this box"}
        out = {"sclk_mhz": sum(fs) / len(fs), "sclk_mhz_min": min(fs), "sclk_mhz_max": max(fs), "samples": len(fs),
               "sclk_mhz_dpm_max": self.fmax_mhz,
               "source": "amdgpu hwmon freq1_input / power1_input sampled every ~2 ms over the timed region"}
        if ps:
            out["socket_power_w"] = sum(ps) / len(ps)
        return out


def clock_spinup(fn, sync, tel, max_s=2.0, enabled=True):
    """Untimed: repeat the step until the shader clock has left deep sleep and two consecutive readings agree within 2 %
    (or max_s).  Without telemetry: 0.4 s of the same work."""
    if not enabled:
        return {"enabled": False}
    t0 = time.perf_counter()
    last, n, f = None, 0, None
    per = 1   # steps between clock readings: ~10 ms of work (8 steps at the headline, 1 for C5's 0.4 - 3.4 s passes)
    while time.perf_counter() - t0 < max_s:
        t1 = time.perf_counter()
        for _ in range(per):
            fn()
        sync()
        n += per
        per = max(1, min(8, int(0.01 / max(1e-6, (time.perf_counter() - t1) / per))))
        f, _ = tel.read()
        if f is None:
            if time.perf_counter() - t0 > 0.4:
                break
            continue
        if last is not None and f > 400.0 and abs(f - last) <= 0.02 * f and time.perf_counter() - t0 > 0.15:
            break
        last = f
    return {"enabled": True, "steps": n, "ms": (time.perf_counter() - t0) * 1e3, "sclk_mhz_at_end": f}


def cpu_baseline(w, filter_type, full_tape=True, target_s=12.0, n_sample=None, n_steps=None, mvn_const="exact"):
    """The C restatement of the reference algorithm (oracle/kf_oracle.c, kind "port") on the
    host cores, on the leading draws of the SAME workload, computing the same outputs.
    Built here, on the box that times it, with -O3 -march=native (oracle/Makefile `native`).

    The sample is processed in chunks whose output tapes reuse one pre-touched host buffer
    (first-touch page faults of multi-GB fresh arrays would otherwise dominate the CPU time;
    recycling the buffer only favours the CPU)."""
    import numpy as np

    from oracle import c_oracle as co

    flavour = co.use_native()
    cores, quota = usable_cores()
    B = w["B"]
    chunk = min(max(8, 4 * cores), B)
    n_all = w["n"]
    if n_steps is not None and n_steps < n_all:
        # large k_states (C5: 0.4 GFLOP per draw-step in scalar loops): time the first n_steps steps of each
        # sampled draw and scale to whole evaluations -- the recursion costs the same at every step
        w = dict(w, y=w["y"][:n_steps], n=n_steps)
        chunk = min(chunk, cores)

    def run(lo, hi, out=None, full=full_tape):
        sl = slice(lo, hi)
        return co.kalman_filter(w["y"], w["a0"][sl], w["P0"][sl], w["T"][sl], w["Z"][sl],
                                w["R"][sl], w["H"][sl], w["Q"][sl], filter_type=filter_type,
                                mvn_const=mvn_const, full=full, nthreads=cores, out=out)

    buf = run(0, chunk)          # untimed: loads the library, touches the tape buffer
    t0 = time.perf_counter()
    run(0, chunk, out=buf)       # calibration pass
    rate = chunk / (time.perf_counter() - t0)
    if n_sample is None:
        n_sample = int(min(B, max(chunk, rate * target_s)))
    n_sample = min(B, max(chunk, (n_sample // chunk) * chunk))
    logp = np.empty(n_sample)
    t0 = time.perf_counter()
    for lo in range(0, n_sample, chunk):
        hi = min(lo + chunk, n_sample)
        if hi - lo == chunk:
            run(lo, hi, out=buf)
            logp[lo:hi] = buf["logp"]
        else:
            logp[lo:hi] = run(lo, hi)["logp"]
    dt = time.perf_counter() - t0
    frac = w["n"] / n_all
    out = dict(value=n_sample * frac / dt, unit="evals/s", cores=cores, kind="port",
               cores_note=f"threads used = min(sched_getaffinity = {len(os.sched_getaffinity(0))}, cgroup cpu quota = "
                          f"{'none' if quota is None else f'{quota:.1f}'}); os.cpu_count() = {os.cpu_count()}",
               sample=f"first {n_sample} draws x {'all' if frac == 1.0 else 'the first'} {w['n']} "
                      f"{'' if frac == 1.0 else f'of {n_all} '}steps of the same workload"
                      f"{'' if frac == 1.0 else ' (scaled to whole evaluations)'}, "
                      f"{'full tapes' if full_tape else 'logp only'}, {dt:.2f} s wall, "
                      f"oracle/kf_oracle.c ({flavour}), OpenMP over draws on {cores} threads")
    if full_tape:
        # the same sample with logp as the only output (what a sampler consumes): no tape stores
        n2 = min(n_sample, 4 * chunk)
        t0 = time.perf_counter()
        for lo in range(0, n2, chunk):
            run(lo, min(lo + chunk, n2), full=False)
        out["value_logp_only"] = n2 * frac / (time.perf_counter() - t0)
    return logp, out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """--gpus N without a launcher: start N ranks of this script (one per GPU).  Runs in a parent
    that has imported neither torch nor HIP: a process that has touched the GPU must not spawn or
    exec GPU children on this pool."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--filter", default=None)
    ap.add_argument("--batch", type=int, default=None,
                    help="draws per GPU (weak) / in total (strong); default: the workload's B")
    ap.add_argument("--logp-only", action="store_true", help="skip the tapes (what NUTS consumes)")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-ops", action="store_true", help="skip the untimed extras (logp only, logp + gradient, smoother)")
    ap.add_argument("--exact-recursion", action="store_true",
                    help="KF_FLAG_EXACT_RECURSION: the covariance recursion at every step (no steady-state phase)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-strong-probe", action="store_true",
                    help="N > 1, weak: skip the extra strong-scaling measurement")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--time-varying", default="", metavar="NAMES",
                    help="comma-separated subset of T,Z,R,H,Q,c,d: those matrices get a time axis (B,n,.,.) -- the same values at "
                         "every step, so results equal the static run's -- and the call takes the time-varying path "
                         "(KF_FLAG_TV_*, SURVEY 8(f) row 3; reference filters/kalman_filter.py:79-124)")
    ap.add_argument("--dense-h", action="store_true",
                    help="p > 1: replace the workload's diagonal H by a dense positive definite one (reference "
                         "kalman_filter.py:262-275 takes any PD H); no KF_FLAG_H_DIAGONAL kernels")
    ap.add_argument("--no-spinup", action="store_true",
                    help="skip the untimed clock spin-up before the warm-up steps (measures from a cold, deep-sleep clock)")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import workloads as W
    from pymc_statespace_amd import KalmanEngine
    from pymc_statespace_amd.distributed import ShardedKalman, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("KF_BENCH_SHARE_GPU") == "1":
            # test mode for 1-GPU boxes: every rank on cuda:0, exchange over gloo (RCCL refuses two
            # ranks on one device).  Exercises the N > 1 control flow, not xGMI; never a bench number.
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                sys.exit(f"bench.py: rank {rank} wants cuda:{local_rank} but only "
                         f"{torch.cuda.device_count()} GPUs are visible")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    ft = args.filter or DEFAULT_FILTER.get(args.workload, "standard")
    B0 = args.batch or DEFAULT_BATCH.get(args.workload)
    kw = {} if B0 is None else {"B": B0}
    base = W.WORKLOADS[args.workload](**kw)   # the 1-GPU workload (seeded)
    B0, n, p, m, r = base["B"], base["n"], base["p"], base["m"], base["r"]
    tv = tuple(x for x in args.time_varying.split(",") if x)
    if args.dense_h:
        if p < 2:
            sys.exit("--dense-h needs a workload with p > 1 (c4, c4_m10)")
        rng_h = np.random.default_rng(99)
        Lh = np.tril(0.05 * rng_h.standard_normal((B0, p, p)), -1)
        Lh[:, np.arange(p), np.arange(p)] = np.sqrt(np.einsum("bii->bi", base["H"]))
        base = dict(base, H=Lh @ np.swapaxes(Lh, 1, 2), name=base["name"] + "+denseH")
    big = m > 64                               # HBM-workspace path (C5): logp only, MFMA bound
    full = not (args.logp_only or big)
    strong = args.scaling == "strong"

    eng = KalmanEngine(dev, lanes_per_draw=args.lanes)
    eng.set_exact_recursion(args.exact_recursion)
    sk = ShardedKalman(eng)
    fkw = dict(filter_type=ft, mvn_const="exact", store_ll_obs=full or big, store_filtered=full,
               store_predicted=full)
    if tv:
        fkw["time_varying"] = tv
    if args.dense_h:
        fkw["h_diagonal"] = False

    def shard_of(w, lo, hi):
        out = [eng.to_device(w["y"])]
        for k in KEYS:
            x = eng.to_device(w[k][lo:hi])
            if k in tv:   # the same slice at every step, materialised: (B, n, rows, cols) in HBM, read slice by slice
                x = x[:, None].expand(x.shape[0], n, *x.shape[1:]).contiguous()
            out.append(x)
        if "c" in tv or "d" in tv:
            sys.exit("--time-varying c / d: not wired into bench.py (the workloads have no intercepts)")
        return out

    # ---- the timed configuration -------------------------------------------------------------------
    if strong or world == 1:
        B_global = B0
        lo, hi = shard_bounds(B_global, rank, world)
        d = shard_of(base, lo, hi)
        w_mine, lo_mine = base, lo
    else:
        # weak: rank k holds draws [k B0, (k+1) B0) of a global batch of world x B0 draws; the shard is
        # generated from its own seed so that no rank materialises the whole batch on the host
        B_global = world * B0
        w_mine = W.WORKLOADS[args.workload](**{**kw, "seed": W.SEED_DRAWS + 1000 + rank}) if rank else base
        lo, hi = rank * B0, (rank + 1) * B0
        d = shard_of(w_mine, 0, B0)
        lo_mine = 0
    B_local = hi - lo
    res_holder = {}

    def one_step(inputs, Bg, out=None):
        g, res = sk.forward(*inputs, B_global=Bg, out=out, **fkw)
        return g, res

    gathered, res = one_step(d, B_global)   # allocates the output tapes once; reused by every step
    torch.cuda.synchronize()

    tel = GpuTelemetry(dev.index)
    spin = {}
    tel_timed = {}

    def timed(inputs, Bg, out, steps, warmup, main=False):
        if main:
            spin.update(clock_spinup(lambda: sk.forward_local(*inputs, out=out, **fkw), torch.cuda.synchronize, tel,
                                     enabled=not args.no_spinup))
        for _ in range(warmup):
            one_step(inputs, Bg, out)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(steps)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if main:
            tel.start()
        t0 = time.perf_counter()
        for k in range(steps):
            # the events bracket the filter launch only; the all-gather of the per-draw
            # log-likelihoods is outside them but inside the wall time
            ev[k][0].record()
            sk.forward_local(*inputs, out=out, **fkw)
            ev[k][1].record()
            res_holder["g"] = sk.gather_logp(out.logp, Bg)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if main:
            tel_timed.update(tel.stop())
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64,
                                device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if steps else float("nan")
        return elapsed, kern_ms

    elapsed, kern_ms = timed(d, B_global, res, args.steps, args.warmup, main=True)
    gathered = res_holder["g"]

    # ---- N > 1, weak: the extra strong-scaling measurement (global batch = the 1-GPU run's) -----------
    strong_obj = None
    if world > 1 and not strong and not args.no_strong_probe:
        slo, shi = shard_bounds(B0, rank, world)
        ds = shard_of(base, slo, shi)
        _, rs = one_step(ds, B0)
        ks, kws = max(5, args.steps), 2
        el_s, kern_s = timed(ds, B0, rs, ks, kws)
        g_strong = res_holder["g"]
        del rs, ds
        # rank 0 alone on the full batch (the other ranks wait at the barrier)
        el_1 = None
        if rank == 0:
            d1 = shard_of(base, 0, B0)
            r1 = sk.forward_local(*d1, out=res if B_local == B0 else None, **fkw)
            for _ in range(kws):
                sk.forward_local(*d1, out=r1, **fkw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ks):
                sk.forward_local(*d1, out=r1, **fkw)
            torch.cuda.synchronize()
            el_1 = time.perf_counter() - t0
            dmax = float((g_strong - r1.logp).abs().max().item())
            strong_obj = {
                "global_draws": B0, "draws_per_gpu": shi - slo, "steps": ks,
                "ms_per_step": el_s / ks * 1e3, "value": B0 * ks / el_s, "unit": "evals/s",
                "single_gpu_ms_per_step": el_1 / ks * 1e3,
                "speedup_vs_single_gpu": el_1 / el_s,
                "max_abs_dlogp_sharded_vs_single_gpu": dmax,
            }
        dist.barrier()

    # ---- N > 1: the other two entry points a sampler / post-processing step calls, sharded the same way (outside the timed region):
    # logp + gradient (one kf_filter_value_and_grad per rank + ONE all-gather of the packed rows [logp | info | nine gradients]) and the
    # smoother over this rank's own tapes (nothing exchanged); max over ranks of the wall time per call
    multi_ops = None
    if world > 1 and not big and not args.no_ops and not tv:
        def _ms_all(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            el = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64,
                              device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            return float(el.item())

        multi_ops = {}
        try:
            vkw = dict(filter_type=ft, mvn_const="exact", h_diagonal=not args.dense_h)
            multi_ops["logp_and_gradient_gathered_ms"] = _ms_all(lambda: sk.value_and_grad(*d, B_global=B_global, **vkw))
            multi_ops["logp_and_gradient_local_grads_ms"] = _ms_all(
                lambda: sk.value_and_grad(*d, B_global=B_global, gather_grads=False, **vkw))
            if full and B_local > 0:
                multi_ops["rts_smoother_local_ms"] = _ms_all(
                    lambda: sk.smooth_local(d[3], d[5], d[7], res.filtered_states, res.filtered_covs), reps=3)
                if eng.smooth_from_filter_ok(w_mine["p"], w_mine["m"]):
                    multi_ops["smoother_from_filter_local_ms"] = _ms_all(
                        lambda: sk.smooth_from_filter_local(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs,
                                                            filter_type=ft), reps=3)
        except Exception as e:  # noqa: BLE001 -- an extra must never break the contract line
            multi_ops["error"] = str(e)[:200]

    parity_failed = False
    if rank == 0:
        out = {
            "metric": "Kalman logp evals/sec (batch x T x m)",
            "value": B_global * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if os.environ.get("KF_BENCH_SHARE_GPU") != "1"
            else "synthetic; TEST MODE: all ranks share cuda:0 over gloo -- not a measurement",
            "config": {
                "workload": f"{args.workload}: {base['name']} B={B_local} draws/GPU x T={n} x m={m} "
                            f"(p={p}, r={r}), {ft} filter, "
                            f"{'all six reference outputs written to HBM' if full else 'logp only'}",
                "draws_per_gpu": B_local, "T": n, "k_states": m, "k_endog": p, "k_posdef": r,
                "time_varying": list(tv), "dense_h": bool(args.dense_h),
                "filter": ft, "outputs": "full_tape" if full else "logp_only",
                "global_draws": B_global, "parallelism": f"draws sharded x{world}",
                "draw_steps_per_s": B_global * n * args.steps / elapsed,
                "covariance_recursion": "every step (KF_FLAG_EXACT_RECURSION)" if args.exact_recursion else
                "library default (include/kf.h, KF_FLAG_EXACT_RECURSION): every step for p = 1, 4 < m <= 10 below 1536 steps -- the "
                "north-star shape -- and for <= 1024 draws; else until the predicted covariance has converged, then the mean only "
                "with the converged gain; other_entry_points.forward_exact_recursion_ms / forward_steady_state_ms time the same "
                "call with either behaviour forced",
            },
        }
        diag = bool(p > 1 and np.all(base["H"] == base["H"] * np.eye(p)))
        kname = eng.forward_kernel_name(B_local, n, p, m, r, filter_type=ft, h_diagonal=diag, time_varying=tv)
        sparse_T = big and eng.last_forward_kernel() == "bsp_step_kernel"
        if sparse_T:
            # T sparse (the structural models): T P_f T' + R Q R' as a gather over a CSR copy of T -- m^2 nnz-per-row flops, one pass
            # over P.  P does not fit on chip (B_local x m^2 doubles), so every step reads it and writes it once: 2 m^2 doubles per
            # draw and step is what the recursion must move, and HBM is the roofline
            aflops = algorithmic_flops(B_local, n, p, m, r)
            sbytes = 2.0 * 8.0 * B_local * m * m * n
            achieved = sbytes / (kern_ms * 1e-3) / 1e9
            traffic, tsrc = None, None
            try:
                ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{args.workload}:logp:B{B_local}")
                if ent and "bsp_step_kernel" in str(ent.get("kernels", ent.get("kernel"))):
                    traffic, tsrc = ent["bytes"], ent.get("source")
            except Exception:  # noqa: BLE001
                pass
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": tsrc, "kernel": "bsp_step_kernel", "kernel_ms": kern_ms, "launches_per_step": 2 * n,
                "algorithmic_bytes_per_launch": sbytes,
                "reference_flops_per_launch": aflops,
                "reference_flops_tflops": aflops / (kern_ms * 1e-3) / 1e12,
                "note": "kernel_ms = the whole forward pass (2 launches per time step: gain, bsp_step_kernel); the library found every "
                        "draw's T sparse (kf_last_forward_kernel) and evaluates T P_f T' + R Q R' as a CSR gather: achieved = the "
                        "predicted covariance read once and written once per step (2 x 8 m^2 bytes per draw and step -- it cannot "
                        "stay on chip) over that time; traffic = the whole pass's HBM bytes from the counters (profiles/traffic.json); "
                        "KF_BIG_SPARSE=0 runs the two fp64 MFMA products instead (other_entry_points.mfma_products_ms: the "
                        "round-4 path); reference_flops_tflops = SURVEY 8(d)'s dense flop count of the REFERENCE's sequence over "
                        "the same time -- what a dense implementation would have to sustain to match (above the 78.6 TF peak), "
                        "not a utilisation",
            }
        elif big:
            aflops = algorithmic_flops(B_local, n, p, m, r)
            xflops = executed_flops_big(B_local, n, m, w_mine["T"][lo_mine:lo_mine + B_local] if w_mine["T"].ndim == 3 else None)
            ach = xflops / (kern_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TF, "traffic": None,
                "kernel": kname, "kernel_ms": kern_ms,
                "launches_per_step": (3 if (p == 1 and not full) else 7) * n,
                "executed_flops_per_launch": xflops,
                "reference_flops_per_launch": aflops,
                "reference_flops_tflops": aflops / (kern_ms * 1e-3) / 1e12,
                "note": "kernel_ms = the whole forward pass (p = 1 without the filtered covariance tape: 3 launches per time "
                        "step -- gain, W = T P_f with P_f formed on the way into LDS and a_hat beside it, S = W T' + R Q R' "
                        "mirrored in the epilogue; else 7); achieved / frac = the fp64 MFMA flops ACTUALLY ISSUED (S on the tiles that "
                        "touch the upper triangle; round 5: of both products only the K chunks whose 64 x 32 block of T holds "
                        "a non-zero -- 61 % / 38 % of them for the daily-seasonality T) over that time and over the 78.6 TF dense peak; "
                        "reference_flops_tflops = SURVEY 8(d)'s dense flop count of the REFERENCE's sequence (Joseph form, "
                        "no symmetry) over the same time -- what the reference would have to sustain to match, not a "
                        "utilisation (rounds 1-2 reported it as frac)",
            }
        else:
            abytes = algorithmic_bytes(B_local, n, p, m, r, full)
            tv_elems = {"T": m * m, "Z": p * m, "R": m * r, "H": p * p, "Q": r * r}
            abytes += sum(8 * B_local * (n - 1) * tv_elems[k] for k in tv)   # a time-varying matrix is read slice by slice
            achieved = abytes / (kern_ms * 1e-3) / 1e9
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    ent = json.load(open(tfile)).get(f"{args.workload}:{'full' if full else 'logp'}:B{B_local}")
                    if ent and any(k.startswith(kname) for k in ent.get("kernels", [ent.get("kernel", "")])):
                        traffic, tsrc = ent["bytes"], ent.get("source")
                except Exception:  # noqa: BLE001
                    traffic = None
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "kernel": kname, "kernel_ms": kern_ms,
                "algorithmic_bytes_per_launch": abytes,
            }
        out["roofline"].update({k: v for k, v in tel_timed.items() if k != "note"})
        out["roofline"]["clock_spinup"] = dict(spin, note="untimed, BEFORE the W warm-up steps: the same step repeated until the "
                                               "shader clock has left deep sleep and settled (an idle MI355X idles at ~100 MHz and the "
                                               "issue-bound kernels follow sclk); --no-spinup measures from the cold clock")
        if tel_timed.get("sclk_mhz"):
            # the kernel is issue-bound: its time in shader cycles is the box-independent figure
            out["roofline"]["kernel_mcycles_at_sclk"] = kern_ms * 1e-3 * tel_timed["sclk_mhz"]
        if strong_obj is not None:
            out["strong_scaling"] = strong_obj
        if world == 1 and big and not args.no_ops:
            # m > 64 (C5): the smoother and the gradient over a PREFIX of the series (their tapes are m^2 doubles per draw and step:
            # 1.1 MB at m = 367), per time step -- the recursions cost the same at every step.  AFTER and outside the timed region
            ops = {}
            try:
                ns = int(max(6, min(40, 20e9 // (B_local * m * m * 8))))
                d40 = [d[0][:ns].contiguous()] + d[1:]

                def _ms1(fn):
                    fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) * 1e3
                r40 = eng.forward(*d40, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_predicted=False)
                ops["prefix_steps"] = ns
                if 12 * r > m + 11:
                    ops["rts_smoother_ms_per_step"] = _ms1(lambda: eng.smooth(d[3], d[5], d[7], r40.filtered_states,
                                                                             r40.filtered_covs)) / (ns - 1)
                else:
                    # state noise of few columns behind a DENSE T (c5_dense: r = 3, m = 367): P_hat sits on pinv's cut-off at every
                    # step and the reference's recursion redoes each draw with the workgroup eigen-decomposition -- measured once,
                    # 17.3 s PER STEP for 128 draws (profiles/r06/bench_c5_dense.json); not re-timed on every bench run.  The
                    # inverse-free recursion below is what a caller with the filter's tapes gets (1.6 ms per step)
                    ops["rts_smoother_ms_per_step"] = None
                    ops["rts_smoother_note"] = "skipped: rank-deficient state noise behind a dense T -- 17.3 s per step measured once (profiles/r06)"
                del r40
                torch.cuda.empty_cache()
                r40 = eng.forward(*d40, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False)
                ops["forward_with_predicted_tapes_ms_per_step"] = _ms1(lambda: eng.forward(
                    *d40, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False, out=r40)) / ns
                ops["gradient_sweep_ms_per_step"] = _ms1(lambda: eng.backward(
                    *d40, predicted_states=r40.predicted_states, predicted_covs=r40.predicted_covs,
                    g_logp=torch.ones(B_local, dtype=torch.float64, device=dev), filter_type=ft, mvn_const="exact")) / ns
                if eng.smooth_from_filter_ok(p, m):
                    # the smoother from the predicted tapes (kf_smoother_from_filter, workspace path): no inverse of P_hat
                    ops["smoother_from_filter_ms_per_step"] = _ms1(lambda: eng.smooth_from_filter(
                        d40[0], d[3], d[4], d[6], r40.predicted_states, r40.predicted_covs, filter_type=ft)) / ns
                del r40
                torch.cuda.empty_cache()
                ops["logp_and_gradient_one_call_ms_per_step"] = _ms1(lambda: eng.value_and_grad(
                    *d40, filter_type=ft, mvn_const="exact")) / ns
                ops["forward_ms_per_step"] = kern_ms / n
                if sparse_T:   # the same prefix on the two MFMA products (the path a dense T takes)
                    os.environ["KF_BIG_SPARSE"] = "0"
                    try:
                        ops["mfma_products_ms_per_step"] = _ms1(lambda: eng.forward(
                            *d40, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)) / ns
                    finally:
                        del os.environ["KF_BIG_SPARSE"]
            except Exception as e:  # noqa: BLE001 -- an extra must never break the contract line
                ops["error"] = str(e)[:300]
            out["other_entry_points"] = ops
        if multi_ops is not None:
            out["other_entry_points_sharded"] = multi_ops
        if world == 1 and not big and not args.no_ops:
            # the other entry points on the same resident inputs, AFTER and outside the timed region (not part of `value`):
            # log-likelihood only, logp + gradient in one call (kf_filter_value_and_grad), RTS smoother over the filtered tapes
            def _ms(fn, reps=10):
                fn()
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3
            ops = {}
            try:
                xkw = {k: v for k, v in fkw.items() if k in ("time_varying", "h_diagonal")}
                lo_res = eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False,
                                     store_predicted=False, **xkw)
                ops["logp_only_ms"] = _ms(lambda: eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False,
                                                              store_filtered=False, store_predicted=False, out=lo_res, **xkw))
                # the logp-only call moves almost no bytes -- priced against the fp64 peak with the reference's own dense
                # flop count (algorithmic_flops: what the reference would execute, not what the kernels issue)
                ops["roofline_logp_only"] = {
                    "bound": "fp64", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TF,
                    "achieved": algorithmic_flops(B_local, n, p, m, r) / (ops["logp_only_ms"] * 1e-3) / 1e12,
                    "frac": algorithmic_flops(B_local, n, p, m, r) / (ops["logp_only_ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF,
                    "note": "reference dense flops (SURVEY 8d) / logp_only_ms / fp64 peak"}
                if not tv and B_local <= 1024 and p == 1 and m <= 10 and os.environ.get("KF_FORWARD_CHUNKED", "")[:1] != "0":
                    # small batches (one GPU's share of a sharded job) run the time-chunked pipeline (kf_forward_b4c.hip); the
                    # sequential kernel on the same inputs, for the record
                    os.environ["KF_FORWARD_CHUNKED"] = "0"
                    try:
                        sq_res = eng.forward(*d, **fkw)
                        ops["forward_sequential_kernel_ms"] = _ms(lambda: eng.forward(*d, out=sq_res, **fkw))
                        ops["max_rel_dlogp_sequential_vs_default"] = float(((sq_res.logp - res.logp).abs() / sq_res.logp.abs()).max())
                        ops["logp_only_sequential_kernel_ms"] = _ms(lambda: eng.forward(
                            *d, filter_type=ft, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False,
                            out=lo_res, **xkw))
                        del sq_res
                    finally:
                        del os.environ["KF_FORWARD_CHUNKED"]
                if not tv:
                    ops["logp_and_gradient_one_call_ms"] = _ms(lambda: eng.value_and_grad(
                        *d, filter_type=ft, mvn_const="exact", h_diagonal=not args.dense_h))
                    # the same with the adjoint of the recursion at every step (KF_FLAG_EXACT_RECURSION: no steady-state adjoint)
                    ops["logp_and_gradient_exact_recursion_ms"] = _ms(lambda: eng.value_and_grad(
                        *d, filter_type=ft, mvn_const="exact", h_diagonal=not args.dense_h, exact_recursion=True))
                    # ... and with the steady-state adjoint requested (KF_FLAG_STEADY_STATE), whatever AUTO picked above
                    ops["logp_and_gradient_steady_state_ms"] = _ms(lambda: eng.value_and_grad(
                        *d, filter_type=ft, mvn_const="exact", h_diagonal=not args.dense_h, exact_recursion=False))
                    # the calls above repeat ONE set of parameters, so the placement hint the steady-state path takes from the
                    # previous call's convergence points (include/kf.h; outputs never depend on it) is exact; without it:
                    os.environ["KF_VG_SCHED_HINT"] = "0"
                    try:
                        ops["logp_and_gradient_no_placement_hint_ms"] = _ms(lambda: eng.value_and_grad(
                            *d, filter_type=ft, mvn_const="exact", h_diagonal=not args.dense_h))
                    finally:
                        del os.environ["KF_VG_SCHED_HINT"]
                # the timed call runs the library's default; the same call with the covariance recursion forced at every
                # step (KF_FLAG_EXACT_RECURSION) and with the steady-state phase requested (KF_FLAG_STEADY_STATE), and how far
                # the three log-likelihood vectors are apart
                ex_res = eng.forward(*d, exact_recursion=True, **fkw)
                ops["forward_exact_recursion_ms"] = _ms(lambda: eng.forward(*d, exact_recursion=True, out=ex_res, **fkw))
                ops["max_rel_dlogp_exact_vs_default"] = float(((ex_res.logp - res.logp).abs() / ex_res.logp.abs()).max())
                ss_res = eng.forward(*d, exact_recursion=False, **fkw)
                ops["forward_steady_state_ms"] = _ms(lambda: eng.forward(*d, exact_recursion=False, out=ss_res, **fkw))
                ops["max_rel_dlogp_exact_vs_steady_state"] = float(((ex_res.logp - ss_res.logp).abs() / ex_res.logp.abs()).max())
                del ex_res, ss_res
                if full:
                    tvs = tuple(k for k in tv if k in ("T", "R", "Q"))
                    ops["rts_smoother_ms"] = _ms(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states,
                                                                     res.filtered_covs, time_varying=tvs), reps=3)
                    # the same smoothed moments from the filter's PREDICTED tapes (kf_smoother_from_filter: what the smoother
                    # node behind this library's filter node runs -- op.kalman_smoother_graph; p == 1, m <= 15, static), and how
                    # far the two are apart, relative to each draw's largest entry
                    if not tv and eng.smooth_from_filter_ok(w_mine["p"], w_mine["m"]):
                        sff = lambda: eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states,  # noqa: E731
                                                             res.predicted_covs, filter_type=ft)
                        ops["smoother_from_filter_ms"] = _ms(sff, reps=3)
                        sa, sP = sff()
                        ra, rP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
                        nB = sa.shape[0]
                        dev_ = lambda x, r: float(((x - r).abs().reshape(nB, -1).max(dim=1).values /  # noqa: E731
                                                   (1e-300 + r.abs().reshape(nB, -1).max(dim=1).values)).max())
                        ops["smoother_from_filter_vs_rts_max_rel"] = {"states": dev_(sa, ra), "covariances": dev_(sP, rP)}
                        nbytes = 2 * 8 * (sa.numel() + sP.numel())
                        ops["smoother_from_filter_frac_of_hbm_peak"] = nbytes / (ops["smoother_from_filter_ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
                        del sa, sP, ra, rP
            except Exception as e:  # noqa: BLE001 -- an extra must never break the contract line
                ops["error"] = str(e)[:200]
            out["other_entry_points"] = ops
        if not args.no_cpu:
            # bounded sample (~12 s of host work); doubles as the in-bench parity check of the
            # GATHERED log-likelihoods (rank 0's own draws lead the gathered vector)
            ns_cpu = 48 if big else None
            cpu_logp, cb = cpu_baseline(w_mine if not strong else base, ft, full,
                                        n_sample=args.cpu_sample, n_steps=ns_cpu, mvn_const=fkw["mvn_const"])
            ns = min(cpu_logp.shape[0], B_local)
            out["cpu_baseline"] = cb
            if ns_cpu is None:
                gpu_lp = gathered[lo_mine:lo_mine + ns].cpu().numpy()
            else:  # the oracle ran a prefix of the series: compare with the same prefix of ll_obs
                gpu_lp = res.ll_obs[:ns, :ns_cpu].sum(-1).cpu().numpy()
            dl = np.abs(gpu_lp - cpu_logp[:ns])
            rel = dl / np.maximum(1.0, np.abs(cpu_logp[:ns]))
            # the tolerance of tests/test_gpu_parity.py: 1e-8 absolute on |logp| <= 1e4, relative beyond
            tol = 1e-8 * np.maximum(1.0, np.abs(cpu_logp[:ns]) / 1e4)
            ok = bool(np.all(dl <= tol) and rel.max() <= PARITY_REL_FAIL)
            out["parity"] = {"max_abs_dlogp_vs_oracle": float(dl.max()), "max_rel_dlogp_vs_oracle": float(rel.max()),
                             "draws_checked": int(ns), "max_abs_logp": float(np.abs(cpu_logp).max()),
                             "mvn_const": fkw["mvn_const"], "tolerance": "1e-8 * max(1, |logp| / 1e4) per draw",
                             "pass": ok}
            parity_failed = not ok
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: the in-run parity check against the oracle FAILED (see the parity object)")


if __name__ == "__main__":
    main()
