#!/usr/bin/env python3
"""bench.py -- Kalman log-likelihood throughput on MI355X (contract: see the round brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c2|c3|c4|c4_m10|weekly|hourly]

A "step" is ONE pass of the hot path over one batch of synthetic draws already resident in
HBM: the fused forward filter (reference pymc_statespace/filters/kalman_filter.py:126-252)
for every draw, writing the same six outputs the reference graph registers
(filtered/predicted states + covariances, logp, ll_obs -- core/statespace.py:164-174), i.e.
the "reference-equivalent op" of SURVEY.md section 8(d), which is HBM-write bound.
With N > 1 GPUs (one process per GPU, torch.distributed over RCCL) the draw axis is sharded
-- each rank filters its own B draws (weak scaling) -- and the per-draw log-likelihoods are
all-gathered (the only exchange step of the path) inside the timed region.

Prints ONE JSON line on rank 0.  `value` = draws filtered per second over all ranks
("evals/s": one evaluation = one draw's full T-step filter).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured copy)

DEFAULT_FILTER = {"headline": "standard", "c2": "standard", "c3": "cholesky", "c4": "standard",
                  "c4_m10": "standard", "c1": "standard", "c5": "standard"}


def algorithmic_bytes(B, n, p, m, r, full_tape=True):
    """bytes one launch must move (fp64)."""
    inp = 8 * (p * m + 2 * m * m + m * r + r * r + p * p + m)
    out = 8 * (n * m + (n + 1) * m + n * m * m + (n + 1) * m * m + n + 1) if full_tape else 8
    return B * (inp + out) + 8 * n * p


def cpu_baseline(w, filter_type, full_tape=True, target_s=15.0, n_sample=None):
    """The C restatement of the reference algorithm (oracle/kf_oracle.c, kind "port") on the
    host cores, on the leading draws of the SAME workload, computing the same outputs.

    The sample is processed in chunks whose output tapes reuse one pre-touched host buffer
    (first-touch page faults of multi-GB fresh arrays would otherwise dominate the CPU time;
    recycling the buffer only favours the CPU)."""
    from oracle import c_oracle as co

    cores = os.cpu_count() or 1
    chunk = max(8, 4 * cores)
    B = w["B"]

    def run(lo, hi, out=None):
        sl = slice(lo, hi)
        return co.kalman_filter(w["y"], w["a0"][sl], w["P0"][sl], w["T"][sl], w["Z"][sl],
                                w["R"][sl], w["H"][sl], w["Q"][sl], filter_type=filter_type,
                                full=full_tape, nthreads=cores, out=out)

    chunk = min(chunk, B)
    buf = run(0, chunk)          # untimed: loads the library, touches the tape buffer
    t0 = time.perf_counter()
    run(0, chunk, out=buf)       # calibration pass
    rate = chunk / (time.perf_counter() - t0)
    if n_sample is None:
        n_sample = int(min(B, max(chunk, rate * target_s)))
    n_sample = max(chunk, (n_sample // chunk) * chunk)
    logp = np.empty(n_sample)
    t0 = time.perf_counter()
    for lo in range(0, n_sample, chunk):
        run(lo, lo + chunk, out=buf)
        logp[lo:lo + chunk] = buf["logp"]
    dt = time.perf_counter() - t0
    return logp, dict(value=n_sample / dt, unit="evals/s", cores=cores, kind="port",
                      sample=f"first {n_sample} draws x all {w['n']} steps of the same workload, "
                             f"{'full tapes' if full_tape else 'logp only'}, {dt:.2f} s wall, "
                             f"oracle/kf_oracle.c, OpenMP over draws on {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline")
    ap.add_argument("--filter", default=None)
    ap.add_argument("--batch", type=int, default=None, help="draws per GPU (default: workload's B)")
    ap.add_argument("--logp-only", action="store_true", help="skip the tapes (what NUTS consumes)")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lanes", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import workloads as W
    from pymc_statespace_amd import KalmanEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    ft = args.filter or DEFAULT_FILTER.get(args.workload, "standard")
    kw = {} if args.batch is None else {"B": args.batch}
    w = W.WORKLOADS[args.workload](**kw)  # same seeded draws on every rank (weak scaling)
    B, n, p, m, r = w["B"], w["n"], w["p"], w["m"], w["r"]
    full = not args.logp_only

    eng = KalmanEngine(dev, lanes_per_draw=args.lanes)
    keys = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
    d = [eng.to_device(w[k]) for k in keys]
    fkw = dict(filter_type=ft, mvn_const="exact", store_ll_obs=full, store_filtered=full,
               store_predicted=full)
    res = eng.forward(*d, **fkw)  # allocates the output tapes once; reused by every step
    gathered = torch.empty(world * B, device=dev, dtype=torch.float64) if world > 1 else None
    torch.cuda.synchronize()

    def step():
        eng.forward(*d, out=res, **fkw)
        if world > 1:
            dist.all_gather_into_tensor(gathered, res.logp)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        eng.forward(*d, out=res, **fkw)
        ev[k][1].record()
        if world > 1:
            dist.all_gather_into_tensor(gathered, res.logp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        abytes = algorithmic_bytes(B, n, p, m, r, full)
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"{args.workload}:{'full' if full else 'logp'}")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "Kalman logp evals/sec (batch x T x m)",
            "value": world * B * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {w['name']} B={B} draws/GPU x T={n} x m={m} "
                            f"(p={p}, r={r}), {ft} filter, "
                            f"{'all six reference outputs written to HBM' if full else 'logp only'}",
                "draws_per_gpu": B, "T": n, "k_states": m, "k_endog": p, "k_posdef": r,
                "filter": ft, "outputs": "full_tape" if full else "logp_only",
                "global_draws": world * B, "parallelism": f"draws sharded x{world}",
                "draw_steps_per_s": world * B * n * args.steps / elapsed,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": eng.forward_kernel_name(B, n, p, m, r), "kernel_ms": kern_ms,
                "algorithmic_bytes_per_launch": abytes,
            },
        }
        if not args.no_cpu:
            # bounded sample (~15 s of host work); doubles as the in-bench parity check
            cpu_logp, cb = cpu_baseline(w, ft, full, n_sample=args.cpu_sample)
            ns = cpu_logp.shape[0]
            out["cpu_baseline"] = cb
            dl = np.abs(res.logp[:ns].cpu().numpy() - cpu_logp)
            out["parity"] = {"max_abs_dlogp_vs_oracle": float(dl.max()), "draws_checked": int(ns),
                             "max_abs_logp": float(np.abs(cpu_logp).max())}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
