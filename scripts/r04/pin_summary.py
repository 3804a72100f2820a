"""Summary of scripts/r04/headline_pin.sh: the un-profiled bench lines, the rocprofv3 kernel average over ALL launches and over
the TIMED launches (the last `steps` of the run: the launches before them are the clock spin-up and warm-up), the counters per
launch, the clocks read around every leg -- and the instruction budget per wave-step restated in cycles at the MEASURED clock."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "kf_fwd_b4_kernel"


def bench(tag):
    f = os.path.join(out, f"bench_{tag}.json")
    if not os.path.exists(f):
        return None
    for line in open(f):
        if line.startswith("{"):
            return json.loads(line)
    return None


print(f"# headline pin: {out}  (commit {open('.kf_commit').read().strip() if os.path.exists('.kf_commit') else '?'})\n")
print("## bench lines (one lease)")
print(f"{'leg':18s} {'ms/step':>8s} {'kernel_ms':>9s} {'frac':>6s} {'sclk MHz (min..max)':>22s} {'W':>6s} {'Mcycles':>8s}  spin-up")
for tag in ("driver_cmd", "cold", "sustained", "exact", "kt", "fetch", "write", "sq1", "sq2", "sq3"):
    j = bench(tag)
    if not j:
        continue
    r = j["roofline"]
    sc = r.get("sclk_mhz")
    su = r.get("clock_spinup", {})
    sc_txt = "" if not sc else "%6.0f (%.0f..%.0f)" % (sc, r.get("sclk_mhz_min", 0), r.get("sclk_mhz_max", 0))
    su_txt = "off" if not su.get("enabled") else "%s steps / %.0f ms" % (su.get("steps"), su.get("ms", 0))
    print(f"{tag:18s} {j['ms_per_step']:8.4f} {r['kernel_ms']:9.4f} {r['frac']:6.3f} {sc_txt:>22s} "
          f"{r.get('socket_power_w', float('nan')):6.0f} {r.get('kernel_mcycles_at_sclk', float('nan')):8.3f}  {su_txt}")
    if tag == "driver_cmd":
        ops = j.get("other_entry_points", {})
        print("   other entry points:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ops.items()})
        print("   parity:", j.get("parity"))

print("\n## rocprofv3 --kernel-trace: the dominant kernel")
rows = []
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
jk = bench("kt")
if rows and jk:
    K = jk["steps"]
    names = defaultdict(list)
    for s, e, nm in rows:
        names[nm].append((e - s) / 1e3)
    for nm, v in names.items():
        print(f"{nm[:90]}: {len(v)} launches, average over all {sum(v) / len(v):.2f} us")
    dom = max(names.items(), key=lambda kv: sum(kv[1]))[1]
    timed = dom[-K:]
    print(f"timed launches (the last {K}): average {sum(timed) / len(timed):.2f} us, min {min(timed):.2f}, max {max(timed):.2f}")
    print(f"first 5 launches: {[round(x, 1) for x in dom[:5]]} us (clock still ramping)")
    print(f"bench line of the same run: kernel_ms {jk['roofline']['kernel_ms'] * 1e3:.2f} us (HIP events)  -> "
          f"agreement {100 * (sum(timed) / len(timed) / (jk['roofline']['kernel_ms'] * 1e3) - 1):+.2f} %")
    jd = bench("driver_cmd")
    if jd:
        print(f"un-profiled line (driver's command): kernel_ms {jd['roofline']['kernel_ms'] * 1e3:.2f} us -> profile's timed average is "
              f"{100 * (sum(timed) / len(timed) / (jd['roofline']['kernel_ms'] * 1e3) - 1):+.2f} % away")
        ab = jd["roofline"]["algorithmic_bytes_per_launch"]
        print(f"roofline from the profile's timed average: {ab / (sum(timed) / len(timed) * 1e-6) / 1e9:.0f} GB/s = "
              f"{ab / (sum(timed) / len(timed) * 1e-6) / 8e12:.3f} of 8 TB/s")

print("\n## counters per launch (timed launches only: the last `steps`)")
ctr = defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3"):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                per[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    for c, v in per.items():
        v.sort()
        vals = [x for _, x in v][-20:]
        ctr[c] = vals
        print(f"{c:24s} {sum(vals) / len(vals):.6g}   (n = {len(vals)})")
jd = bench("driver_cmd") or bench("sustained")
if jd and ctr.get("SQ_WAVE_CYCLES") and ctr.get("SQ_INSTS_VALU"):
    r = jd["roofline"]
    B, T = jd["config"]["draws_per_gpu"], jd["config"]["T"]
    ws = B * T
    per = {c: sum(v) / len(v) / ws for c, v in ctr.items()}
    print(f"\n## per wave-step ({B} wavefronts x {T} steps)")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
              "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if c in per:
            print(f"   {c:20s} {per[c]:10.2f}")
    sc = r.get("sclk_mhz")
    if sc:
        cyc = r["kernel_ms"] * 1e-3 * sc * 1e6
        simds = 1024
        print(f"   kernel: {r['kernel_ms']:.4f} ms at {sc:.0f} MHz = {cyc / 1e6:.3f} Mcycles; {B} wavefronts on {simds} SIMDs = "
              f"{B / simds:.0f} per SIMD -> {cyc / T:.0f} cycles per time step of a SIMD's {B / simds:.0f} wavefronts = "
              f"{cyc / T / (B / simds):.0f} cycles of SIMD issue time per wave-step")
        n_inst = sum(per.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR"))
        print(f"   instructions per wave-step: {n_inst:.0f} -> {cyc / T / (B / simds) / n_inst:.2f} SIMD cycles per instruction")

print("\n## clocks around every leg (amdgpu hwmon, read by the wrapper script)")
f = os.path.join(out, "clocks.txt")
if os.path.exists(f):
    print(open(f).read())
