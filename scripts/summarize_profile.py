"""Summarise a scripts/profile.sh output directory: per-kernel stats + HBM traffic per launch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE reads exactly 1/2 of the
bytes of a wide coalesced streaming read -> the read side is reported raw AND doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (8-B-per-lane stores uncalibrated).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def _commit():
    """The commit the measured tree was built from: KF_COMMIT, else the stamp scripts/grun.sh writes before
    the snapshot leaves for the GPU box (there is no .git there), else git itself."""
    c = os.environ.get("KF_COMMIT")
    if c:
        return c
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        return open(os.path.join(root, ".kf_commit")).read().strip() or "unknown"
    except OSError:
        pass
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001
        return "unknown"


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def kernel_stats(d):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    return rows


def pmc(d, counter):
    per = defaultdict(list)
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    out = sys.argv[1]
    bench = None
    benches = {}
    print(f"# profile summary: {out}")
    for tag in ("kt", "fetch", "write"):
        f = os.path.join(out, f"bench_{tag}.json")
        if os.path.exists(f):
            for line in open(f):
                if line.startswith("{"):
                    j = json.loads(line)
                    print(f"bench[{tag}]: value={j['value']:.4g} {j['unit']} ms_per_step={j['ms_per_step']:.4f} "
                          f"kernel_ms={j['roofline']['kernel_ms']:.4f} achieved={j['roofline']['achieved']:.1f} GB/s "
                          f"frac={j['roofline']['frac']:.4f}  [{j['config']['workload']}]")
                    abytes = j["roofline"].get("algorithmic_bytes_per_launch")
                    bench = j
                    benches[tag] = j
    print("\n## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in sorted(kernel_stats(os.path.join(out, "kt")), key=lambda r: -float(r["TotalDurationNs"])):
        print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} "
              f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")
    print("\n## HBM traffic per launch (separate --pmc passes)")
    entries = []
    fetch = pmc(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    ours = lambda name: any(t in name for t in ("kf_", "big_", "bsp_", "bbb_", "sb_"))   # noqa: E731  (this library's kernels)
    for k in sorted(set(fetch) | set(write)):
        if not ours(k):
            continue
        fk = fetch.get(k, [])
        wk = write.get(k, [])
        fmean = sum(fk) / len(fk) * 1024 if fk else float("nan")
        wmean = sum(wk) / len(wk) * 1024 if wk else float("nan")
        print(f"{k[:70]}")
        print(f"   launches fetch/write passes: {len(fk)}/{len(wk)}")
        print(f"   FETCH_SIZE raw  = {fmean / 1e9:.4f} GB/launch  (x2 gfx950 wide-read correction = {2 * fmean / 1e9:.4f} GB)")
        print(f"   WRITE_SIZE      = {wmean / 1e9:.4f} GB/launch")
        try:
            print(f"   algorithmic     = {abytes / 1e9:.4f} GB/launch ; traffic(write + 2*fetch) = "
                  f"{(wmean + 2 * fmean) / 1e9:.4f} GB ; traffic(write + fetch raw) = {(wmean + fmean) / 1e9:.4f} GB")
        except NameError:
            pass
        if fk and wk:
            entries.append(dict(kernel=k.split("<")[0].split("::")[-1].strip(), bytes=int(wmean + 2 * fmean),
                                write_bytes=int(wmean), fetch_bytes_raw=int(fmean)))
    # machine-readable: merged into profiles/traffic.json by scripts/merge_traffic.py (bench.py reads that file and
    # labels roofline.traffic with its source)
    if entries and bench is not None:
        cfg = bench["config"]
        key = f"{bench['config']['workload'].split(':')[0]}:{'full' if cfg['outputs'] == 'full_tape' else 'logp'}:B{cfg['draws_per_gpu']}"
        top = max(entries, key=lambda e: e["bytes"])
        # a pass that is several kernels (the split pipeline: recursion chunks, compaction, fill, scan, pair kernel): traffic of the
        # WHOLE pass = the sum over every kf_ launch of the run / the number of passes the run made (bench.py --no-ops: one
        # allocation pass + warmup + steps identical passes)
        # Round 5: the clock spin-up bench.py runs before the warm-up (roofline.clock_spinup.steps, a
        # different number in every run) makes passes too; each --pmc leg is its own run, so each sum is divided by ITS run's count
        def passes_of(tag):
            j = benches.get(tag) or bench
            spin = ((j.get("roofline") or {}).get("clock_spinup") or {}).get("steps", 0) or 0
            return 1 + int(spin) + int(j.get("warmup", 0)) + int(j.get("steps", 0))

        p_w, p_f = passes_of("write"), passes_of("fetch")
        tot_w = sum(sum(v) for k, v in write.items() if ours(k)) * 1024 / p_w
        tot_f = sum(sum(v) for k, v in fetch.items() if ours(k)) * 1024 / p_f
        if len(entries) > 1:
            print(f"\nwhole pass ({len(entries)} kernels; {p_w} / {p_f} passes in the write / fetch run, spin-up included): "
                  f"WRITE_SIZE = {tot_w / 1e9:.4f} GB, FETCH_SIZE raw = {tot_f / 1e9:.4f} GB, "
                  f"traffic(write + 2*fetch) = {(tot_w + 2 * tot_f) / 1e9:.4f} GB")
            top = dict(kernel=top["kernel"], kernels=sorted(e["kernel"] for e in entries), bytes=int(tot_w + 2 * tot_f),
                       write_bytes=int(tot_w), fetch_bytes_raw=int(tot_f), passes_write_run=p_w, passes_fetch_run=p_f)
        top.update(key=key, commit=_commit(),
                   source=f"{os.path.basename(out.rstrip('/'))}: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes, "
                          "write + 2 x fetch (gfx950 wide-read correction)")
        json.dump(top, open(os.path.join(out, "traffic_entry.json"), "w"), indent=1)
        print(f"\ntraffic entry: {json.dumps(top)}")


if __name__ == "__main__":
    main()
