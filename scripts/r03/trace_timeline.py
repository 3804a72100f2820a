"""Timeline of the LAST forward pass in a rocprofv3 kernel trace: per launch start (us from the pass's first kernel), duration, name."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "kf_" in r["Kernel_Name"]]
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# passes are separated by a chunk-0 recursion launch: find the last kernel named kf_fwd_b4_kernel preceded by a non-recursion kernel... simpler: last N rows
names = [r["Kernel_Name"] for r in rows]
# last pass = rows after the last gap > 200 us
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
cut = 0
n_rec = sum("kf_fwd_b4_kernel" in x for x in names)
per_pass = len(rows) // npass if len(rows) % npass == 0 else None
if per_pass:
    cut = len(rows) - per_pass
t0 = starts[cut]
tot = {}
for r, s, e in zip(rows[cut:], starts[cut:], ends[cut:]):
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8s}  {nm}")
    tot[nm] = tot.get(nm, 0) + (e - s)
print(f"pass wall: {(max(ends[cut:]) - t0) / 1e3:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"   {v / 1e3:9.1f} us total  {k}")
