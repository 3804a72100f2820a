"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    print(f"{'kernel':72s} {'calls':>5s} {'avg ms':>9s} {'total ms':>9s} {'%':>6s}")
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        nm = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:72]
        print(f"{nm:72s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e6:9.3f} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):6.2f}")
