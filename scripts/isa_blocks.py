"""Per-basic-block instruction mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): LDS / fp64 / MFMA / memory
counts for the blocks above a size threshold -- what a loop body issues per trip.  usage: isa_blocks.py file.s kernel-substring [min]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r"^(_Z\S*%s\S*):.*\n" % re.escape(pat), s, re.M)
if not m:
    sys.exit("kernel not found")
body = s[m.end():s.index("s_endpgm", m.end())]
parts = re.split(r"\n(\.LBB\d+_\d+):", body)
names = ["entry"] + parts[1::2]
bodies = [parts[0]] + parts[2::2]
print(m.group(1)[:100])
for n, b in zip(names, bodies):
    ins = [ln.split()[0] for ln in b.split("\n") if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", "."))]
    if len(ins) < lo:
        continue
    c = Counter()
    for x in ins:
        if x.startswith("ds_"):
            c[x] += 1
        elif x.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
            c["fp64"] += 1
        elif x.startswith("v_mfma"):
            c["mfma"] += 1
        elif x.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c[x] += 1
        elif x.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif x.startswith("v_"):
            c["valu_other"] += 1
        else:
            c["salu"] += 1
    print(f"{n:12s} {len(ins):5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
