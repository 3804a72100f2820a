"""Instruction histogram of the smallest loop of a kernel that holds at least `min_mfma` MFMAs (the unrolled step loop), from a gfx950
assembly listing (hipcc -S --cuda-device-only).  usage: isa_loop.py file.s kernel-substring [min_mfma] [dump.s]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = re.search(r"^(_Z\S*%s\S*):.*\n" % re.escape(pat), s, re.M)
end = re.compile(r"^\.Lfunc_end\d+:", re.M).search(s, m.end())
body = s[m.end():end.start()]
labels, ins = {}, []
for ln in body.split("\n"):
    mm = re.match(r"(\.LBB\d+_\d+):", ln)
    if mm:
        labels[mm.group(1)] = len(ins)
        continue
    if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", ".")):
        ins.append(ln.strip())
print("kernel:", len(ins), "instructions,", sum("mfma" in y for y in ins), "mfma")
best = None
for i, x in enumerate(ins):
    mm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", x)
    if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
        seg = ins[labels[mm.group(1)]:i + 1]
        if sum("mfma" in y for y in seg) >= min_mfma and (best is None or len(seg) < len(best[1])):
            best = (mm.group(1), seg)
if best is None:
    sys.exit("no such loop")
l, seg = best
print("loop", l, len(seg), "instructions (static, all paths),", sum("mfma" in y for y in seg), "mfma")
c = Counter(y.split()[0] for y in seg)
print("  ".join(f"{k}={v}" for k, v in c.most_common(70)))
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write("\n".join(seg))
