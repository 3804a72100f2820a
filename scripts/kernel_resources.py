"""Registers / spills / occupancy per kernel of one HIP source (clang's -Rpass-analysis=kernel-resource-usage).
   usage: python scripts/kernel_resources.py pymc_statespace_amd/csrc/kf_forward_b4.hip [name-filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for (k, v), d in zip(rows.items(), dem):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(.*\)$", "", d)
    if flt and flt not in d:
        continue
    print(f"{d[:70]:70s} {v.get('VGPRs', -1):5d} {v.get('AGPRs', -1):5d} {v.get('SGPRs', -1):5d} {v.get('VGPRs Spill', -1):6d} "
          f"{v.get('SGPRs Spill', -1):6d} {v.get('ScratchSize', -1):7d} {v.get('Occupancy', -1):4d} {v.get('LDS Size', -1):6d}")
