#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, spills, scratch, occupancy, LDS) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: tools/kres.py <file.hip> [name filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        cur = re.sub(r"\(.*", "", name).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if not rows:
    print(out[-3000:])
print("%-52s %5s %5s %6s %7s %4s %6s" % ("kernel", "VGPR", "AGPR", "vspill", "scratch", "occ", "sgpr"))
for k, r in rows.items():
    if flt in k:
        print("%-52s %5d %5d %6d %7d %4d %6d" % (k[:52], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("VGPRs Spill", 0),
                                                r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("TotalSGPRs", 0)))
