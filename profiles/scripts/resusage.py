#!/usr/bin/env python3
"""usage: resusage.py file.hip [filter]  -> per-kernel VGPR / AGPR / scratch / occupancy / LDS table (hipcc remarks)"""
import re, subprocess, sys, os
EXTRA = os.environ.get("RESUSAGE_FLAGS", "").split()
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *EXTRA, "-c", sys.argv[1],
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def flush():
    if cur.get("name") and flt in cur["name"]:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("cddp_dev::", "").split("(")[0][:110]
        print("%-110s VGPR %4s AGPR %4s scratch %6s occ %2s LDS %6s" % (name, cur.get("v"), cur.get("a"), cur.get("s"), cur.get("o"), cur.get("l")))
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: flush(); cur = {"name": m.group(1)}; continue
    for key, pat in (("v", r" VGPRs: (\d+)"), ("a", r"AGPRs: (\d+)"), ("s", r"ScratchSize \[bytes/lane\]: (\d+)"), ("o", r"Occupancy \[waves/SIMD\]: (\d+)"), ("l", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m: cur[key] = m.group(1)
flush()
