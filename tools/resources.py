#!/usr/bin/env python3
"""Summarise -Rpass-analysis=kernel-resource-usage remarks from build/log/*.log."""
import glob, re, subprocess, sys
rows = []
for path in sorted(glob.glob("build/log/*.log")):
    cur = None
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for k, short in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("TotalSGPRs:", "sgpr"),
                         ("ScratchSize [bytes/lane]:", "scratch"),
                         ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds")):
            if k in line and cur is not None:
                cur[short] = line.split(k)[1].split()[0]
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r, n in zip(rows, names):
    n = n.split("(")[0]
    if pat and pat not in n:
        continue
    print(f"{n[:64]:64s} vgpr={r.get('vgpr'):>4} agpr={r.get('agpr'):>3} sgpr={r.get('sgpr'):>4} "
          f"scratch={r.get('scratch'):>5} occ={r.get('occ')} lds={r.get('lds')}")
