#!/usr/bin/env python3
"""Mean duration of every (kernel, grid size) pair among the thin kernels and their table-path counterparts in the second
half of a rocprofv3 kernel trace of config 4.    python tools/dbg/thin_kernel_times.py gpurun_out/prof_dir"""
import collections, csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
    rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, g))
rows.sort()
rows = rows[3 * len(rows) // 4:]
agg = collections.defaultdict(list)
for _, dt, n, g in rows:
    if any(k in n for k in ("thin", "_wave", "_tile", "sum_combine", "k_sph_force<", "k_cyl_force<", "project", "contract", "mstep", "k_kick_adjust", "advance")):
        agg[(n, 1 << max(g - 1, 0).bit_length())].append(dt)       # grid sizes in powers of two
for (n, g), v in sorted(agg.items()):
    print(f"{n[:40]:40s} grid <= {g:6d}  n={len(v):4d}  mean {sum(v) / len(v):7.1f} us  min {min(v):7.1f}")
