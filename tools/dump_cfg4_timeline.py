#!/usr/bin/env python3
"""Kernel-by-kernel timeline of one settled master step of config 4 from a rocprofv3 kernel trace
(tools/dbg/prof_cfg4.sh): start offset, duration, queue, grid, kernel -- one line per launch, sub-steps
a marker line at the halo's combined coefficient set of every sub-step.

    python tools/dump_cfg4_timeline.py gpurun_out/prof_cfg4 [back=2] > timeline.txt"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), grid // max(wg, 1), name))
    rows.sort()
    adj_all = [i for i, r in enumerate(rows) if r[4].startswith("k_kick_adjust")]
    dmax = max(rows[i][1] - rows[i][0] for i in adj_all)
    adj = [i for i in adj_all if rows[i][1] - rows[i][0] >= 0.4 * dmax]      # the sweeps over all levels: 2 a master step
    per = 2
    a, b = adj[-(back + 1) * per - 2], adj[-back * per - 2]
    seg = rows[a + 1:b + 1]
    t0 = seg[0][0]
    queues = sorted({r[2] for r in seg})
    sub, nadj, tsub = 0, 0, t0
    print(f"# {os.path.basename(f)}: master step -{back}, {len(seg)} launches, span {(seg[-1][1] - t0) / 1e3:.1f} us; queues {queues}")
    print(f"## sub-step 0")
    for s, e, q, g, n in seg:
        print(f"{(s - t0) / 1e3:9.1f} +{(s - tsub) / 1e3:7.1f} {(e - s) / 1e3:8.1f} us  q{queues.index(q)} grid {g:7d}  {n[:60]}")
        if n.startswith("k_sph_sum_combine") or n.startswith("k_sph_sum_parts"):      # (one a sub-step, halo stream)
            if n.startswith("k_sph_sum_combine"):
                nadj += 1
                print(f"##   ^ sub-step {nadj - 1}'s coefficients (halo)")


if __name__ == "__main__":
    main()
