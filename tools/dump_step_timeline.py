#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the last fused steps of a bench.py run from a rocprofv3 kernel trace: start offset,
duration, queue, grid, kernel -- one line per launch -- and, per step, the wall span, the SUM of the kernel durations and
the time during which kernels of two queues ran at once (the overlap).

    python tools/dump_step_timeline.py <dir> [nsteps=2] > timeline.txt"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), grid // max(wg, 1), name))
    rows.sort()
    # a step ends with its (last) dense force launch; the projection in front of it is launched once a step
    marks = [i for i, r in enumerate(rows) if r[4].startswith("k_sph_project")]
    marks = marks[-(nsteps + 1):]
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        t0 = seg[0][0]
        queues = sorted({r[2] for r in seg})
        span = max(r[1] for r in seg) - t0
        ksum = sum(r[1] - r[0] for r in seg)
        # time covered by >= 2 kernels at once (sweep over the interval ends)
        ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
        depth, last, both, busy = 0, t0, 0, 0
        for t, dlt in ev:
            if depth >= 2:
                both += t - last
            if depth >= 1:
                busy += t - last
            depth += dlt
            last = t
        print(f"# step of {len(seg)} launches: span {span / 1e6:.3f} ms, kernel sum {ksum / 1e6:.3f} ms, "
              f"GPU busy {busy / 1e6:.3f} ms, two kernels at once {both / 1e6:.3f} ms; queues {queues}")
        for s, e, q, g, n in seg:
            if e - s < 20000 and not n.startswith("k_s"):
                continue
            print(f"{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e6:8.3f} ms  q{queues.index(q)} grid {g:7d}  {n[:70]}")


if __name__ == "__main__":
    main()
