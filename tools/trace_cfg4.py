#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/bench_configs.py --only 4: the last complete master step
(16 sub-steps, delimited by the long k_kick_adjust launches of the sweep over all levels): span, GPU-busy
fraction, per-kernel totals, launch count, and where the idle gaps are.

    python tools/trace_cfg4.py gpurun_out/prof_cfg4 [ncomp=2] [multistep=4]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    ncomp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ms = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]),
                     int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    # master steps end with the sweep over ALL levels: the long k_kick_adjust launches, one per component (the sweeps
    # of the sub-steps in between are short, and the ones that cannot move anything are not launched at all)
    adj_all = [i for i, r in enumerate(rows) if r[0].startswith("k_kick_adjust")]
    dmax = max(rows[i][2] - rows[i][1] for i in adj_all)
    adj = [i for i in adj_all if rows[i][2] - rows[i][1] >= 0.4 * dmax]
    per = ncomp
    # one complete steady master step: the one before the last (the last one runs into the bench's
    # final downloads); a master step ends with its last sub-step's adjust launches
    back = int(sys.argv[4]) if len(sys.argv) > 4 else 1

    def union_busy(seg):
        busy, hi = 0.0, seg[0][1]
        for _, s_, e_ in seg:
            if e_ > hi:
                busy += (e_ - max(s_, hi)) / 1e6
                hi = e_
        return busy

    # every complete master step of the trace (the first ones still carry begin_run's level changes)
    k, tot_sp, tot_bz, cnt = 1, 0.0, 0.0, 0
    while (k + 1) * per + ncomp <= len(adj):
        sg = rows[adj[-(k + 1) * per - ncomp]:adj[-k * per - ncomp]]
        sp = (sg[-1][2] - sg[0][1]) / 1e6
        print(f"master step -{k}: span {sp:6.2f} ms, GPU busy {union_busy(sg) / sp:.3f}, {len(sg)} launches")
        if k <= 6:
            tot_sp += sp; tot_bz += union_busy(sg); cnt += 1
        k += 1
    if cnt:
        print(f"last {cnt} master steps: mean span {tot_sp / cnt:.2f} ms, GPU busy {tot_bz / tot_sp:.3f}")
    a, b = adj[-(back + 1) * per - ncomp], adj[-back * per - ncomp]
    seg = rows[a:b]
    span = (seg[-1][2] - seg[0][1]) / 1e6
    ksum = sum(e - s for _, s, e in seg) / 1e6
    # GPU-busy time = the UNION of the kernel intervals (two streams overlap, so the plain sum of the
    # durations can exceed the span)
    busy, hi = 0.0, seg[0][1]
    for _, s, e in seg:
        if e > hi:
            busy += (e - max(s, hi)) / 1e6
            hi = e
    print(f"{os.path.basename(f)}: master step span {span:.2f} ms, GPU busy (union of kernel intervals) "
          f"{busy:.2f} ms = {busy / span:.3f}, kernel time summed {ksum:.2f} ms, {len(seg)} launches")
    tot, cnt = collections.Counter(), collections.Counter()
    for k, s, e in seg:
        tot[k] += (e - s) / 1e6
        cnt[k] += 1
    for k, v in tot.most_common(24):
        print(f"  {k[:44]:44s} n={cnt[k]:4d} {v:7.3f} ms")
    gaps = collections.Counter()
    hi, klast = seg[0][2], seg[0][0]
    for k1, s1, e1 in seg[1:]:
        if s1 > hi:
            gaps[klast[:30] + " -> " + k1[:30]] += (s1 - hi) / 1e6
        if e1 > hi:
            hi, klast = e1, k1
    print(f"  idle {span - busy:.2f} ms; largest gap classes:")
    for k, v in gaps.most_common(8):
        print(f"    {k:64s} {v:6.3f} ms")


if __name__ == "__main__":
    main()
