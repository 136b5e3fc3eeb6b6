#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/bench_configs.py --only 4: the last complete master step
(16 sub-steps, delimited by the k_kick_adjust launches of the first component): span, GPU-busy
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
    adj = [i for i, r in enumerate(rows) if r[0].startswith("k_kick_adjust")]
    per = ncomp * (1 << ms)
    # the last complete master step ends with the last adjust launch of the trace
    a, b = adj[-per - ncomp], adj[-ncomp]
    seg = rows[a:b]
    span = (seg[-1][2] - seg[0][1]) / 1e6
    busy = sum(e - s for _, s, e in seg) / 1e6
    print(f"{os.path.basename(f)}: master step span {span:.2f} ms, busy {busy:.2f} ms ({busy / span:.3f}), "
          f"{len(seg)} launches")
    tot, cnt = collections.Counter(), collections.Counter()
    for k, s, e in seg:
        tot[k] += (e - s) / 1e6
        cnt[k] += 1
    for k, v in tot.most_common(24):
        print(f"  {k[:44]:44s} n={cnt[k]:4d} {v:7.3f} ms")
    gaps = collections.Counter()
    for (k0, s0, e0), (k1, s1, e1) in zip(seg[:-1], seg[1:]):
        g = (s1 - e0) / 1e6
        if g > 0:
            gaps[k0[:30] + " -> " + k1[:30]] += g
    print(f"  idle {span - busy:.2f} ms; largest gap classes:")
    for k, v in gaps.most_common(8):
        print(f"    {k:64s} {v:6.3f} ms")


if __name__ == "__main__":
    main()
