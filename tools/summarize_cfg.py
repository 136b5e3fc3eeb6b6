#!/usr/bin/env python3
"""Condense the profile directories of the secondary configurations into committable summaries.

    python tools/summarize_cfg.py gpurun_out/prof_r02b_cfg3 profiles/r02b_cfg3 [nbodies=1e7]
    python tools/summarize_cfg.py --cfg4 gpurun_out/prof_cfg4 profiles/r02b_cfg4

Config 3 (tools/profile_cfg.sh): <out>_kernel_stats.csv from rocprofv3 --kernel-trace --stats, and
<out>_counters.txt: per-kernel means of FETCH_SIZE / WRITE_SIZE (separate --pmc passes; KiB; FETCH x2.0
on gfx950 for 8-byte/lane streams, the calibration of profiles/r02b_traffic.json) and of the SQ sets,
with the derived bytes per particle and VALU instructions per 64-particle wave.
Config 4 (tools/dbg/prof_cfg4.sh): <out>_kernel_stats.csv and <out>_trace.txt (tools/trace_cfg4.py: one
steady master step, launches, GPU-busy fraction, per-kernel totals, idle gaps)."""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict


def stats(src_glob, out):
    rows = []
    for path in glob.glob(src_glob):
        for r in csv.DictReader(open(path)):
            if r["Name"].startswith(("k_", "void k_")):
                rows.append(r)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms\n")
        for r in rows:
            n = r["Name"].split("(")[0].replace("void ", "")
            f.write(f"{n},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.4f},{float(r['AverageNs'])/1e6:.4f},"
                    f"{float(r['MinNs'])/1e6:.4f},{float(r['MaxNs'])/1e6:.4f}\n")
    return rows


def main():
    if sys.argv[1] == "--cfg4":
        src, out = sys.argv[2], sys.argv[3]
        latest = sorted(glob.glob(os.path.join(src, "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
        stats(latest, out)
        txt = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "trace_cfg4.py"), src],
                             capture_output=True, text=True).stdout
        open(out + "_trace.txt", "w").write(txt)
        print(txt)
        return
    src, out = sys.argv[1], sys.argv[2]
    n = float(sys.argv[3]) if len(sys.argv) > 3 else 1e7
    stats(os.path.join(src, "stats", "*", "*_kernel_stats.csv"), out)
    agg = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out + "_counters.txt", "w") as f:
        f.write(f"# per-launch means (first launch dropped), {n:.0e} particles; FETCH_SIZE/WRITE_SIZE in KiB from separate\n"
                "# --pmc passes; bytes/particle = (2.0 x FETCH_SIZE + WRITE_SIZE) x 1024 / N (gfx950 FETCH calibration)\n")
        for k, d in sorted(agg.items()):
            m = {c: (sum(v[1:]) / max(1, len(v[1:])) if len(v) > 1 else v[0]) for c, v in d.items()}
            line = k + " : " + ", ".join(f"{c}={m[c]:.5g}" for c in sorted(m))
            extra = []
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                extra.append(f"HBM {(2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0 / n:.1f} B/particle")
            if "SQ_INSTS_VALU" in m:
                extra.append(f"VALU {m['SQ_INSTS_VALU'] / (n / 64.0):.0f} per 64-particle wave")
            f.write(line + (("   => " + ", ".join(extra)) if extra else "") + "\n")
    print(open(out + "_counters.txt").read())


if __name__ == "__main__":
    main()
