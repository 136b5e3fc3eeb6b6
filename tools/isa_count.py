#!/usr/bin/env python3
"""Static fp64 instruction mix per kernel from the device assembly of one .hip file:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -S \
          exp_amd/csrc/cyl.hip -o build/asm/cyl.s
    python tools/isa_count.py build/asm/cyl.s [name filter]

Counts v_fma/v_fmac_f64 (2 flop), v_mul/v_add/v_min/v_max_f64 (1 flop) and the other f64 VALU
instructions per function.  The per-particle kernels are unrolled straight-line code (static_for
over (l, m)), so for them the static count of the main path is the executed count per lane up to
the small prologue branches; kernels with run-time loops are only indicative."""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, rows = None, {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            rows[cur] = {"fma": 0, "muladd": 0, "other64": 0, "valu": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0}
            continue
        if cur is None:
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        r = rows[cur]
        if op.startswith("v_"):
            r["valu"] += 1
            if "f64" in op:
                if op.startswith(("v_fma_f64", "v_fmac_f64")):
                    r["fma"] += 1
                elif op.startswith(("v_mul_f64", "v_add_f64", "v_min_f64", "v_max_f64")):
                    r["muladd"] += 1
                else:
                    r["other64"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            r["smem"] += 1
        elif op.startswith("s_"):
            r["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            r["vmem"] += 1
        elif op.startswith("ds_"):
            r["lds"] += 1
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    for (k, r), n in zip(rows.items(), names):
        n = n.split("(")[0].replace("void ", "")
        if pat and pat not in n:
            continue
        if r["valu"] == 0:
            continue
        flops = 2 * r["fma"] + r["muladd"]
        print(f"{n[:48]:48s} fma={r['fma']:5d} mul/add={r['muladd']:5d} other_f64={r['other64']:4d} "
              f"flop/lane={flops:6d} valu={r['valu']:5d} salu={r['salu']:5d} smem={r['smem']:4d} "
              f"vmem={r['vmem']:4d} lds={r['lds']:4d}")


if __name__ == "__main__":
    main()
