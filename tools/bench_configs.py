#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (configs 2-4) on one GPU; one JSON line each.

    python tools/bench_configs.py [--n 1e7] [--steps 30] [--only 2|3|4]

config 2: 1e7 NFW halo, SphericalSL lmax 6 nmax 18            (fused KDK step)
config 3: 1e7 exponential disk, EmpCylSL mmax 6 nmax 12        (fused KDK step)
config 4: disk + halo (1e7 each), both self and both cross forces, multistep 4 (C++ step driver);
          reported as master-step-equivalents (N particles x 1 per dtime) and raw sub-steps.
The same runners produce bench.py's `other_configs` extra key (bench.other_configs).  These are
parity-test configurations, not the headline bench line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--only", type=int, default=0)
    args = ap.parse_args()
    import torch
    from bench import other_configs
    from exp_amd.runtime import Context
    device = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device)
    torch.cuda.set_stream(ts)
    ctx = Context(0, stream=ts.cuda_stream)
    if os.environ.get("BENCH_THIN_MAX"):                     # (A/B of exp_amd_ctx_set_thin_max; default 8192)
        ctx.set_thin_max(int(os.environ["BENCH_THIN_MAX"]))
    which = (args.only,) if args.only else (2, 3, 4)
    for o in other_configs(ctx, device, int(args.n), which, args.steps):
        print(json.dumps(o), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
