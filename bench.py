#!/usr/bin/env python3
"""Headline benchmark: particle-steps/sec (coef + force + kick-drift) of the spherical BFE
hot path on MI355X.

Workload (BASELINE.json config 5, the configuration the metric is quoted on): 1e8-particle
truncated-NFW halo, SphericalSL lmax=10 nmax=24 (numr 2000), fp64, multistep 0.  A "step" is
one KDK leapfrog step of every particle: kick dt/2, drift dt, cell sort, coefficient
accumulation (+ one all-reduce of the 2904-double coefficient buffer when N > 1), force and
potential evaluation, kick dt/2.  The total particle count is FIXED as GPUs are added (strong
scaling): each rank owns N/world particles and the only exchange is the coefficient all-reduce.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
from __future__ import annotations

import argparse
import json
import math
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
# (several ranks on one node: the host driver of this pool supports dmabuf IPC only -- without this RCCL's and torch's
# buffer sharing between processes fails with `hipIpcGetMemHandle: invalid argument`; set before anything touches HIP)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0    # ... and what a streaming kernel reaches on this part (same guide): the second denominator
FP64_VECTOR_PEAK_TF = 78.6   # MI355X fp64 vector (non-matrix) peak, datasheet; ubench: ~73 (tools/dbg/ubench_dp.hip)
# Algorithmic HBM bytes per particle per launch of each pass (SURVEY.md section 8d accounting,
# restated in DESIGN.md): whole step 232 B = pass A (kick/2 + drift + accumulate) 128 B +
# pass B (force + kick/2) 104 B.  `roofline.achieved` uses these contract figures.
ALGO_BYTES = {
    "k_sph_force": 104.0,        # reads x,y,z,vx,vy,vz (48) ; writes ax,ay,az,pot,vx,vy,vz (56)
    "k_cyl_force": 104.0,
    "k_sph_accumulate": 32.0,    # reads x,y,z,m
    "k_cyl_accumulate": 32.0,
    "k_kick": 72.0,
    "k_drift": 72.0,
    "k_scatter_adv": 148.0,      # contract accounting: reads x,v,a,m,id (84 + 4) ; writes x,v,m,id (60); the pass itself
                                 # moves 112 (no a: prekick; no m: equal-mass component)
    "step": 232.0,
}
# What the kernels of the FUSED step themselves move: the force pass reads x,y,z,vx,vy,vz
# (48) and writes ax,ay,az,pot (32), the velocities with both half-kicks of the step boundary applied (24:
# exp_amd_ctx_set_prekick, so that the reordering pass does not read the accelerations) and the 4-byte sort key of
# the next step = 108 B, i.e. the contract's 104 + the key.
OWN_BYTES = {"k_sph_force": 108.0, "k_cyl_force": 108.0}
# ... and in the APPEND form of the fused step (exp_amd_ctx_set_append_min), where the force pass also
# drifts the particle and places it in the next step's cell order: reads x,y,z,vx,vy,vz,id (52); writes the next position
# (24), the velocities with both half-kicks (24), ax,ay,az,pot (32), id and the slot it came from (8) = 140 B.  The
# contract figure quoted as `achieved` stays pass B's 104 B.
OWN_BYTES_APPEND = {"k_sph_force": 140.0}
# (--append-lean, exp_amd_ctx_set_append_lean: acceleration and potential are not placed but re-evaluated for whoever looks: 108 B)
OWN_BYTES_APPEND_LEAN = {"k_sph_force": 108.0}
# fp64 operations executed per particle (FMA = 2), static count of the unrolled fast paths
# (tools/isa_count.py on the gfx950 assembly)
# (k_cyl_force: its static count of 1427 holds both table paths -- scalar rows for a cell-uniform wave,
# per-lane gathers otherwise, ~400 flops each -- and the erf taper; a wave executes one of them: ~820)
EXEC_FLOPS = {("k_sph_force", 10): 2579.0, ("k_sph_force", 6): 1368.0, ("k_cyl_force", 6): 820.0,
              ("k_sph_accumulate", 10): 1100.0, ("k_sph_accumulate", 6): 640.0, ("k_cyl_accumulate", 6): 350.0}
# reference formulation, SURVEY.md section 8d: flops per particle-step
REF_FLOPS = {"S6": 7800.0, "S10": 22600.0, "C6": 6200.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nbodies", type=float, default=1e8,
                    help="particle count: the TOTAL over all ranks with --scaling strong (default), PER GPU with "
                         "--scaling weak")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --nbodies particles in total, N/world per GPU (BASELINE config 5: 1e8 sharded); "
                         "weak: --nbodies particles on EVERY GPU (SURVEY 8d config 5 asks for both curves; the "
                         "8-GPU share of the strong run is `--scaling weak --nbodies 1.25e7`)")
    ap.add_argument("--lmax", type=int, default=10)
    ap.add_argument("--nmax", type=int, default=24)
    ap.add_argument("--numr", type=int, default=2000)
    ap.add_argument("--dt", type=float, default=0.002)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes of this same "
                         "command, FETCH_SIZE and WRITE_SIZE, about a minute together); it is also skipped with "
                         "--no-cpu-baseline (the quick runs of the A/B and profiling tools), with several ranks, and "
                         "when this process itself runs under a profiler.  The line then carries the figure of "
                         "profiles/traffic.json and says so")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the secondary BASELINE configurations (2: 1e7 S6 halo, 3: 1e7 C6 disk, "
                         "4: disk + halo, multistep 4) measured after the headline on rank 0 at N = 1")
    ap.add_argument("--graph", action="store_true",
                    help="time the K steps as ONE call of exp_amd_step_kdk_n (pairs of steady-state steps replayed from a "
                         "HIP graph, the RCCL all-reduce included) instead of K calls of exp_amd_step_kdk; per-kernel "
                         "events are off then (roofline.avg_launch_ms comes from a second, eager region)")
    ap.add_argument("--no-append", action="store_true",
                    help="the ordinary fused step (key histogram, scan, scatter pass every step) instead of the APPEND form, in "
                         "which the force pass places every particle in the next step's cell order itself "
                         "(exp_amd_ctx_set_append_min; profiles/r06_append_ab.txt)")
    ap.add_argument("--no-lean-ab", action="store_true",
                    help="skip the extra K-step region with the append step's lean payload reported as `append_lean_ab`")
    ap.add_argument("--append-lean", action="store_true",
                    help="A/B: the append step with the LEAN payload (exp_amd_ctx_set_append_lean: the placing pass stores neither "
                         "acceleration nor potential; they are re-evaluated for the first call that looks at the particles -- here the "
                         "self-check after the timed region).  NOT the default line: the default leaves the complete state in the "
                         "store after every step")
    ap.add_argument("--split", action="store_true",
                    help="the opt-in split fused step (exp_amd_ctx_set_split_min): the store as two independently sorted "
                         "halves, the HBM-bound sort passes of one half on a second stream under the fp64-bound accumulate / "
                         "force passes of the other (A/B: profiles/r06_overlap_ab.txt)")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the extra >= 2 s timed region reported as `sustained`")
    ap.add_argument("--other-n", type=float, default=1e7, help="particles per component of those")
    ap.add_argument("--cpu-sample", type=int, default=40000)
    ap.add_argument("--force-comm", action="store_true",
                    help="initialise the process group and run the coefficient all-reduce even at "
                         "world size 1 (exercises the multi-GPU path on a single-GPU box)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="dress rehearsal of the N > 1 launch on a ONE-GPU box: every rank uses device 0, the process "
                         "group is gloo, and the coefficient all-reduce is host-staged through it (RCCL refuses two ranks "
                         "on one device).  Sharding, the communicator vote and its fallback, the MAX-reduce of the timings "
                         "and the rank-0-only JSON line are the production code; the rate it prints is NOT a scaling "
                         "measurement (the ranks time-share one GPU) and the line says so")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl",
                    help="coefficient all-reduce through the library's own RCCL communicator on the "
                         "compute stream (default; falls back to torch.distributed if its self-check "
                         "fails) or through a torch.distributed callback")
    return ap.parse_args()


def make_halo(model, n, seed, device):
    """n equal-mass NFW particles + Jeans-dispersion velocities, generated in HBM."""
    import torch
    from exp_amd.models import sphere_sampling_tables
    u_tab, r_tab, s_tab = sphere_sampling_tables(model, 0.98 * model.rmax)
    gen = torch.Generator(device=device).manual_seed(seed)
    f64 = torch.float64
    ut = torch.tensor(u_tab, device=device, dtype=f64)
    rt = torch.tensor(r_tab, device=device, dtype=f64)
    sg = torch.tensor(s_tab, device=device, dtype=f64)
    u = torch.rand(n, device=device, dtype=f64, generator=gen)
    idx = torch.searchsorted(ut, u).clamp_(1, len(u_tab) - 1)
    w = (u - ut[idx - 1]) / (ut[idx] - ut[idx - 1])
    r = rt[idx - 1] + w * (rt[idx] - rt[idx - 1])
    sig = sg[idx - 1] + w * (sg[idx] - sg[idx - 1])
    del u, w, idx
    ct = torch.rand(n, device=device, dtype=f64, generator=gen) * 2 - 1
    ph = torch.rand(n, device=device, dtype=f64, generator=gen) * (2 * math.pi)
    st = torch.sqrt(1 - ct * ct)
    x = (r * st * torch.cos(ph)).contiguous()
    y = (r * st * torch.sin(ph)).contiguous()
    z = (r * ct).contiguous()
    del ct, ph, st, r
    vx = torch.randn(n, device=device, dtype=f64, generator=gen) * sig
    vy = torch.randn(n, device=device, dtype=f64, generator=gen) * sig
    vz = torch.randn(n, device=device, dtype=f64, generator=gen) * sig
    return x, y, z, vx, vy, vz


def _cpu_steps(orc, grid, prm, dt, m, pos, vel, nthreads, budget_s, max_steps):
    """KDK steps of the oracle with the reference's thread structure (src/PotAccel.cc:97-130:
    nthrds contiguous particle slices; per-thread coefficient arrays summed afterwards,
    src/SphericalBasis.cc:855-903).  ctypes releases the GIL inside the C calls."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(m)
    cuts = [n * k // nthreads for k in range(nthreads + 1)]
    sl = [slice(cuts[k], cuts[k + 1]) for k in range(nthreads)]
    p, v, a = pos.copy(), vel.copy(), np.zeros_like(pos)
    t0 = time.perf_counter()
    nsteps = 0
    with ThreadPoolExecutor(nthreads) as ex:
        while True:
            v += a * (0.5 * dt)
            p += v * dt
            parts = list(ex.map(lambda s_: orc.sph_accumulate(grid, prm, p[s_], m[s_])[0], sl))
            coef = np.sum(parts, axis=0)
            res = list(ex.map(lambda s_: orc.sph_accel(grid, prm, p[s_], coef)[0], sl))
            a = np.concatenate(res)
            v += a * (0.5 * dt)
            nsteps += 1
            el = time.perf_counter() - t0
            if el > budget_s or nsteps >= max_steps:
                break
    return n * nsteps / el, nsteps


def _cpu_steps_tuned(orc, grid, prm, dt, m, pos, vel, nthreads, budget_s, max_steps):
    """The same KDK steps with oracle/tuned_cpu.c: the n-dependence hoisted out of the particle loops
    (per-thread cell moments + one contraction; projected table for the force) -- the algorithm of
    the device path on CPU threads, an upper bound for what CPUs can do here."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(m)
    cuts = [n * k // nthreads for k in range(nthreads + 1)]
    sl = [slice(cuts[k], cuts[k + 1]) for k in range(nthreads)]
    t = orc.tuned(grid)
    nrows = (grid.lmax + 1) ** 2
    Ws = [np.zeros((grid.numr - 1, nrows, 2)) for _ in range(nthreads)]
    p, v, a = pos.copy(), vel.copy(), np.zeros_like(pos)
    t0 = time.perf_counter()
    nsteps = 0

    def mom(k):
        Ws[k][:] = 0.0
        orc.tuned_moments(grid, t, prm, p[sl[k]], m[sl[k]], Ws[k])

    with ThreadPoolExecutor(nthreads) as ex:
        while True:
            v += a * (0.5 * dt)
            p += v * dt
            list(ex.map(mom, range(nthreads)))
            coef = orc.tuned_contract(grid, t, np.sum(Ws, axis=0))
            G = orc.tuned_project(grid, t, coef)
            res = list(ex.map(lambda s_: orc.tuned_accel(grid, t, prm, p[s_], G)[0], sl))
            a = np.concatenate(res)
            v += a * (0.5 * dt)
            nsteps += 1
            el = time.perf_counter() - t0
            if el > budget_s or nsteps >= max_steps:
                break
    return n * nsteps / el, nsteps


def host_cpus():
    """What this process may really use: min(affinity mask, cgroup CPU quota) -- a container that shows hundreds
    of CPUs often owns a fraction of them in quota, and a threaded rate scales with the quota -- and the CPU model."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:      # pragma: no cover
        ncpu = os.cpu_count() or 1
    host = {"affinity_cpus": ncpu, "os_cpu_count": os.cpu_count()}
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().strip()
        host["cgroup_cpu.max"] = txt
        a, b = txt.split()
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            host["cgroup_cfs_quota_us"] = q
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    host["cgroup_quota_cpus"] = quota
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                host["cpu_model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = ncpu if quota is None else max(1, min(ncpu, int(math.floor(quota + 1e-9))))
    host["usable_cpus"] = max(1, min(usable, 64))
    return host


def cpu_baseline(grid, model, nsample, dt):
    """The oracle (CPU restatement of EXP's CPU path, scalar fp64) timed on this host on a bounded
    sample of the same workload: once on 1 thread, once on all the cores this process may use (its
    affinity mask capped by the cgroup CPU quota), sliced the way the reference slices particles over its
    pthreads; then the device algorithm on those threads (`tuned`) and the reference's DATA STRUCTURE around
    the same arithmetic (`reference_structure`).  Baseline only -- DESIGN.md."""
    from exp_amd.models import sample_sphere
    from tests.oracle_lib import Oracle
    orc = Oracle()
    host = host_cpus()
    nthreads = host["usable_cpus"]
    prm = orc.params(rmin=grid.rmin, rmax=grid.rmax)
    m, pos, vel = sample_sphere(model, nsample, seed=777)
    v1, s1 = _cpu_steps(orc, grid, prm, dt, m, pos, vel, 1, 6.0, 50)
    big = nsample * min(nthreads, 16)
    m, pos, vel = sample_sphere(model, big, seed=778)
    vn, sn = _cpu_steps(orc, grid, prm, dt, m, pos, vel, nthreads, 8.0, 50)
    # tuned CPU mode: the device path's hoisting on the same threads
    mt, post, velt = sample_sphere(model, big * 4, seed=779)
    vt, st = _cpu_steps_tuned(orc, grid, prm, dt, mt, post, velt, nthreads, 6.0, 50)
    # reference-structure mode (oracle/refstruct_cpu.c): hash map of individually allocated
    # particles, level list of keys, five separate pthread-forked passes per step -- at N = 1e5 and 1e6
    refstruct = {}
    for nref in (100_000, 1_000_000):
        mr, pr, vr = sample_sphere(model, nref, seed=780)
        rs = orc.refstruct(mr, pr, vr)
        G = orc.grid(grid)
        orc.refstruct_field(rs, grid, prm, nthreads)
        t0, ns = time.perf_counter(), 0
        while True:
            orc.refstruct_step(rs, grid, prm, dt, nthreads, G)
            ns += 1
            el = time.perf_counter() - t0
            if el > 4.0 or ns >= 50:
                break
        orc.refstruct_free(rs)
        refstruct[f"N={nref:.0e}"] = {"value": nref * ns / el, "steps": ns, "seconds": el}
    # BASELINE config 1 at ITS OWN parameters: 1e5-particle Plummer halo, SphericalSL lmax 6 nmax 18, np = 1 -- the
    # reference's CPU-runnable plumbing case, in the reference's data structure and in the plain oracle loop, one thread
    config1 = None
    try:
        from exp_amd.models import PlummerModel
        from exp_amd.slgrid import build_slgrid
        pm = PlummerModel(1.0, 1.0, 1e-3, 50.0)
        g1 = build_slgrid(pm, 6, 18, numr=grid.numr, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0, nel=32, P=8)
        p1 = orc.params(rmin=g1.rmin, rmax=g1.rmax)
        m1, x1, w1 = sample_sphere(pm, 100_000, seed=781)
        rs = orc.refstruct(m1, x1, w1)
        G1 = orc.grid(g1)
        orc.refstruct_field(rs, g1, p1, 1)
        t0, ns = time.perf_counter(), 0
        while True:
            orc.refstruct_step(rs, g1, p1, dt, 1, G1)
            ns += 1
            el = time.perf_counter() - t0
            if el > 4.0 or ns >= 20:
                break
        orc.refstruct_free(rs)
        vo, so = _cpu_steps(orc, g1, p1, dt, m1, x1, w1, 1, 4.0, 20)
        config1 = {"workload": "BASELINE config 1: 1e5-particle Plummer halo, SphericalSL lmax 6 nmax 18, np = 1",
                   "reference_structure": {"value": 100_000 * ns / el, "steps": ns, "seconds": el, "cores": 1},
                   "port": {"value": vo, "steps": so, "cores": 1}, "unit": "particle-steps/s"}
    except Exception as e:      # pragma: no cover
        config1 = {"error": repr(e)}
    return {"value": max(vn, v1), "unit": "particle-steps/s", "cores": nthreads if vn >= v1 else 1,
            "config1": config1,
            "kind": "port", "value_1thread": v1, "thread_speedup": vn / v1 if v1 > 0 else None, "host": host,
            "reference_structure": {**refstruct, "cores": nthreads,
                                    "what": "oracle/refstruct_cpu.c: the same arithmetic behind EXP's data structure -- "
                                            "unordered_map-style hash of ~200-byte particle objects, level list of keys, "
                                            "Mass/Pos/AddAcc lookups per access, five pthread-forked passes per step "
                                            "(kick, drift, coefficients, force, kick); a restatement, not a build of EXP"},
            "tuned": {"value": vt, "cores": nthreads, "steps": st, "particles": big * 4,
                      "what": "oracle/tuned_cpu.c: cell moments + contraction, projected force table "
                              "(the device algorithm on CPU threads), gcc -O3"},
            "sample": f"{sn} KDK steps of {big} NFW particles on {nthreads} threads (contiguous "
                      f"slices, per-thread coefficient sums) and {s1} steps of {nsample} on 1 "
                      f"thread; same basis (lmax {grid.lmax}, nmax {grid.nmax}, numr {grid.numr}); "
                      "oracle/bfe_oracle.c, scalar fp64, gcc -O2"}


# ---- secondary configurations (BASELINE.json configs 2-4): parity-test workloads, reported as extras ----

def _counter_rows(out_dir, ctr, dom):
    """{kernel variant: [counter value of each launch]} of kernel `dom` from a rocprofv3 --pmc output directory."""
    import csv
    import glob
    vals = {}
    for path in glob.glob(os.path.join(out_dir, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.split("<")[0] == dom and row["Counter_Name"] == ctr:
                vals.setdefault(name, []).append(float(row["Counter_Value"]))
    return vals


def _traffic_from_counters(per):
    """HBM bytes per launch from {"FETCH_SIZE": {variant: [KiB per launch]}, "WRITE_SIZE": {...}}: the variants that run
    once per step (the most launches; a one-off variant such as an initial full pass is left out), first launch dropped,
    KiB -> bytes, the read side doubled (gfx950, MI355X_MICROARCH.md HBM section).  -> (total, per variant, launches)."""
    top = max(len(v) for v in per["FETCH_SIZE"].values())
    total = 0.0
    parts = {}
    for ctr, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for name, v in per[ctr].items():
            if len(v) != top:
                continue
            steady = v[1:] if len(v) > 2 else v
            b = sum(steady) / len(steady) * 1024.0 * scale
            parts.setdefault(name, {})[ctr] = b
            total += b
    return total, parts, top


def live_traffic(args, dom, nloc, budget_s=300.0):
    """HBM bytes per launch of kernel `dom`, from the PMC counters of THIS box: two child runs of this same command under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two do not fit one, MI355X_MICROARCH.md
    counter table), each with --kernel-trace only.  A process cannot read these counters about itself, and the profiler
    must start the program (its preloaded library initialises the GPU first), so the passes are children: `python3
    bench.py` directly behind `--`, started with subprocess (never exec'd).  Corrections as the guide's HBM section
    prescribes: both counters are KiB; on gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes, so the
    read side is doubled (the calibration of tools/summarize_profile.py on k_kick's known byte count gives 2.0000);
    WRITE_SIZE is exact.  The first launch of a kernel is dropped (first touch); template variants that run once per
    step each (the fast and the general pass of k_sph_force) are summed, as in profiles/traffic.json.
    Returns (bytes_per_launch or None, note)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="exp_amd_pmc_", dir="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
             "--no-cpu-baseline", "--no-other-configs", "--no-sustained", "--no-live-traffic",
             "--nbodies", repr(float(args.nbodies)), "--lmax", str(args.lmax), "--nmax", str(args.nmax),
             "--numr", str(args.numr), "--dt", repr(float(args.dt))]
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.perf_counter() - t0)
            if left < 20.0:
                return None, "time budget of the counter passes used up"
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child
            # (a session of its own: on a timeout the profiler AND the program it started are ended, by process group --
            # never by a pattern)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            if rc != 0:
                return None, "rocprofv3 --pmc %s: exit code %d" % (ctr, rc)
            vals = _counter_rows(out, ctr, dom)
            if not vals:
                return None, "no %s rows for %s" % (ctr, dom)
            per[ctr] = vals
        total, parts, top = _traffic_from_counters(per)
        note = ("measured in THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate child passes of this "
                "command (--steps 4 --warmup 2, %d launches each, first dropped), KiB -> bytes, FETCH_SIZE x 2 (gfx950 "
                "tallies wide coalesced reads at half their bytes), WRITE_SIZE as it is; %.0f s for both passes"
                % (top, time.perf_counter() - t0))
        return total, {"note": note, "per_variant_bytes": parts, "n_particles": int(nloc)}
    except subprocess.TimeoutExpired:
        return None, "a counter pass ran into its time budget"
    except Exception as e:          # the line must come out whatever the profiler does
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_disk(n, a, h, seed, device, vscale):
    """n exponential-disk particles (Sigma ~ exp(-R/a), sech^2(z/h): the reference's conditioning
    density, src/Cylinder.cc:315-322) on roughly circular orbits, generated in HBM."""
    import torch
    gen = torch.Generator(device=device).manual_seed(seed)
    f64 = torch.float64
    x = torch.linspace(0.0, 12.0, 16384, device=device, dtype=f64)
    cdf = 1.0 - (1.0 + x) * torch.exp(-x)
    cdf = cdf / cdf[-1]
    u = torch.rand(n, device=device, dtype=f64, generator=gen)
    idx = torch.searchsorted(cdf, u).clamp_(1, 16383)
    w = (u - cdf[idx - 1]) / (cdf[idx] - cdf[idx - 1])
    R = a * (x[idx - 1] + w * (x[idx] - x[idx - 1]))
    ph = torch.rand(n, device=device, dtype=f64, generator=gen) * (2 * math.pi)
    uz = torch.rand(n, device=device, dtype=f64, generator=gen).clamp_(1e-12, 1 - 1e-12)
    z = h * torch.atanh(2 * uz - 1)
    X, Y = (R * torch.cos(ph)).contiguous(), (R * torch.sin(ph)).contiguous()
    # roughly circular orbits + dispersion so that the cell order really changes every step
    vc = vscale * torch.sqrt(R / (R + a))
    vx = (-vc * torch.sin(ph) + 0.1 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    vy = (vc * torch.cos(ph) + 0.1 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    vz = (0.05 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    return X, Y, z.contiguous(), vx, vy, vz


def _timed(fn, steps, warmup):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def box_copy_rate(device, nbytes=1 << 30, reps=20):
    """What a plain device-to-device copy reaches on THIS box (read + write bytes over the time, GB/s): the boxes of the pool
    differ by 10-15 % in their HBM-bound passes (profiles/README.md), and this figure says which kind the line comes from.
    torch's copy kernel on torch's stream, timed with its events; not part of any timed region."""
    import torch
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
    b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def _kernel_fracs(prof, nsteps, n, basis, lmax):
    """per-kernel ms/step + the dominant kernel's algorithmic-byte and executed-fp64 fractions"""
    kern = {k: v["ms_total"] / nsteps for k, v in prof.items() if v["launches"] > 0}
    dom = max(kern, key=kern.get)
    out = {"kernels_ms_per_step": {k: round(v, 4) for k, v in kern.items()}, "dominant_kernel": dom}
    t = kern[dom] * 1e-3
    if dom in ALGO_BYTES:
        out["dominant_hbm_frac_algorithmic"] = ALGO_BYTES[dom] * n / t / 1e9 / HBM_PEAK_GBS
    if dom in OWN_BYTES:
        out["dominant_hbm_frac_own_bytes"] = OWN_BYTES[dom] * n / t / 1e9 / HBM_PEAK_GBS
    fl = EXEC_FLOPS.get((dom, lmax))
    if fl:
        out["dominant_fp64_frac_executed"] = fl * n / t / 1e12 / FP64_VECTOR_PEAK_TF
        out["dominant_executed_flops_per_particle"] = fl
    out["reference_flops_per_particle_step"] = REF_FLOPS[basis]
    return out


def other_configs(ctx, device, n, which=(2, 3, 4), steps=30, min_seconds=0.0):
    """BASELINE.json configs 2-4 on one GPU: one dict each."""
    import torch
    from exp_amd.empcyl import build_empcyl
    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from exp_amd.slgrid import build_slgrid
    model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
    grid = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)

    def halo(scale=1.0, vfac=1.0, mult=0):
        x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
        mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
        c = Component(ctx, n)
        ic = (mass, (x * scale).contiguous(), (y * scale).contiguous(), (z * scale).contiguous(),
              (vx * vfac).contiguous(), (vy * vfac).contiguous(), (vz * vfac).contiguous())
        c.upload_device(*ic)
        f = SphereSL(ctx, grid, scale=scale, rmin=grid.rmin * scale, rmax=grid.rmax * scale, multistep=mult)
        return c, f, ic

    def fused(c, f, dt, basis, lmax, label, ic=None):
        def start():
            if ic is not None:
                c.upload_device(*ic)           # back to the initial state (device-to-device copies)
            f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
        start()
        el = _timed(lambda: f.step_kdk(c, dt), steps, 3)
        ns = steps
        # A region of >= min_seconds in total, as bursts of `steps` steps that each start from the
        # INITIAL state: the synthetic sets are not equilibria (the disk spreads within two orbital
        # times: 1.05 -> 1.57 ms/step over 900 steps), and the configuration names the state.
        if min_seconds > 0.0 and ic is not None:
            nb = int(math.ceil(min_seconds / (el * steps)))
            tot = 0.0
            for _ in range(nb):
                start()
                tot += _timed(lambda: f.step_kdk(c, dt), steps, 3) * steps
            ns = nb * steps
            el = tot / ns
            start()
            for _ in range(3):
                f.step_kdk(c, dt)
        ctx.profile(True); ctx.profile_reset()
        for _ in range(4):
            f.step_kdk(c, dt)
        prof = ctx.profile_report()
        ctx.profile(False)
        o = {"config": label, "n": n, "timed_steps": ns, "ms_per_step": 1e3 * el, "particle_steps_per_s": n / el,
             "step_hbm_frac_232B": ALGO_BYTES["step"] * n / el / 1e9 / HBM_PEAK_GBS}
        o.update(_kernel_fracs(prof, 4, n, basis, lmax))
        c.close(); f.close()
        return o

    out = []
    if 2 in which:
        c, f, ic = halo()
        out.append(fused(c, f, 0.002, "S6", 6, "2: NFW halo, SphericalSL lmax 6 nmax 18, fused KDK step", ic))
        del ic
    a, h = 0.01, 0.001
    cg = None
    if 3 in which or 4 in which:
        # (C6 of SURVEY section 8d: ncylodd 4-6 -- eight vertically symmetric and four antisymmetric functions per m)
        cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=32,
                          nmaxfid=24, numr=2000, rnum=200, tnum=80, nodd=4)
    if 3 in which:
        X, Y, Z, vx, vy, vz = make_disk(n, a, h, 34567, device, vscale=7.0)
        mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
        c = Component(ctx, n)
        c.upload_device(mass, X, Y, Z, vx, vy, vz)
        out.append(fused(c, Cylinder(ctx, cg), 2e-5, "C6", 6,
                         "3: exponential disk, EmpCylSL mmax 6 nmax 12 (256x128 grid), fused KDK step",
                         (mass, X, Y, Z, vx, vy, vz)))
    if 4 in which:
        ms = 4
        ch, fh, ich = halo(scale=0.1, vfac=math.sqrt(10.0), mult=ms)      # a / rs = 0.1
        X, Y, Z, vx, vy, vz = make_disk(n, a, h, 34567, device, vscale=7.0)
        mass = torch.full((n,), 0.1 / n, device=device, dtype=torch.float64)
        cd = Component(ctx, n)
        cd.upload_device(mass, X, Y, Z, vx, vy, vz)
        fd = Cylinder(ctx, cg, multistep=ms)
        icd = (mass, X, Y, Z, vx, vy, vz)

        def new_sim():
            s_ = Simulation(ctx, 4e-4, multistep=ms)
            a1, a2 = s_.add_component(ch, fh), s_.add_component(cd, fd)
            s_.add_interaction(a1, a2)
            s_.add_interaction(a2, a1)
            s_.init()
            return s_

        sim = new_sim()
        nper = max(3, steps // 5)
        # the level populations settle over the first ~7 master steps after begin_run (every particle starts from the
        # assignment of the initial state: 43, 17, 11, 11, 10, 10, 9 ms ... in profiles/r03_cfg4_trace.txt); a run is
        # thousands of master steps long, so the timed ones come after that transient
        settle = 8
        el = _timed(lambda: sim.step(1), nper, settle)
        nm = nper
        if min_seconds > 0.0:
            # bursts of `nper` master steps, each from the initial state through begin_run (see fused())
            nb = int(math.ceil(min_seconds / (el * nper)))
            tot = 0.0
            for _ in range(nb):
                sim.close()
                ch.upload_device(*ich); cd.upload_device(*icd)
                sim = new_sim()
                tot += _timed(lambda: sim.step(1), nper, settle) * nper
            nm = nb * nper
            el = tot / nm
        lev_h = np.bincount(ch.download_levels(), minlength=ms + 1)
        lev_d = np.bincount(cd.download_levels(), minlength=ms + 1)
        # ... and once more, further along the same run: the populations of the upper levels keep thinning for a few dozen
        # master steps (level 1 of the halo: 2.5e5 particles at step 14, 8e4 at step 48), which is where a long run lives
        later = None
        if min_seconds > 0.0:
            nlate = 20
            el_late = _timed(lambda: sim.step(1), nlate, 40 - settle - nper if 40 - settle - nper > 0 else 0)
            later = {"ms_per_master_step": 1e3 * el_late, "timed_master_steps": nlate,
                     "untimed_master_steps_before": max(40, settle + nper),
                     "levels_halo": np.bincount(ch.download_levels(), minlength=ms + 1).tolist(),
                     "levels_disk": np.bincount(cd.download_levels(), minlength=ms + 1).tolist()}
        ctx.profile(True); ctx.profile_reset()
        sim.step(1)
        rep_ = ctx.profile_report()
        prof = {k: round(v["ms_total"], 3) for k, v in rep_.items() if v["launches"] > 0}
        prof_n = {k: int(v["launches"]) for k, v in rep_.items() if v["launches"] > 0}
        ctx.profile(False)
        sub = sum(int(lev_h[M] + lev_d[M]) * (1 << M) for M in range(ms + 1))
        # algorithmic bytes per particle-sub-step: 232 (its own step) + 32 (acc / pot read-modify-write
        # of the cross force applied to it), SURVEY.md section 8d
        out.append({"config": "4: disk + halo, SphericalSL lmax 6 nmax 18 + EmpCylSL mmax 6 nmax 12, "
                              "multistep 4, both self and both cross forces (C++ step driver)",
                    "n": 2 * n, "timed_master_steps": nm, "untimed_master_steps_before": settle,
                    "ms_per_master_step": 1e3 * el,
                    "master_step_particle_steps_per_s": 2 * n / el,
                    "raw_particle_substeps_per_s": sub / el,
                    "substeps_hbm_frac_264B": 264.0 * sub / el / 1e9 / HBM_PEAK_GBS,
                    "levels_halo": lev_h.tolist(), "levels_disk": lev_d.tolist(),
                    "later_in_the_run": later,
                    "level_switches_last_master_step": sim.step_switches,
                    "kernels_ms_per_master_step": prof, "kernel_scopes_per_master_step": prof_n})
        sim.close(); ch.close(); cd.close(); fh.close(); fd.close()
    return out


# BASELINE.json's metric, verbatim
METRIC = "particle-steps/sec (coef+force+kick-drift), 1e8 SphericalSL halo, 1/2/4/8 GPU"


def _flush_c_stdio():
    """RCCL prints a banner through C stdio; on a pipe it would only come out at exit -- after the
    JSON line, and from every rank under torchrun.  Push it out now."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process becomes the launcher -- one
    child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the layout `python -m torch.distributed.run
    --nproc-per-node N` gives (one rank per GPU, /root/reference/src/begin.cc:146-210).  Nothing here imports torch or
    touches HIP, the children are started with Popen (never exec), they inherit stdout (rank 0 alone prints the JSON line),
    and the first child to fail takes the others down: the exit code is non-zero then."""
    import socket
    import subprocess
    import time
    n = int(args.gpus)
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   EXP_AMD_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            c = p_.poll()
            if c is None:
                continue
            live.remove(p_)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {procs.index(p_)} exited with {c}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()         # (the exact children started above)
                t_end = time.time() + 10.0
                while any(q.poll() is None for q in live) and time.time() < t_end:
                    time.sleep(0.1)
                for q in live:
                    if q.poll() is None:
                        q.kill()
    return rc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # no launcher: be one (before torch / HIP are touched by this process)
            raise SystemExit(self_launch(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a launcher's world that is not the one asked for: measuring it would put a wrong n_gpus next to the rate
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch "
                         f"{args.gpus} ranks (or unset WORLD_SIZE and let bench.py start them)")
    probe = os.environ.get("EXP_AMD_BENCH_LAUNCH_PROBE")
    if probe is not None and os.environ.get("EXP_AMD_BENCH_SELF_LAUNCHED"):
        # test hook of the launcher (tests/test_bench_launch_cpu.py): "<code>[:<rank>]" -- say who this child is and leave
        code, _, who = probe.partition(":")
        print("probe rank=%s local_rank=%s world=%s addr=%s" % (os.environ["RANK"], os.environ["LOCAL_RANK"],
              os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"]), flush=True)
        raise SystemExit(int(code) if (not who or who == os.environ["RANK"]) else 0)
    # (multi-process GPU work on this pool needs dmabuf IPC: already exported where the driver runs; kept for any other shell)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = bool(args.rehearse_shared_gpu)
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_comm = world > 1 or args.force_comm
    # tensors handed to torch.distributed (the vote, the MAX of the timings): on the device for RCCL, on the host for gloo
    cdev = torch.device("cpu") if rehearse else device
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def reduce_max(v):
        """MAX over the ranks of one host number (the timed regions)"""
        if not use_comm:
            return float(v)
        t_ = torch.tensor([v], device=cdev, dtype=torch.float64)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, Context, SphereSL
    from exp_amd.slgrid import build_slgrid

    # ---- basis tables (init; not timed) -------------------------------------------------------
    model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
    grid = build_slgrid(model, args.lmax, args.nmax, numr=args.numr, rmin=1e-3, rmax=49.5,
                        cmap=1, rmap=1.0)

    # ---- particles: static block shard of the total -----------------------------------------------
    if args.scaling == "weak":
        nloc = int(args.nbodies)
        ntot = nloc * world
        x, y, z, vx, vy, vz = make_halo(model, nloc, seed=23456 + rank, device=device)
    else:
        # strong scaling: ONE particle set whatever the rank count -- every rank draws the whole set from the same seed
        # and keeps its block, so that an N-rank run is the 1-rank run sharded (selfcheck.coef_00_0 agrees to rounding)
        ntot = int(args.nbodies)
        n0 = ntot * rank // world
        n1 = ntot * (rank + 1) // world
        nloc = n1 - n0
        full = make_halo(model, ntot, seed=23456, device=device)
        x, y, z, vx, vy, vz = full if world == 1 else tuple(a[n0:n1].clone() for a in full)
        del full
    mass = torch.full((nloc,), 1.0 / ntot, device=device, dtype=torch.float64)
    torch.cuda.synchronize()

    # the context runs on an explicit torch stream so that torch.distributed's all-reduce
    # (issued from the coefficient callback) is ordered with the kernels
    tstream = torch.cuda.Stream(device)
    torch.cuda.set_stream(tstream)
    ctx = Context(dev_index, stream=tstream.cuda_stream)
    if args.split:
        ctx.set_split_min(1)
    ctx.set_append_min(0 if (args.no_append or args.split or args.graph) else 1 << 20)          # (2^20 is the library's default)
    ctx.set_append_lean(bool(args.append_lean))
    step_form = "ordinary" if (args.no_append or args.split or args.graph or nloc < (1 << 20)) else "append"
    comp = Component(ctx, nloc)
    comp.upload_device(mass, x, y, z, vx, vy, vz)
    del x, y, z, vx, vy, vz, mass
    torch.cuda.empty_cache()
    force = SphereSL(ctx, grid)

    comm_used = "none"
    comm_note = None
    if use_comm:
        comm_used = None
        if args.comm == "rccl":
            # the library's own communicator: id from rank 0 through the process group, then a
            # known-answer all-reduce (every rank contributes rank + 1) before it is trusted
            ok = False
            if rehearse and world > 1:
                # RCCL refuses two ranks on one device: not attempted; every rank votes "failed", which takes the
                # same vote + fallback path a real failure takes
                comm_note = "native RCCL communicator not attempted (ranks share device 0): voted down"
            else:
                try:
                    ids = [Context.rccl_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    ctx.init_rccl(ids[0], world, rank)
                    with torch.cuda.stream(tstream):
                        probe = torch.full((64,), float(rank + 1), device=device, dtype=torch.float64)
                        ctx.allreduce(probe.data_ptr(), probe.numel())
                        tstream.synchronize()
                    ok = bool((probe == world * (world + 1) / 2.0).all().item())
                except Exception as e:          # pragma: no cover
                    print(f"[bench] rank {rank}: native RCCL communicator failed ({e!r})", file=sys.stderr)
                    ok = False
                # test hook: EXP_AMD_BENCH_FAIL_RCCL = "all" or a rank number makes that rank vote "failed" AFTER the
                # (collective) set-up, so that the fallback below can be exercised where RCCL works
                forced = os.environ.get("EXP_AMD_BENCH_FAIL_RCCL")
                if forced and (forced == "all" or forced == str(rank)):
                    ok = False
                    comm_note = f"native RCCL communicator voted down by EXP_AMD_BENCH_FAIL_RCCL={forced}"
            flag = torch.tensor([1 if ok else 0], device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comm_used = "rccl (library communicator, ncclAllReduce on the compute stream)"
            elif rank == 0:
                print("[bench] falling back to the all-reduce callback", file=sys.stderr)
        if comm_used is None:
            fb = " (fallback)" if args.comm == "rccl" else ""
            if rehearse:
                from exp_amd.dist import host_staged_allreduce_callback
                ctx.set_allreduce(host_staged_allreduce_callback(), world, rank)
                comm_used = "host-staged all-reduce callback through gloo (rehearsal on a shared GPU)" + fb
            else:
                from exp_amd.dist import torch_allreduce_callback
                ctx.set_allreduce(torch_allreduce_callback(device), world, rank)
                comm_used = "torch.distributed all-reduce callback" + fb

    def barrier():
        if use_comm:
            dist.barrier()
        torch.cuda.synchronize()

    # initial coefficients + accelerations so that step 1 kicks with a real field
    force.determine_coefficients(comp)
    comp.zero_acceleration(0)
    force.get_acceleration_and_potential(comp)

    for _ in range(args.warmup):
        force.step_kdk(comp, args.dt)
    barrier()
    _flush_c_stdio()            # (every rank: the communicator exists by now)
    graph_region = None
    if args.graph:
        # the K steps as one call, replayed from the graph of a pair of steps (no per-kernel events: they cannot be
        # captured); the eager region below then only supplies the per-kernel breakdown
        force.step_kdk_n(comp, args.dt, 4)          # capture + one replay, untimed
        barrier()
        tg = time.perf_counter()
        force.step_kdk_n(comp, args.dt, args.steps)
        barrier()
        eg = reduce_max(time.perf_counter() - tg)
        graph_region = {"steps": args.steps, "seconds": eg, "ms_per_step": 1e3 * eg / args.steps}
    ctx.profile(True)
    ctx.profile_reset()
    # per-step device times: one event per step boundary on the context's stream (SURVEY 8d timing protocol:
    # the median of these is reported beside the mean of the wall-clock region that `value` is)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record(tstream)
    for k in range(args.steps):
        force.step_kdk(comp, args.dt)
        marks[k + 1].record(tstream)
    barrier()
    el = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile(False)
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    step_times = {"median_ms": step_ms[len(step_ms) // 2] if len(step_ms) % 2 else
                  0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
                  "min_ms": step_ms[0], "max_ms": step_ms[-1],
                  "what": "hipEvent time of each of the K timed steps on the context's stream (per-kernel events on)"}

    # A second, longer region (>= 2 s of steps, no per-kernel events): `value` stays the K steps the
    # contract asks for; this shows that the rate holds when the region is not a quarter of a second.
    sustained = None
    if not args.no_sustained:
        # (the step count from the SLOWEST rank's region: every rank must run the same number of steps -- each carries an
        # all-reduce --, and a count formed from a rank's own clock differs between ranks by a step or two)
        ns = max(args.steps, int(math.ceil(2.0 / max(reduce_max(el) / args.steps, 1e-6))))
        barrier()
        t1 = time.perf_counter()
        if args.graph:
            force.step_kdk_n(comp, args.dt, ns)
        else:
            for _ in range(ns):
                force.step_kdk(comp, args.dt)
        barrier()
        es = reduce_max(time.perf_counter() - t1)
        sustained = {"steps": ns, "seconds": es, "ms_per_step": 1e3 * es / ns, "value": ntot * ns / es}

    # The same K steps once more with the append step's LEAN payload (exp_amd_ctx_set_append_lean: acceleration and potential not
    # placed, re-evaluated for the first call that looks) -- reported beside the line, never as `value`; then back to the full
    # payload, so that the state checked below is the default step's.
    lean_ab = None
    # (skipped with --no-sustained as well: the profiling tools and the counter passes of this file measure the default step alone)
    if step_form == "append" and not args.append_lean and not args.no_lean_ab and not args.no_sustained and not args.graph:
        ctx.set_append_lean(True)
        for _ in range(3):
            force.step_kdk(comp, args.dt)
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            force.step_kdk(comp, args.dt)
        barrier()
        e2 = reduce_max(time.perf_counter() - t2)
        ctx.set_append_lean(False)
        for _ in range(2):
            force.step_kdk(comp, args.dt)
        barrier()
        lean_ab = {"steps": args.steps, "ms_per_step": 1e3 * e2 / args.steps, "value": ntot * args.steps / e2,
                   "what": "the K steps again with exp_amd_ctx_set_append_lean(1): the placing pass stores neither acceleration "
                           "nor potential (32 of 88 B a particle), which the first call that looks at the particles has "
                           "re-evaluated from the coefficient set kept at the completed step; opt-in, not the default, not `value`"}

    # Full-size sanity of what was just timed (size-independent properties, no oracle): every
    # particle is inside the expansion window by construction, a self-gravitating system's
    # total force vanishes (sum m a: here up to the expansion's truncation), its centre of mass
    # stays put, and the monopole coefficient of a unit-mass halo has the same sign/size at any N.
    used = force.Used()
    com = comp.fix_positions(0)          # all-reduced over ranks like the coefficients
    coef00 = float(force.get_coefs()[0, 0])
    selfcheck = {"used_rank0": int(used), "particles_rank0": int(nloc),
                 "center_of_mass": [float(v) for v in com["com"]],
                 "center_of_acceleration": [float(v) for v in com["coa"]],
                 "mtot": com["mtot"], "coef_00_0": coef00}

    el = reduce_max(el)
    comm_info = ctx.comm_info()          # (every rank: a query only -- no collective behind it)

    if rank == 0:
        if graph_region:                # --graph: the timed region is the replayed one
            el_eager, el = el, graph_region["seconds"]
        value = ntot * args.steps / el
        ms_step = 1e3 * el / args.steps
        # dominant kernel by measured time on this rank's stream
        kern = {k: v for k, v in prof.items() if v["launches"] > 0}
        dom = max(kern, key=lambda k: kern[k]["ms_total"]) if kern else None
        roof = None
        box_copy = None
        try:
            box_copy = box_copy_rate(device)
        except Exception:
            box_copy = None
        if dom:
            # k_sph_accumulate is launched once per m-split per step: a "launch" here is the
            # whole split group (ProfScope brackets the group)
            avg_ms = kern[dom]["ms_total"] / kern[dom]["launches"]
            algo = ALGO_BYTES.get(dom, 0.0) * nloc
            achieved = algo / (avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    ent = tj.get(dom)
                    if ent and int(ent.get("n_particles", -1)) == nloc:
                        traffic = ent["hbm_bytes_per_launch"]
                except Exception:
                    traffic = None
            traffic_src = ("REPLAYED from profiles/traffic.json, not measured in this run: rocprofv3 --pmc "
                           "FETCH_SIZE / WRITE_SIZE, separate passes, of this same command on an earlier box "
                           "(tools/profile.sh)") if traffic else None
            traffic_extra = None
            if world == 1 and not args.no_live_traffic and not args.no_cpu_baseline:
                replayed = traffic
                lt, info = live_traffic(args, dom, nloc)
                if lt is not None:
                    traffic, traffic_src = lt, info["note"]
                    traffic_extra = {"per_variant_bytes": info["per_variant_bytes"],
                                     "replayed_from_profiles_traffic_json": replayed}
                else:
                    traffic_src = "%s; the live counter passes did not run: %s" % (traffic_src, info)
            # bound / achieved / peak / unit / frac are the contract's HBM accounting (algorithmic bytes over the kernel's
            # time); the roof the kernel sits closest to -- the larger of the HBM fraction and the executed-fp64
            # fraction -- is DERIVED below as `binding_limit`
            roof = {"bound": "hbm", "frac_of": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_particle": ALGO_BYTES.get(dom, 0.0),
                    "step_achieved": ALGO_BYTES["step"] * nloc / (ms_step * 1e-3) / 1e9,
                    "step_frac": ALGO_BYTES["step"] * nloc / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    # the same two fractions against the rate a plain streaming kernel measures on MI355X
                    "measured_hbm_peak": HBM_MEASURED_GBS,
                    "box_copy_gbs": box_copy,
                    "frac_of_measured_peak": achieved / HBM_MEASURED_GBS,
                    "step_frac_of_measured_peak": ALGO_BYTES["step"] * nloc / (ms_step * 1e-3) / 1e9 / HBM_MEASURED_GBS,
                    "kernels_ms_per_step": {k: v["ms_total"] / args.steps for k, v in kern.items()}}
            if traffic_extra:
                roof["traffic_detail"] = traffic_extra
                roof["traffic_over_algorithmic"] = traffic / algo if algo else None
            # what the kernel itself moves in the fused step (the contract's figure counts the v store
            # of the closing half-kick, which lives in the next scatter pass here)
            if dom in OWN_BYTES:
                ownb = ((OWN_BYTES_APPEND_LEAN if args.append_lean else OWN_BYTES_APPEND).get(dom, OWN_BYTES[dom])
                        if step_form == "append" else OWN_BYTES[dom])
                own = ownb * nloc / (avg_ms * 1e-3) / 1e9
                roof["kernel_own_bytes_per_particle"] = ownb
                roof["kernel_own_achieved"] = own
                roof["kernel_own_frac"] = own / HBM_PEAK_GBS
                if traffic:
                    # the counters against what THIS kernel has to move: the figure that says whether anything is re-read
                    roof["traffic_over_own_bytes"] = traffic / (ownb * nloc)
                    if step_form == "append":
                        roof["traffic_note"] = (
                            "append form: this kernel is the contract's pass B (104 B) AND pass A's advance and the reorder -- it "
                            "reads x, v, id (52 B) and writes the next position, the velocities, acceleration, potential, id and the "
                            f"source slot ({ownb - 52:.0f} B): {ownb:.0f} B of its own.  traffic / own = "
                            f"{traffic / (ownb * nloc):.2f} (nothing is re-read); `traffic_over_algorithmic` holds the counters against "
                            "pass B's 104 B alone, and `achieved` / `frac` stay on those 104 B although the kernel now does more per "
                            "launch -- the step's fraction (`step_frac`, 232 B a particle-step) is the one that rose with this form")
            # The binding limit of this kernel is the fp64 vector ALU: executed
            # flops per particle from the ISA (tools/isa_count.py).
            fl = EXEC_FLOPS.get((dom, args.lmax))
            if fl:
                tf = fl * nloc / (avg_ms * 1e-3) / 1e12
                roof["fp64_vector"] = {"achieved": tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": tf / FP64_VECTOR_PEAK_TF, "flops_per_particle": fl}
                own_frac = roof.get("kernel_own_frac", roof["frac"])
                roof["binding_limit"] = "fp64_vector" if tf / FP64_VECTOR_PEAK_TF > max(own_frac, roof["frac"]) else "hbm"
                # `bound` stays the contract's value ("hbm" | "mfma": the roof achieved / peak / unit / frac are quoted against --
                # "nominally HBM, and that is the figure to report"); the roof the kernel actually sits closest to is
                # `binding_limit`, with its own achieved / peak / frac in `fp64_vector`
                roof["bound_note"] = ("achieved / peak / frac are the HBM accounting the contract asks for; the kernel's binding "
                                      "limit is `binding_limit` (fp64 vector issue: `fp64_vector`)")
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(grid, model, args.cpu_sample, args.dt)
        others = None
        if not args.no_other_configs and world == 1:
            # free the headline's 19 GB first; these are extras, a failure must not lose the line
            comp.close(); force.close()
            try:
                others = other_configs(ctx, device, int(args.other_n), min_seconds=1.0)
            except Exception as e:      # pragma: no cover
                others = [{"error": repr(e)}]
        line = {
            "metric": METRIC,
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "step_times": step_times,
            "stepping": ("exp_amd_step_kdk_n: pairs of steps replayed from a HIP graph (eager region with per-kernel "
                         f"events: {1e3 * el_eager / args.steps:.4f} ms/step)") if graph_region else
                        ("exp_amd_step_kdk per step (eager launches, per-kernel events on" +
                         ("; APPEND form: no sort passes, the force pass places every particle in the next step's cell order)"
                          if step_form == "append" else ")") +
                         ("; SPLIT fused step: two half stores, sort passes on a second stream" if args.split else "")),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{ntot:.0e}-particle truncated-NFW halo, SphericalSL "
                                   f"lmax={args.lmax} nmax={args.nmax} numr={args.numr}, "
                                   "multistep=0, KDK step (kick/2, drift, sort, coef, force, kick/2)",
                       # "append": no sort passes -- the force pass drifts every particle and places it in the next step's cell
                       # order (exp_amd_ctx_set_append_min); "ordinary": key histogram, scan and scatter pass every step
                       "step_form": step_form,
                       # "full": the placing pass stores the complete state (next position, velocities, acceleration, potential);
                       # "lean" (--append-lean): acceleration and potential re-evaluated for whoever looks at the particles
                       "append_payload": (None if step_form != "append" else "lean" if args.append_lean else "full"),
                       "nbodies_total": ntot, "nbodies_per_gpu": nloc, "lmax": args.lmax,
                       "nmax": args.nmax, "numr": args.numr, "dt": args.dt,
                       "parallelism": f"particle-shard x{world}, 1 coef all-reduce/step"
                                      if world > 1 else "single GPU",
                       # which all-reduce really ran, the rank count it saw and how often it was issued
                       "comm": {"path": comm_used, "note": comm_note, **comm_info},
                       "rehearsal": ("ranks time-share ONE GPU (gloo process group, host-staged all-reduce): a dress "
                                     "rehearsal of the launch, not a scaling measurement") if rehearse else None},
            "roofline": roof,
            "cpu_baseline": cpu,
            "selfcheck": selfcheck,
            "sustained": sustained,
            "append_lean_ab": lean_ab,
            "other_configs": others,
        }
        _flush_c_stdio()        # the JSON line is the last thing on stdout
        print(json.dumps(line), flush=True)

    comp.close()           # (idempotent)
    force.close()
    ctx.close()
    if use_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
