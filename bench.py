#!/usr/bin/env python3
"""bench.py -- V-cycles/s of the MI355X-native geometric-multigrid hot path, with the
finest-grid smoother priced against the HBM roofline and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json metric "V-cycles/sec + finest-grid smoother GB/s, 3D Poisson
512^3"): 3-D Poisson on 513^3 nodes (nominal 512^3: vertex-centred grid, boundary nodes
included), 6-level V(2,2), damped Jacobi (omega 6/7) on every level, full-
weighting restriction, coarsest grid (17^3) iterated to relative residual 0.1 like the
reference's Solver (include/solvers.hpp:324-342; ~85 sweeps, LDS-resident kernel), fp64, zero
initial guess, hash-noise right-hand side (synthetic). A step is one V-cycle.

One JSON line is printed by rank 0. `roofline` prices the dominant kernel (finest-grid
Jacobi sweep: 24 B of compulsory traffic per grid point -- read u, read rhs, write u')
from HIP events recorded on the library's own stream INSIDE the timed region. When the library
fuses the V(2,2) sweep pairs (k_jacobi2, two sweeps per launch) a launch processes two sweeps'
worth of algorithmic bytes: `achieved` stays per-sweep-equivalent (24 B x points / sweep_ms) and
`launch_ms` = 2 x `sweep_ms` is what rocprofv3 shows for that kernel.
`cpu_baseline` times the CPU oracle (our restatement of the reference algorithm; the
reference itself has no 3-D path) on the host cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=513, help="nodes per side (513 = nominal 512^3)")
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--smoother", choices=["jacobi", "rbgs", "zebra"], default="jacobi",
                    help="zebra = zebra line Gauss-Seidel along y (for --aniso-y >> 1)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cycles", type=int, default=1, help="oracle V-cycles timed for cpu_baseline")
    ap.add_argument("--transport", choices=["rccl", "gloo"], default="rccl",
                    help="N>1 halo transport: rccl = GPU-to-GPU over xGMI (one GPU per rank, the real thing); "
                         "gloo = rehearsal through host memory, ranks may share a GPU (numbers meaningless)")
    ap.add_argument("--dist-min-n", type=int, default=0, help="mg_desc.dist_min_n (0 = library default)")
    ap.add_argument("--aniso-y", type=float, default=1.0, help="y-coupling multiplier of -(dxx + a dyy + dzz)")
    ap.add_argument("--aniso-eps", type=float, default=1.0, help="z-coupling multiplier eps of -(dxx + dyy + eps dzz)")
    ap.add_argument("--semi", type=int, default=0, help="number k of leading x,y-only coarsenings (mg_desc.semi_xy); "
                    "BASELINE config 5: --aniso-eps 0.01 --semi 3 --levels 8 --smoother rbgs")
    return ap.parse_args()


def workload_desc(mod, a):
    return mod.make_desc(
        dim=3, n=a.n, levels=a.levels, dtype=mod.MG_F64 if a.dtype == "f64" else mod.MG_F32,
        length=1.0, alpha=1.0, cycle=mod.CYCLE_V,
        smoother={"jacobi": mod.SMOOTH_JACOBI, "rbgs": mod.SMOOTH_RBGS, "zebra": mod.SMOOTH_ZEBRA_Y}[a.smoother],
        omega=6.0 / 7.0 if a.smoother == "jacobi" else 1.0, nu_pre=2, nu_post=2,
        restriction=mod.RESTRICT_FULLW,
        **({} if a.semi else {"coarse_mode": mod.COARSE_TOL, "coarse_maxit": 2000, "coarse_tol": 0.1}),
        outer_pre_gs=0, dist_min_n=a.dist_min_n, aniso=(1.0, a.aniso_y, a.aniso_eps), semi_xy=a.semi,
        **({"coarse_mode": mod.COARSE_FIXED, "coarse_maxit": 20} if a.semi else {}))


def hash_rhs(n, dtype, z0=0, nz=None, seed=12345):
    """Same counter-based noise as the oracle's orc_fill_rhs_3d(kind=1); numpy, slab-wise."""
    nz = n if nz is None else nz
    out = np.zeros((nz, n, n), dtype)
    jj, ii = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64), indexing="ij")
    inner = (jj > 0) & (jj < n - 1) & (ii > 0) & (ii < n - 1)
    with np.errstate(over="ignore"):
        for k in range(nz):
            gk = z0 + k
            if gk == 0 or gk == n - 1:
                continue
            idx = (np.uint64(gk) * np.uint64(n) + jj) * np.uint64(n) + ii
            z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            v = (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
            out[k] = np.where(inner, v, 0.0)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    from multigrid_prj_amd import capi

    if world > 1:
        # a communication problem must end the run, not hang the node: every rank aborts
        # itself if the whole benchmark has not finished in time
        import signal
        signal.alarm(900)

    dist = None
    comm_id = None
    host_comm = None
    device = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        if a.transport == "rccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            ids = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm_id = ids[0]
        else:
            from multigrid_prj_amd.dist import torch_host_comm
            device = local_rank % max(capi.device_count(), 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            host_comm = torch_host_comm()

    desc = workload_desc(capi, a)
    s = capi.Solver(desc, device=device, rank=rank, nranks=world, comm_id=comm_id, host_comm=host_comm)
    z0, nz, first_gathered = capi.plan_slab(desc, world, rank, 0)
    npdt = np.float64 if a.dtype == "f64" else np.float32
    s.set_rhs(hash_rhs(a.n, npdt, z0, nz))
    s.zero_array(capi.ARR_U, 0)

    def barrier():
        s.sync()
        if dist is not None:
            import torch
            dist.barrier()
            if a.transport == "rccl":
                torch.cuda.synchronize()

    # convergence sanity of the benchmarked cycle (not timed), from the zero guess so the ratio is not
    # taken at the round-off floor: asymptotic residual reduction per cycle. Done BEFORE the timed
    # region: it also brings the clocks up and loads every kernel, so that short runs (small W) measure
    # the same steady state as long ones; u is zeroed again afterwards.
    hist, _ = s.solve(0.0, 4)
    s.zero_array(capi.ARR_U, 0)
    barrier()
    # warmup (untimed)
    s.cycle_async(a.warmup)
    barrier()
    # timed region: exactly K cycles, finest-grid smoother bracketed by HIP events
    s.profile_begin()
    t0 = time.perf_counter()
    s.cycle_async(a.steps)
    barrier()
    t1 = time.perf_counter()
    sm_ms, sm_sweeps = s.profile_end()
    fu_ms, fu_sweeps = s.profile_fused()   # post-smoothing pairs that also carried the prolongation
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if a.transport == "rccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / a.steps
    cycles_per_s = a.steps / elapsed
    esz = 8 if a.dtype == "f64" else 4
    pts_local = nz * a.n * a.n
    bytes_per_sweep = 3 * esz * pts_local          # read u, read rhs, write u'
    sweep_ms = sm_ms / max(sm_sweeps, 1)
    achieved = bytes_per_sweep / (sweep_ms * 1e-3) / 1e9 if sm_sweeps else 0.0

    traffic, traffic_src = profiled_traffic(a) if world == 1 else (None, None)
    # On a whole (non-distributed) level whose rows are 64/128/256 vectors wide the library runs
    # the V(2,2) sweep pairs as ONE launch (k_jacobi2: two sweeps per pass over HBM).
    vec = 2 if a.dtype == "f64" else 4
    fused_pair = (a.smoother == "jacobi" and world == 1 and (a.n - 1) % vec == 0 and (a.n - 1) // vec in (64, 128, 256)
                  and os.environ.get("MG_FUSED_PAIR", "1") != "0")
    sweeps_per_launch = 2 if fused_pair else 1
    out = {
        "metric": f"V-cycles/sec (3D Poisson {a.n}^3 V(2,2)) + finest-grid smoother GB/s vs HBM roofline",
        "value": cycles_per_s, "unit": "V-cycles/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"3D Poisson {a.n}^3 nodes (nominal {a.n - 1}^3), {a.levels}-level V(2,2), "
                               f"{a.smoother}{' omega=6/7' if a.smoother == 'jacobi' else ''}, full-weighting, "
                               f"{f'eps={a.aniso_eps}, first {a.semi} coarsenings in x,y only, ' if a.semi or a.aniso_eps != 1.0 else ''}"
                               f"{f'y-coupling x{a.aniso_y}, ' if a.aniso_y != 1.0 else ''}"
                               + (f"20 coarse sweeps, {a.dtype}" if a.semi else
                                  f"coarse {((a.n - 1) >> (a.levels - 1)) + 1}^3 iterated to rel. residual 0.1, {a.dtype}"),
                   "parallelism": (f"z-slab x{world}" + ("" if a.transport == "rccl" else " (gloo rehearsal)")) if world > 1 else "single GPU",
                   "first_gathered_level": first_gathered},
        "roofline": {"bound": "hbm",
                     "kernel": (f"finest-grid fused double Jacobi sweep k_jacobi2 ({a.n}^3, 2 sweeps per launch)" if fused_pair
                                else f"finest-grid {a.smoother} sweep ({a.n}^3)"),
                     "sweeps_per_launch": sweeps_per_launch, "launch_ms": sweep_ms * sweeps_per_launch,
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "sweep_ms": sweep_ms, "sweeps_timed": sm_sweeps,
                     "algorithmic_bytes_per_sweep": bytes_per_sweep,
                     "note": ("temporal blocking: one launch performs two sweeps in one pass over HBM, so the algorithmic "
                              "two-sweep bytes per launch can exceed what a streaming kernel could move; `traffic` is the "
                              "measured fabric-side traffic of that launch" if fused_pair else None)},
        "smoother_gbps": achieved,
        # one launch = prolong-add (read u, read coarse, write u: not executed as such) + two sweeps
        "prolong_folded_pair": ({"kernel": ("k_jacobi2<CORR>: J(J(u + P e)) in one pass" if a.smoother == "jacobi"
                                            else "post-smoothing segment: RB(u + P e) in one pass, then the remaining sweep(s)"),
                                 "launch_ms": 2 * fu_ms / fu_sweeps,
                                 "launches_timed": fu_sweeps // 2,
                                 "replaces_bytes": 2 * bytes_per_sweep + int(2.125 * esz * pts_local),
                                 "equivalent_gbps": (2 * bytes_per_sweep + 2.125 * esz * pts_local) / (2 * fu_ms / fu_sweeps * 1e-3) / 1e9}
                                if fu_sweeps else None),
        "residual_drop_per_cycle": float(hist[-1] / hist[-2]) if len(hist) >= 2 and hist[-2] > 0 else None,
        "device_bytes": s.device_bytes(),
    }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    if rank == 0:
        print(json.dumps(out))
    s.close()
    if dist is not None:
        dist.destroy_process_group()


def profiled_traffic(a):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    this same command (profiles/r01_kernel_summary.csv, made by tools/profile.sh +
    tools/summarize_prof.py: separate --pmc FETCH_SIZE / WRITE_SIZE passes, read = 2 x FETCH_SIZE
    x 1024 on gfx950). Counters cannot be collected inside a timed run, so this is not live;
    None when the workload is not the profiled default."""
    if not (a.n == 513 and a.dtype == "f64" and a.smoother == "jacobi" and a.levels == 6 and not a.semi):
        return None, None
    fused = os.environ.get("MG_FUSED_PAIR", "1") != "0"
    path = os.path.join(ROOT, "profiles", "r01_kernel_summary.csv")
    if not os.path.exists(path):
        return None, None
    import csv
    best = None
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith("k_jacobi2<double" if fused else "k_sweep3d<double, 0,") and r["read_MB"] and r["write_MB"]:
            if best is None or int(r["grid_threads"]) > int(best["grid_threads"]):
                best = r
    if best is None:
        return None, None
    return (float(best["read_MB"]) + float(best["write_MB"])) * 1e6, "profiles/r01_kernel_summary.csv (rocprofv3 --pmc passes of this command)"


def cpu_baseline(a):
    """The oracle (kind "port": our CPU restatement -- the reference has no 3-D path and
    cannot travel to the GPU box) on the host cores: the SAME workload, `cpu_cycles`
    V-cycles, OpenMP over all host threads."""
    from oracle import pyoracle as po
    d = workload_desc(po, a)
    o = po.Solver(d)
    o.set_rhs(po.fill_rhs_3d(a.n, 1.0, 1.0, 1))
    t0 = time.perf_counter()
    for _ in range(a.cpu_cycles):
        o.cycle()
    dt = time.perf_counter() - t0
    # one finest-grid sweep alone, for a like-for-like smoother figure
    t1 = time.perf_counter()
    o.smooth_fine(d.smoother, 1)
    ds = time.perf_counter() - t1
    esz = 8 if a.dtype == "f64" else 4
    o.close()
    return {"value": a.cpu_cycles / dt, "unit": "V-cycles/s", "cores": po.lib().orc_omp_threads(), "kind": "port",
            "sample": f"{a.cpu_cycles} V-cycle(s) of the same {a.n}^3 workload ({dt:.1f} s) with the OpenMP oracle",
            "smoother_gbps": 3 * esz * a.n ** 3 / ds / 1e9}


if __name__ == "__main__":
    main()
