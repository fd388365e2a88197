#!/usr/bin/env python3
"""bench.py -- V-cycles/s of the MI355X-native geometric-multigrid hot path, with the
finest-grid kernels priced against the HBM roofline and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W

N > 1 without RANK/WORLD_SIZE in the environment: this process only LAUNCHES -- it starts one
child per GPU (env RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), never
touches the GPU or imports torch itself, relays rank 0's JSON line and exits non-zero if any
child fails or the deadline passes. Under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` (WORLD_SIZE set) every process is a rank right away.

Workload (BASELINE.json metric "V-cycles/sec + finest-grid smoother GB/s, 3D Poisson
512^3"): 3-D Poisson on 513^3 nodes (nominal 512^3: vertex-centred grid, boundary nodes
included), 6-level V(2,2), damped Jacobi (omega 6/7) on every level, full-
weighting restriction, coarsest grid (17^3) iterated to relative residual 0.1 like the
reference's Solver (include/solvers.hpp:324-342; ~85 sweeps, LDS-resident kernel), fp64, zero
initial guess, hash-noise right-hand side (synthetic). A step is one V-cycle.

One JSON line is printed by rank 0. `roofline` prices the finest-grid smoother launch from HIP
events recorded on the library's own stream INSIDE the timed region:
  achieved = bytes the launch MUST move / launch time.  A streaming Jacobi sweep must move
  24 B per point (read u, read rhs, write u'); the fused pair k_jacobi2 makes ONE pass for two
  sweeps, so it must move the same 24 B per point -- `frac` is that over the 8 TB/s HBM peak and is
  <= 1 by construction. What two streaming sweeps would have had to move per second of this
  launch is reported separately as `sweep_equivalent_gbps` (it may exceed the peak: temporal
  blocking). `traffic` = fabric-side bytes per launch from the committed rocprofv3 PMC passes of
  this same command (profiles/), `traffic_frac` = traffic / launch time / peak.
`kernels` prices the other finest-level launches of the cycle the same way, `streaming_sweep` times
single (unfused) Jacobi sweeps of the same grid right after the timed region.
`cpu_baseline` times the CPU oracle (our restatement of the reference algorithm; the
reference itself has no 3-D path) on the host cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=513, help="nodes per side (513 = nominal 512^3)")
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--smoother", choices=["jacobi", "rbgs", "zebra", "zebrax"], default="jacobi",
                    help="zebra / zebrax = zebra line Gauss-Seidel along y / along x (for --aniso-y / --aniso-x >> 1)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cycles", type=int, default=8, help="oracle V-cycles timed for cpu_baseline (~1 s each at 513^3)")
    ap.add_argument("--transport", choices=["rccl", "gloo"], default="rccl",
                    help="N>1 halo transport: rccl = GPU-to-GPU over xGMI (one GPU per rank, the real thing); "
                         "gloo = rehearsal through host memory, ranks may share a GPU (numbers meaningless)")
    ap.add_argument("--dry-rank", type=int, default=-1,
                    help="MEASUREMENT TOOL, one process, one GPU: run rank R of --gpus N with no peers (every exchange is a "
                         "no-op, results meaningless) to time that rank's compute schedule; the line is marked dry_run")
    ap.add_argument("--dry-raw-coarse", action="store_true",
                    help="with --dry-rank: leave the coarse solve uncapped. By default the dry run caps its coarse-grid sweeps at the "
                         "count the real problem takes (measured first on this GPU with the single-GPU solver, same cycles): without "
                         "exchanges the slab's data are inconsistent at its faces and the 17^3 solve needs 1.4x the sweeps, which "
                         "k ranks == 1 rank rules out for a real run")
    ap.add_argument("--rccl-same-gpu", action="store_true",
                    help="REHEARSAL on a one-GPU box: every rank uses GPU 0 and tells RCCL it sits on a host of its own (NCCL_HOSTID), "
                         "so that RCCL accepts two ranks on one device and carries the messages over its socket transport on the "
                         "loopback interface -- the real RcclComm code path with N real ranks; the numbers mean nothing")
    ap.add_argument("--dist-min-n", type=int, default=0, help="mg_desc.dist_min_n (0 = library default)")
    ap.add_argument("--aniso-y", type=float, default=1.0, help="y-coupling multiplier of -(dxx + a dyy + dzz)")
    ap.add_argument("--aniso-x", type=float, default=1.0, help="x-coupling multiplier of -(a dxx + dyy + dzz)")
    ap.add_argument("--aniso-eps", type=float, default=1.0, help="z-coupling multiplier eps of -(dxx + dyy + eps dzz)")
    ap.add_argument("--semi", type=int, default=0, help="number k of leading x,y-only coarsenings (mg_desc.semi_xy); "
                    "BASELINE config 5: --aniso-eps 0.01 --semi 3 --levels 8 --smoother rbgs")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="launcher: seconds before the ranks are stopped")
    ap.add_argument("--rehearse", action="store_true",
                    help="plumbing check without a GPU: ranks rendezvous over gloo, plan their slabs, time nothing")
    ap.add_argument("--rehearse-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------ launcher
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(a, argv):
    """Parent of an N-rank run: one child per GPU, nothing here touches the GPU. Returns the exit code."""
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if a.rccl_same_gpu:
            env.update(NCCL_HOSTID=f"mg-rehearsal-rank{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                       NCCL_SHM_DISABLE="1")
        # rank 0's stdout carries the JSON line; everything else goes to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))
    deadline = time.monotonic() + a.launch_timeout
    failed = None
    # rank 0's pipe is drained by a thread from the start: a rank 0 that writes more than a pipe buffer before the other
    # ranks finish must not stall against a parent that only polls
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    out0 = ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed = f"ranks still running after {a.launch_timeout:.0f} s"
                break
            time.sleep(0.05)
        if failed is None:
            reader.join(timeout=10)
            out0 = "".join(c for c in chunks if c)
    finally:
        for p in procs:          # the exact children we started, nothing by pattern
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            while p.poll() is None and time.monotonic() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
    if failed:
        print(f"bench.py launcher: {failed}", file=sys.stderr)
        return 1
    line = [l for l in out0.splitlines() if l.startswith("{")]
    for l in out0.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if len(line) != 1:
        print(f"bench.py launcher: expected one JSON line from rank 0, got {len(line)}", file=sys.stderr)
        return 1
    print(line[0])
    return 0


# ------------------------------------------------------------------------------------ workload
def workload_desc(mod, a):
    return mod.make_desc(
        dim=3, n=a.n, levels=a.levels, dtype=mod.MG_F64 if a.dtype == "f64" else mod.MG_F32,
        length=1.0, alpha=1.0, cycle=mod.CYCLE_V,
        smoother={"jacobi": mod.SMOOTH_JACOBI, "rbgs": mod.SMOOTH_RBGS, "zebra": mod.SMOOTH_ZEBRA_Y, "zebrax": mod.SMOOTH_ZEBRA_X}[a.smoother],
        omega=6.0 / 7.0 if a.smoother == "jacobi" else 1.0, nu_pre=2, nu_post=2,
        restriction=mod.RESTRICT_FULLW,
        **({} if a.semi else {"coarse_mode": mod.COARSE_TOL, "coarse_maxit": 2000, "coarse_tol": 0.1}),
        outer_pre_gs=0, dist_min_n=a.dist_min_n, aniso=(a.aniso_x, a.aniso_y, a.aniso_eps), semi_xy=a.semi,
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


def rehearse(a, rank, world):
    """No GPU: rendezvous, the library's host-only slab plan, one all-reduce -- the launcher's and the
    rank bookkeeping's plumbing, exercised by tests/test_bench_launcher.py on CPU."""
    import torch
    import torch.distributed as dist
    from multigrid_prj_amd import capi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == a.rehearse_fail_rank:
        os._exit(3)
    desc = workload_desc(capi, a)
    z0, nz, fg = capi.plan_slab(desc, world, rank, 0)
    t = torch.tensor([float(nz), float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    planes = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(planes, torch.tensor([float(z0), float(nz)], dtype=torch.float64))
    if rank == 0:
        assert int(t[0].item()) == a.n, "slabs do not cover the grid"
        print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "first_gathered_level": fg, "slabs": [[int(p[0]), int(p[1])] for p in planes]}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    dry = a.dry_rank >= 0 and a.gpus > 1
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and a.gpus > 1 and not dry:
        sys.exit(launch(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if dry:
        rank, world, local_rank = a.dry_rank, a.gpus, 0
    if world != a.gpus:
        a.gpus = world   # under torch.distributed.run the environment is authoritative
    if world > 1:
        # a communication problem must end the run, not hang the node: every rank stops itself if the
        # whole benchmark has not finished in time
        import signal
        signal.alarm(int(a.launch_timeout))
    if a.rehearse:
        return rehearse(a, rank, world)

    from multigrid_prj_amd import capi

    dist = None
    comm_id = None
    host_comm = None
    device = local_rank
    if world > 1 and not dry:
        if a.rccl_same_gpu:   # also under torch.distributed.run, where nobody prepared the rank's environment
            os.environ.update(NCCL_HOSTID=f"mg-rehearsal-rank{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                              NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1")
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        if a.transport == "rccl":
            if a.rccl_same_gpu:
                device = local_rank = 0
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
    npdt = np.float64 if a.dtype == "f64" else np.float32
    dry_coarse = None
    if dry and not a.dry_raw_coarse and not a.semi:
        # k ranks == 1 rank bit for bit, so a real N-rank run makes exactly the single-GPU run's coarse sweeps: count them
        # (same cycles as the timed region) and cap the dry run's coarse solve there -- same kernel, same tests per sweep
        with capi.Solver(workload_desc(capi, a), device=device) as s1:
            s1.set_rhs(hash_rhs(a.n, npdt))
            s1.zero_array(capi.ARR_U, 0)
            counts = [s1.cycle().coarse_iters for _ in range(a.warmup + a.steps)]
        dry_coarse = {"single_gpu_sweeps_per_cycle": float(np.mean(counts[a.warmup:])), "capped_at": max(1, int(round(np.mean(counts[a.warmup:]))))}
        desc.coarse_maxit = dry_coarse["capped_at"]
    s = capi.Solver(desc, device=device, rank=rank, nranks=world, comm_id=comm_id, host_comm=host_comm, dry=dry)
    z0, nz, first_gathered = capi.plan_slab(desc, world, rank, 0)
    s.set_rhs(hash_rhs(a.n, npdt, z0, nz))
    s.zero_array(capi.ARR_U, 0)
    _, _, transport_ranks, transport_name = s.comm_info()

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
    # the same steady state as long ones; u is zeroed again afterwards. The same loop gives the cost of the
    # per-cycle residual norm mg_solve adds to a cycle (`value` counts bare cycles, mg_cycle_async).
    hist, _ = s.solve(0.0, 4)
    s.zero_array(capi.ARR_U, 0)
    barrier()
    # warmup (untimed)
    s.cycle_async(a.warmup)
    barrier()
    # timed region: exactly K cycles, finest-level launches bracketed by HIP events
    s.profile_begin()
    groups0, sent0 = s.comm_stats()
    t0 = time.perf_counter()
    s.cycle_async(a.steps)
    barrier()
    t1 = time.perf_counter()
    groups1, sent1 = s.comm_stats()
    _, sm_sweeps = s.profile_end()
    prof = {k: s.profile_get(getattr(capi, "PROF_" + k)) for k in ("SMOOTH", "SMOOTH_PROLONG", "RESID_RESTRICT", "PROLONG")}
    elapsed = t1 - t0
    per_rank_ms = [1e3 * elapsed / a.steps]
    if dist is not None:
        import torch
        dev = "cuda" if a.transport == "rccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [1e3 * float(x.item()) / a.steps for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # mg_solve's cycle = this cycle + one finest-grid residual norm: time a few of them for the record
    barrier()
    # (marginal cost of one more iteration of mg_solve's loop: the loop's fixed part -- sum b^2, the first norm and, where the
    # norm rides on the next cycle's first pair, the one speculative pair that is dropped at the end -- is not a per-cycle cost)
    def timed_solve(k):
        s.zero_array(capi.ARR_U, 0)
        barrier()
        t0 = time.perf_counter()
        s.solve(0.0, k)
        barrier()
        return time.perf_counter() - t0
    timed_solve(2)
    solve_ms_per_cycle = 1e3 * (timed_solve(14) - timed_solve(4)) / 10

    # single streaming sweeps of the finest grid (the unfused kernel the fused pair replaces)
    stream_ms = None
    if world == 1 and a.smoother == "jacobi":
        s.smooth(0, capi.SMOOTH_JACOBI, 1, capi.ARR_U, capi.ARR_RHS)
        s.timer_start()
        for _ in range(9):
            s.smooth(0, capi.SMOOTH_JACOBI, 1, capi.ARR_U, capi.ARR_RHS)
        stream_ms = s.timer_stop() / 9

    ms_per_step = 1e3 * elapsed / a.steps
    cycles_per_s = a.steps / elapsed
    esz = 8 if a.dtype == "f64" else 4
    pts_local = nz * a.n * a.n
    sweep_bytes = 3 * esz * pts_local                 # read u, read rhs, write u': what one pass must move
    sm_ms, sm_launches = prof["SMOOTH"]
    sweeps_timed = sm_sweeps                            # sweeps those launches performed
    launch_ms = sm_ms / max(sm_launches, 1)
    sweeps_per_launch = sweeps_timed / max(sm_launches, 1)
    # a launch that does less than a sweep (one colour pass of a zebra line smoother: half a sweep) is priced per FULL sweep:
    # 24 B/pt against the time of the launches that make one sweep
    bytes_per_launch = sweep_bytes * min(sweeps_per_launch, 1.0) if sm_launches else 0.0
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if sm_launches else 0.0
    fused_pair = sweeps_per_launch > 1.5
    vlen = 2 if a.dtype == "f64" else 4
    wide = world == 1 and (a.n - 1) % vlen == 0 and (a.n - 1) // vlen in (128, 256) and os.environ.get("MG_PAIR_WIDE", "1") != "0"
    pair_kernel = "k_pairw (wide tiles, mg_pair_wide.hip)" if wide else "k_jacobi2"

    traffic_rows = profiled_traffic(a) if world == 1 else {}
    tr = traffic_rows.get("pair" if fused_pair else "single")
    traffic = tr["bytes"] if tr else None
    # arrays beyond the 256 MB Infinity Cache stream from HBM: their compulsory bytes cannot move faster than the peak
    assert achieved <= HBM_PEAK_GBS or pts_local * esz <= (256 << 20), "frac > 1 on an HBM-resident level"

    def priced(kind, label, must_move, replaces, tkey):
        ms, n = prof[kind]
        if not n:
            return None
        lm = ms / n
        row = traffic_rows.get(tkey)
        return {"kernel": label, "launch_ms": lm, "launches_timed": n, "compulsory_bytes": int(must_move),
                "achieved": must_move / (lm * 1e-3) / 1e9, "frac": must_move / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "replaces_bytes": int(replaces), "equivalent_gbps": replaces / (lm * 1e-3) / 1e9,
                "traffic": row["bytes"] if row else None,
                "traffic_frac": row["bytes"] / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS if row else None}

    pts_c = ((a.n + 1) // 2) ** 2 * (a.n if a.semi else (a.n + 1) // 2)
    kernels = [k for k in (
        # J(J(u + P e)): reads u, the coarse correction (1/8 of the points) and rhs, writes once
        priced("SMOOTH_PROLONG", f"post-smoothing launch that also applies the coarse correction ({pair_kernel}, CORR)",
               sweep_bytes + esz * pts_c, 2 * sweep_bytes + 2 * esz * pts_local + esz * pts_c, "corr"),
        # R(rhs - A u): reads u and rhs, writes the coarse right-hand side
        priced("RESID_RESTRICT", f"finest-level residual + full-weighting restriction ({'k_rrw (wide tiles, mg_rr_wide.hip)' if wide else 'k_resid_restrict_fw'})",
               2 * esz * pts_local + esz * pts_c, 3 * esz * pts_local + esz * (pts_local + pts_c), "rr"),
        priced("PROLONG", "separate prolongation into the finest level", 2 * esz * pts_local + esz * pts_c,
               2 * esz * pts_local + esz * pts_c, "prolong"),
    ) if k]

    out = {
        "metric": f"V-cycles/sec (3D Poisson {a.n}^3 V(2,2)) + finest-grid smoother GB/s vs HBM roofline",
        "value": cycles_per_s, "unit": "V-cycles/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"3D Poisson {a.n}^3 nodes (nominal {a.n - 1}^3), {a.levels}-level V(2,2), "
                               f"{a.smoother}{' omega=6/7' if a.smoother == 'jacobi' else ''}, full-weighting, "
                               f"{f'eps={a.aniso_eps}, first {a.semi} coarsenings in x,y only, ' if a.semi or a.aniso_eps != 1.0 else ''}"
                               f"{f'y-coupling x{a.aniso_y}, ' if a.aniso_y != 1.0 else ''}"
                               f"{f'x-coupling x{a.aniso_x}, ' if a.aniso_x != 1.0 else ''}"
                               + (f"20 coarse sweeps, {a.dtype}" if a.semi else
                                  f"coarse {((a.n - 1) >> (a.levels - 1)) + 1}^3 iterated to rel. residual 0.1, {a.dtype}"),
                   "parallelism": (f"z-slab x{world}" + ("" if a.transport == "rccl" else " (gloo rehearsal)") + (" (RCCL rehearsal: all ranks on GPU 0, socket transport)" if a.rccl_same_gpu else "")) if world > 1 else "single GPU",
                   "first_gathered_level": first_gathered},
        **({"dry_run": f"rank {rank} of {world} WITHOUT communication: one rank's compute schedule, not a result",
            "dry_run_coarse_sweeps": dry_coarse} if dry else {}),
        "transport": transport_name, "rccl_ranks": transport_ranks if transport_name == "rccl" else None,
        "ms_per_step_ranks": per_rank_ms,
        # what rank 0 posts per cycle: message groups (halo exchanges, gather, scatter: one ncclGroup each) and bytes sent
        "comm_per_cycle": ({"message_groups": (groups1 - groups0) / a.steps, "bytes_sent": (sent1 - sent0) / a.steps}
                           if world > 1 else None),
        "ms_per_step_with_residual_norm": solve_ms_per_cycle,
        "roofline": {"bound": "hbm",
                     "kernel": ((f"finest-grid fused double {'red-black half-' if a.smoother == 'rbgs' else 'Jacobi '}sweep {pair_kernel} ({a.n}^3, one pass over HBM)" if fused_pair
                                 else f"finest-grid {a.smoother} sweep ({a.n}^3)")
                                + (f"; rank 0's z-slab of {nz} planes, one segment = halo exchange + interior + boundary launches" if world > 1 else "")),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_frac": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "traffic_source": tr["source"] if tr else None,
                     "launch_ms": launch_ms, "launches_timed": sm_launches, "sweeps_per_launch": sweeps_per_launch,
                     "compulsory_bytes_per_launch": int(bytes_per_launch),
                     "sweep_ms": launch_ms / sweeps_per_launch,
                     "sweep_equivalent_gbps": sweeps_per_launch * sweep_bytes / (launch_ms * 1e-3) / 1e9,
                     "note": ("achieved = 24 B/pt (read u, read rhs, write once) / launch time: the launch makes one pass for "
                              "two sweeps; sweep_equivalent_gbps = what two streaming sweeps would have moved per second "
                              "of it" if fused_pair else
                              ("a launch is one colour pass = half a sweep: achieved = 24 B/pt / (2 x launch time), the full sweep's rate"
                               if sweeps_per_launch < 1 else None))},
        "smoother_gbps": achieved,
        "kernels": kernels,
        "streaming_sweep": ({"kernel": f"single Jacobi sweep k_sweep3d ({a.n}^3), timed after the region", "launch_ms": stream_ms,
                             "achieved": sweep_bytes / (stream_ms * 1e-3) / 1e9,
                             "frac": sweep_bytes / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": traffic_rows["single"]["bytes"] if "single" in traffic_rows else None}
                            if stream_ms else None),
        "residual_drop_per_cycle": float(hist[-1] / hist[-2]) if len(hist) >= 2 and hist[-2] > 0 else None,
        "device_bytes": s.device_bytes(),
    }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
    if rank == 0 or dry:
        print(json.dumps(out))
    s.close()
    if dist is not None:
        dist.destroy_process_group()


def _template_args(name):
    i, j = name.find("<"), name.rfind(">")
    return [x.strip() for x in name[i + 1:j].split(",")] if 0 <= i < j else []


def profiled_traffic(a):
    """HBM-side bytes per launch of the finest-level kernels from the committed rocprofv3 PMC passes of
    this same command (profiles/r*_kernel_summary.csv, made by tools/profile.sh + tools/summarize_prof.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, read = 2 x FETCH_SIZE x 1024 on gfx950). Counters cannot
    be collected inside a timed run, so this is not live; rows are picked by exact template signature and
    the finest grid's launch size. Empty when the workload is not the profiled default."""
    import csv
    import glob
    import re
    # which committed summaries belong to this workload: the default command's are profiles/r<NN>[x]_[unfused_]kernel_summary.csv,
    # the other configurations' carry a tag (tools/profile.sh <tag> <bench args>)
    default = a.n == 513 and a.dtype == "f64" and a.smoother == "jacobi" and a.levels == 6 and not a.semi \
        and a.aniso_eps == 1.0 and a.aniso_x == 1.0 and a.aniso_y == 1.0
    if default:
        pat = re.compile(r"r\d+[a-z]?_(unfused_)?kernel_summary\.csv$")
    elif a.n == 513 and a.dtype == "f64" and a.smoother == "rbgs" and a.levels == 6 and not a.semi:
        pat = re.compile(r"r\d+[a-z]?_config3_rbgs_kernel_summary\.csv$")
    elif a.n == 1025 and a.dtype == "f32" and a.smoother == "jacobi" and a.levels == 7:
        pat = re.compile(r"r\d+[a-z]?_config4_grid_kernel_summary\.csv$")
    elif a.n == 513 and a.dtype == "f64" and a.smoother == "zebra" and a.semi == 3:
        pat = re.compile(r"r\d+[a-z]?_config5_semi_zebra_kernel_summary\.csv$")
    else:
        return {}
    T = "double" if a.dtype == "f64" else "float"
    lanes = str((a.n - 1) // (2 if a.dtype == "f64" else 4))
    rows = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_summary.csv"))):
        if not pat.search(os.path.basename(path)):
            continue
        src = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc passes of this command)"
        best = {}
        for r in csv.DictReader(open(path)):
            if not (r["read_MB"] and r["write_MB"]):
                continue
            t = _template_args(r["kernel"])
            key = None
            if r["kernel"].startswith("k_pairw<") and t[:2] == [T, lanes]:
                # <T, TPR, G, DAMPED, CORR, ZEROU, RB[, NORM]>
                if t[4:6] == ["false", "false"]:
                    key = "pair"
                elif t[4:6] == ["true", "false"]:
                    key = "corr"
            elif r["kernel"].startswith("k_jacobi2<") and t[:2] == [T, lanes]:
                # <T, TPR, DAMPED, NTLOAD, CORR, RB, ZEROU, TYO>
                if t[4] == "false" and t[6] == "false":
                    key = "pair"
                elif t[4] == "true" and t[6] == "false":
                    key = "corr"
            elif r["kernel"].startswith("k_zebra_") and t[:1] == [T]:
                key = "pair" if "pair" not in best or int(r["grid_threads"]) > int(best["pair"]["grid_threads"]) else None
            elif r["kernel"].startswith("k_sweep3d<") and t[:2] == [T, "0"] and t[-1] == "false":
                key = "single"
            elif r["kernel"].startswith("k_rrw<") and t[:2] == [T, lanes]:      # <T, TPR, G, SEMI>: the wide-tile residual + restriction
                key = "rr"
            elif r["kernel"].startswith("k_resid_restrict_fw<") and t[:1] == [T]:
                key = "rr"
            elif r["kernel"].startswith("k_prolong3d_fast<") and t[:1] == [T]:
                key = "prolong"
            if key == "rr" and "rr" in best and best["rr"]["kernel"].startswith("k_rrw<") != r["kernel"].startswith("k_rrw<"):
                if r["kernel"].startswith("k_rrw<"):
                    best["rr"] = r        # the wide-tile kernel is the finest level's; the other one only runs below it
                continue
            if key and (key not in best or int(r["grid_threads"]) > int(best[key]["grid_threads"])):
                best[key] = r
        for key, r in best.items():   # later rounds' files win
            if key == "single" and default and int(r["grid_threads"]) < 10_000_000:
                continue              # only a finest-grid launch (11.4 M threads) prices the streaming sweep
            rows[key] = {"bytes": (float(r["read_MB"]) + float(r["write_MB"])) * 1e6, "source": src, "kernel": r["kernel"]}
    return rows


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
