"""`python3 bench.py --gpus N` as the driver types it: without RANK/WORLD_SIZE in the environment the process
only launches one child per rank and relays rank 0's JSON line. CPU rehearsal (--rehearse: gloo rendezvous,
the library's host-only slab plan, no GPU) of exactly that plumbing, world size 2 and 3."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_self_launches_its_ranks(world):
    p = _run(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--rehearse", "--grid", "129", "--levels", "4", "--dist-min-n", "33"])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.strip() == lines[0]      # ONE JSON line on stdout, nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1
    slabs = out["slabs"]
    assert slabs[0][0] == 0 and sum(nz for _, nz in slabs) == 129
    assert all(slabs[r][0] == slabs[r - 1][0] + slabs[r - 1][1] for r in range(1, world))


def test_bench_launcher_fails_when_a_rank_fails():
    p = _run(["--gpus", "2", "--rehearse", "--grid", "65", "--levels", "3", "--rehearse-fail-rank", "1", "--launch-timeout", "60"])
    assert p.returncode != 0 and "rank 1 exited with code 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_launcher_deadline():
    """A rank that never finishes (here: rank 1 waits for a rendezvous rank 0 never joins) ends the run."""
    p = _run(["--gpus", "2", "--rehearse", "--grid", "65", "--levels", "3", "--rehearse-fail-rank", "0", "--launch-timeout", "20"])
    assert p.returncode != 0


@pytest.mark.gpu
def test_bench_dry_rank_runs_one_ranks_schedule_on_one_gpu():
    """--dry-rank R: one process plays rank R of N with no peers (measurement tool). The line must say so, and a middle
    rank's schedule must post the same message groups per cycle as the real transports do (nothing moves)."""
    p = _run(["--gpus", "4", "--dry-rank", "1", "--steps", "3", "--warmup", "1", "--grid", "129", "--levels", "4", "--dist-min-n", "65",
              "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert "dry_run" in out and "WITHOUT communication" in out["dry_run"]
    assert out["dry_run_coarse_sweeps"]["capped_at"] >= 1     # the coarse solve is capped at the real problem's sweep count
    assert out["transport"] == "dry-run" and out["rccl_ranks"] is None and out["n_gpus"] == 4
    assert out["comm_per_cycle"]["message_groups"] > 0 and out["ms_per_step"] > 0
    # every finest-level row is priced per SEGMENT (exchange + interior + boundary launches count as one): one per cycle timed
    rr = [k for k in out["kernels"] if "residual" in k["kernel"]]
    assert rr and all(k["launches_timed"] == 3 for k in rr), out["kernels"]


@pytest.mark.gpu
def test_bench_launches_real_rccl_ranks_on_one_gpu():
    """`python3 bench.py --gpus 2` as the driver types it, with RCCL as the transport: --rccl-same-gpu puts both ranks on GPU 0
    and gives each a host id of its own, so RCCL accepts them and moves the halos over its socket transport. The line must come
    from two RCCL ranks and show the cycle converging like the single-GPU one."""
    p = _run(["--gpus", "2", "--rccl-same-gpu", "--steps", "3", "--warmup", "1", "--grid", "257", "--levels", "5", "--no-cpu-baseline",
              "--launch-timeout", "240"], timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["transport"] == "rccl" and out["rccl_ranks"] == 2 and out["n_gpus"] == 2
    assert len(out["ms_per_step_ranks"]) == 2 and out["comm_per_cycle"]["message_groups"] > 0
    assert 0.15 < out["residual_drop_per_cycle"] < 0.30      # 0.226 on one GPU
