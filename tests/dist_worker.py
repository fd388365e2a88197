"""One rank of a multi-process run, started by tests/test_distributed.py.

    python tests/dist_worker.py <mode> <rank> <world> <port> <case-json> <outdir>

mode "model": CPU only. A numpy model of the slab-decomposed V-cycle -- the partition plan
  comes from libmg_hip's host-only mg_plan_slab, halos / gather / scatter / norm all-reduce
  go through torch.distributed (gloo). It checks that plan + communication schedule
  reproduce the single-rank oracle bit for bit.
mode "hip": the real distributed solver of libmg_hip (mg_create_distributed_hostcomm) with
  its exchanges carried by gloo, all ranks sharing GPU 0.
mode "rccl": the PRODUCT transport. mg_create_distributed with an RCCL communicator of `world` ranks that all sit on
  GPU 0: the test runner gives every rank its own NCCL_HOSTID, so RCCL takes them for ranks on different hosts, accepts
  the shared device and moves the messages over its socket transport on the loopback interface. Everything above the
  wire is the code an 8-GPU node runs: grouped ncclSend/ncclRecv on the main and the communication stream, the
  all-gather of the coarse right-hand side, ncclAllReduce of the norms.
Rank r writes its slab of the solution and its residual history to <outdir>/rank<r>.npz.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    mode, rank, world, port, case_json, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    case = json.loads(case_json)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multigrid_prj_amd import capi
    kw = dict(case["desc"])
    desc = capi.make_desc(**kw)
    n = kw["n"]
    z0, nz, fg = capi.plan_slab(desc, world, rank, 0)
    b = np.load(case["rhs"])[z0:z0 + nz]
    cycles = case["cycles"]
    if mode in ("hip", "rccl"):
        if mode == "rccl":
            ids = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)      # over gloo: the 128 bytes of the RCCL unique id
            s = capi.Solver(desc, device=0, rank=rank, nranks=world, comm_id=ids[0])
            assert s.comm_info()[2:] == (world, "rccl"), s.comm_info()
        else:
            from multigrid_prj_amd.dist import torch_host_comm
            s = capi.Solver(desc, device=0, rank=rank, nranks=world, host_comm=torch_host_comm())
        s.set_rhs(b)
        s.cycle()                      # the first cycle also sends the right-hand side's ghost planes once
        g0, b0 = s.comm_stats()
        s.profile_begin()              # which kind of finest-level launches the remaining cycles make
        for _ in range(cycles - 1):
            s.cycle()
        g1, b1 = s.comm_stats()
        s.profile_end()
        kinds = {k: s.profile_get(getattr(capi, "PROF_" + k))[1] for k in ("SMOOTH", "SMOOTH_PROLONG", "PROLONG")}
        hist, _ = s.solve(0.0, 2)
        u = s.get_solution()
        extra = dict(groups_per_cycle=(g1 - g0) / max(cycles - 1, 1), bytes_per_cycle=(b1 - b0) / max(cycles - 1, 1),
                     fold_launches=kinds["SMOOTH_PROLONG"], prolong_launches=kinds["PROLONG"], smooth_launches=kinds["SMOOTH"])
        if case.get("check_e"):
            # the public mg_smooth on a distributed level must leave the caller's E array alone (inside the
            # V-cycle the fused pair uses E as scratch; the API call may not) and still equal the fused result
            rng = np.random.default_rng(100 + rank)
            mark = rng.standard_normal(s.level_shape(0)).astype(s.np)
            s.set_array(capi.ARR_E, 0, mark)
            s.smooth(0, kw["smoother"], 2, capi.ARR_U, capi.ARR_RHS)
            assert np.array_equal(s.get_array(capi.ARR_E, 0), mark), "mg_smooth overwrote the E array"
            u = s.get_solution()
        s.close()
    else:
        extra = {}
        from tests.slab_model import SlabVCycle
        m = SlabVCycle(desc, rank, world, dist)
        m.set_rhs(b)
        for _ in range(cycles):
            m.cycle()
        hist = m.solve_hist(2)
        u = m.solution()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), u=u, hist=np.asarray(hist), z0=z0, nz=nz, fg=fg, **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
