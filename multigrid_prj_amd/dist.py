"""Host-side transport for the slab-decomposed solver over torch.distributed (gloo).

TEST TRANSPORT: libmg_hip's product path exchanges halo planes GPU-to-GPU with RCCL
(mg_create_distributed). This module wires the library's host-callback variant
(mg_create_distributed_hostcomm) to torch.distributed so that several processes sharing ONE
GPU -- which RCCL refuses -- can exercise the whole multi-rank code path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def torch_host_comm(group=None) -> capi.MgHostComm:
    import torch
    import torch.distributed as dist

    def _tensor(ptr, nbytes):
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.uint8)

    def batch(ctx, ops, nops):
        try:
            reqs = []
            for k in range(nops):
                op = ops[k]
                t = _tensor(op.buf, op.bytes)
                reqs.append(dist.isend(t, op.peer, group=group) if op.is_send else dist.irecv(t, op.peer, group=group))
            for r in reqs:
                r.wait()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("host comm batch failed:", e, flush=True)
            return 1

    def allreduce(ctx, vals, n):
        try:
            arr = np.ctypeslib.as_array(vals, shape=(n,))
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:
            print("host comm allreduce failed:", e, flush=True)
            return 1

    hc = capi.MgHostComm()
    hc.ctx = None
    hc.batch = capi.BATCH_FN(batch)
    hc.allreduce_sum = capi.ALLREDUCE_FN(allreduce)
    hc._keep = (batch, allreduce)
    return hc
