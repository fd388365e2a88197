// mg_comm.h -- point-to-point transport used by the slab-decomposed solver.
// Two backends (mg_dist.cpp): RcclComm (ncclSend/ncclRecv over xGMI, the product path) and
// HostComm (stages through pinned host memory and calls user callbacks; test transport).
#ifndef MG_COMM_H
#define MG_COMM_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

#include "../../include/mg_hip.h"

namespace mg {

struct P2POp {
    int peer;
    bool send;
    void *dptr;  // device pointer
    size_t bytes;
};

class Comm {
public:
    virtual ~Comm() = default;
    // posts all ops as one group on stream `s` (no host synchronisation for RCCL)
    virtual int batch(const P2POp *ops, int n, hipStream_t s) = 0;
    // in-place sum of n device doubles over all ranks, on stream `s`
    virtual int allreduce_sum(double *dptr, int n, hipStream_t s) = 0;
    // ranks the transport itself reports (RCCL: ncclCommCount), for mg_comm_info
    virtual int transport_ranks() const { return nranks; }
    virtual const char *name() const = 0;
    int rank = 0, nranks = 1;
};

// returns nullptr and sets *why on failure
Comm *make_rccl_comm(int rank, int nranks, const void *id128, std::string *why);
Comm *make_host_comm(int rank, int nranks, const mg_host_comm &cb, std::string *why);
Comm *make_dry_comm(int rank, int nranks);  // moves nothing: timing of one rank's schedule only
int rccl_unique_id(void *id128, std::string *why);

}  // namespace mg
#endif
