// mg_dist.cpp -- transports of the slab-decomposed solver (see mg_comm.h).
// RCCL: one communicator per handle, halo planes move GPU-to-GPU over xGMI with grouped
// ncclSend/ncclRecv on the solver's own stream (stream-ordered with the kernels, no host
// round trip); scalars (sum r^2, sum b^2) with ncclAllReduce on one double.
#include "mg_comm.h"

#include <cstring>
#include <vector>

#ifdef MG_WITH_RCCL
#include <rccl/rccl.h>
#endif

namespace mg {

#ifdef MG_WITH_RCCL
namespace {
class RcclComm : public Comm {
public:
    ncclComm_t comm = nullptr;
    ~RcclComm() override { if (comm) ncclCommDestroy(comm); }
    int transport_ranks() const override
    {
        int n = 0;
        return (comm && ncclCommCount(comm, &n) == ncclSuccess) ? n : -1;
    }
    const char *name() const override { return "rccl"; }
    int batch(const P2POp *ops, int n, hipStream_t s) override
    {
        if (n == 0) return MG_OK;
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            if (ops[i].send) r = ncclSend(ops[i].dptr, ops[i].bytes, ncclInt8, ops[i].peer, comm, s);
            else r = ncclRecv(ops[i].dptr, ops[i].bytes, ncclInt8, ops[i].peer, comm, s);
        }
        ncclResult_t e = ncclGroupEnd();
        return (r == ncclSuccess && e == ncclSuccess) ? MG_OK : MG_ERR_COMM;
    }
    int allreduce_sum(double *d, int n, hipStream_t s) override
    {
        return ncclAllReduce(d, d, (size_t)n, ncclDouble, ncclSum, comm, s) == ncclSuccess ? MG_OK : MG_ERR_COMM;
    }
};
}  // namespace

int rccl_unique_id(void *id128, std::string *why)
{
    static_assert(sizeof(ncclUniqueId) <= MG_COMM_ID_BYTES, "unique id does not fit");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) { if (why) *why = "ncclGetUniqueId failed"; return MG_ERR_COMM; }
    std::memset(id128, 0, MG_COMM_ID_BYTES);
    std::memcpy(id128, &id, sizeof(id));
    return MG_OK;
}

Comm *make_rccl_comm(int rank, int nranks, const void *id128, std::string *why)
{
    if (!id128) { if (why) *why = "null RCCL unique id"; return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RcclComm *c = new RcclComm();
    c->rank = rank; c->nranks = nranks;
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        if (why) *why = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        c->comm = nullptr;
        delete c;
        return nullptr;
    }
    return c;
}
#else
int rccl_unique_id(void *, std::string *why) { if (why) *why = "built without RCCL"; return MG_ERR_COMM; }
Comm *make_rccl_comm(int, int, const void *, std::string *why) { if (why) *why = "built without RCCL"; return nullptr; }
#endif

namespace {
class HostComm : public Comm {
public:
    mg_host_comm cb{};
    std::vector<char *> stage;  // pinned staging buffers, one per op slot
    std::vector<size_t> cap;
    double *hscal = nullptr;
    const char *name() const override { return "host-callbacks"; }
    ~HostComm() override
    {
        for (char *p : stage) if (p) (void)hipHostFree(p);
        if (hscal) (void)hipHostFree(hscal);
    }
    char *slot(size_t i, size_t bytes)
    {
        if (stage.size() <= i) { stage.resize(i + 1, nullptr); cap.resize(i + 1, 0); }
        if (cap[i] < bytes) {
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (hipHostMalloc((void **)&stage[i], bytes) != hipSuccess) return nullptr;
            cap[i] = bytes;
        }
        return stage[i];
    }
    int batch(const P2POp *ops, int n, hipStream_t s) override
    {
        if (n == 0) return MG_OK;
        std::vector<mg_p2p_op> hops((size_t)n);
        for (int i = 0; i < n; i++) {
            char *h = slot((size_t)i, ops[i].bytes);
            if (!h) return MG_ERR_HIP;
            hops[i] = mg_p2p_op{ops[i].peer, ops[i].send ? 1 : 0, h, ops[i].bytes};
            if (ops[i].send && hipMemcpyAsync(h, ops[i].dptr, ops[i].bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return MG_ERR_HIP;
        }
        if (hipStreamSynchronize(s) != hipSuccess) return MG_ERR_HIP;
        if (cb.batch(cb.ctx, hops.data(), n) != 0) return MG_ERR_COMM;
        for (int i = 0; i < n; i++)
            if (!ops[i].send && hipMemcpyAsync(ops[i].dptr, hops[i].buf, ops[i].bytes, hipMemcpyHostToDevice, s) != hipSuccess) return MG_ERR_HIP;
        // the staging slots are reused by the next batch: wait for the uploads
        if (hipStreamSynchronize(s) != hipSuccess) return MG_ERR_HIP;
        return MG_OK;
    }
    int allreduce_sum(double *d, int n, hipStream_t s) override
    {
        if (!hscal && hipHostMalloc((void **)&hscal, sizeof(double) * 16) != hipSuccess) return MG_ERR_HIP;
        if (n > 16) return MG_ERR_BAD_ARG;
        if (hipMemcpyAsync(hscal, d, sizeof(double) * n, hipMemcpyDeviceToHost, s) != hipSuccess) return MG_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return MG_ERR_HIP;
        if (cb.allreduce_sum(cb.ctx, hscal, n) != 0) return MG_ERR_COMM;
        if (hipMemcpyAsync(d, hscal, sizeof(double) * n, hipMemcpyHostToDevice, s) != hipSuccess) return MG_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return MG_ERR_HIP;
        return MG_OK;
    }
};
}  // namespace

// Dry run: rank `rank` of `nranks` with nobody on the other side. Nothing moves (ghost planes keep whatever they hold,
// sums stay local), so the numbers a handle on it produces mean nothing; what it gives is ONE rank's launch schedule --
// slab kernels, boundary launches, the replicated coarse levels -- timed on a single GPU (bench.py --transport dry).
namespace {
class DryComm : public Comm {
public:
    const char *name() const override { return "dry-run"; }
    int transport_ranks() const override { return 1; }
    int batch(const P2POp *, int, hipStream_t) override { return MG_OK; }
    int allreduce_sum(double *, int, hipStream_t) override { return MG_OK; }
};
}  // namespace

Comm *make_dry_comm(int rank, int nranks)
{
    DryComm *c = new DryComm();
    c->rank = rank; c->nranks = nranks;
    return c;
}

Comm *make_host_comm(int rank, int nranks, const mg_host_comm &cb, std::string *why)
{
    if (!cb.batch || !cb.allreduce_sum) { if (why) *why = "host comm callbacks missing"; return nullptr; }
    HostComm *c = new HostComm();
    c->rank = rank; c->nranks = nranks; c->cb = cb;
    return c;
}

}  // namespace mg
