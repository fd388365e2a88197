// mg_capi.cpp -- extern "C" entry points declared in include/mg_hip.h.
// No exception crosses this boundary; every failure becomes a negative mg_status
// plus a message kept per host thread (mg_last_error).
#include <cstring>
#include <new>
#include <string>

#include "mg_solver.h"

struct mg_solver {
    mg::Solver *impl;
};

namespace {
int bad(const char *msg)
{
    mg::set_last_error(msg);
    return MG_ERR_BAD_ARG;
}
template <typename F>
int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        mg::set_last_error("host allocation failed");
        return MG_ERR_HIP;
    } catch (const std::exception &e) {
        mg::set_last_error(e.what());
        return MG_ERR_HIP;
    } catch (...) {
        mg::set_last_error("unknown C++ exception");
        return MG_ERR_HIP;
    }
}
}  // namespace

extern "C" {

const char *mg_last_error(void) { return mg::last_error().c_str(); }

int mg_device_count(int *count)
{
    if (!count) return bad("mg_device_count: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return MG_OK;
}

int mg_create(const mg_desc *desc, int device, mg_handle *out)
{
    return guarded([&]() -> int {
        if (!out) return bad("mg_create: null output handle");
        *out = nullptr;
        std::string why;
        int rc = mg::validate_desc(desc, &why);
        if (rc) { mg::set_last_error("mg_create: " + why); return rc; }
        mg::Solver *s = new mg::Solver(*desc, device);
        rc = s->init();
        if (rc) { delete s; return rc; }
        *out = new mg_solver{s};
        return MG_OK;
    });
}

int mg_destroy(mg_handle h)
{
    if (!h) return MG_OK;
    delete h->impl;
    delete h;
    return MG_OK;
}

#define MG_H(h) do { if (!(h) || !(h)->impl) return bad("null handle"); } while (0)

int mg_level_n(mg_handle h, int level, int *n)
{
    MG_H(h);
    if (!n || level < 0 || level >= h->impl->nlevels()) return bad("mg_level_n: bad argument");
    *n = h->impl->level(level).g.nx;
    return MG_OK;
}

int mg_level_nz(mg_handle h, int level, int *nz)
{
    MG_H(h);
    if (!nz || level < 0 || level >= h->impl->nlevels()) return bad("mg_level_nz: bad argument");
    *nz = h->impl->level(level).present ? h->impl->level(level).g.nz : 0;
    return MG_OK;
}

int mg_level_coefficients(mg_handle h, int level, double out[4])
{
    MG_H(h);
    if (!out || level < 0 || level >= h->impl->nlevels()) return bad("mg_level_coefficients: bad argument");
    std::memcpy(out, h->impl->level(level).coef, 4 * sizeof(double));
    return MG_OK;
}

int mg_set_rhs(mg_handle h, const void *b) { MG_H(h); return guarded([&] { return h->impl->set_array(MG_ARR_RHS, 0, b); }); }
int mg_set_solution(mg_handle h, const void *u) { MG_H(h); return guarded([&] { return h->impl->set_array(MG_ARR_U, 0, u); }); }
int mg_get_solution(mg_handle h, void *u) { MG_H(h); return guarded([&] { return h->impl->get_array(MG_ARR_U, 0, u); }); }
int mg_set_array(mg_handle h, int which, int level, const void *host) { MG_H(h); return guarded([&] { return h->impl->set_array(which, level, host); }); }
int mg_get_array(mg_handle h, int which, int level, void *host) { MG_H(h); return guarded([&] { return h->impl->get_array(which, level, host); }); }
int mg_zero_array(mg_handle h, int which, int level) { MG_H(h); return guarded([&] { return h->impl->zero_array(which, level); }); }

int mg_smooth(mg_handle h, int level, int smoother, int sweeps, int arr_x, int arr_rhs)
{
    MG_H(h);
    return guarded([&] { return h->impl->smooth(level, smoother, sweeps, arr_x, arr_rhs); });
}
int mg_residual(mg_handle h, int level, int arr_x, int arr_rhs, int arr_r, double *sumsq_r)
{
    MG_H(h);
    return guarded([&] { return h->impl->residual(level, arr_x, arr_rhs, arr_r, sumsq_r); });
}
int mg_sumsq(mg_handle h, int level, int arr, double *sumsq)
{
    MG_H(h);
    return guarded([&] { return h->impl->sumsq(level, arr, sumsq); });
}
int mg_restrict(mg_handle h, int fine_level, int kind, int arr_src, int arr_dst)
{
    MG_H(h);
    return guarded([&] { return h->impl->restrict_to(fine_level, kind, arr_src, arr_dst); });
}
int mg_prolong(mg_handle h, int coarse_level, int add, int arr_src, int arr_dst)
{
    MG_H(h);
    return guarded([&] { return h->impl->prolong(coarse_level, add, arr_src, arr_dst); });
}
int mg_correct(mg_handle h, int arr_u, int arr_e)
{
    MG_H(h);
    return guarded([&] { return h->impl->correct(arr_u, arr_e); });
}
int mg_coarse_solve(mg_handle h, int level, int arr_x, int arr_rhs, mg_cycle_stats *st)
{
    MG_H(h);
    return guarded([&] { return h->impl->coarse_solve(level, arr_x, arr_rhs, st); });
}
int mg_coarse_solve_ex(mg_handle h, int level, int arr_x, int arr_rhs, int smoother, int maxit,
                       double tol, int fixed, mg_cycle_stats *st)
{
    MG_H(h);
    return guarded([&] { return h->impl->coarse_solve_ex(level, arr_x, arr_rhs, smoother, maxit, tol, fixed, st); });
}
int mg_cycle(mg_handle h, mg_cycle_stats *st) { MG_H(h); return guarded([&] { return h->impl->cycle(st); }); }
int mg_cycle_async(mg_handle h, int count) { MG_H(h); return guarded([&] { return h->impl->cycle_async(count); }); }
int mg_solve(mg_handle h, double tol, int maxit, double *hist, int hist_cap, int *n_hist,
             mg_cycle_stats *per_cycle)
{
    MG_H(h);
    if (maxit < 0) return bad("mg_solve: negative maxit");
    return guarded([&] { return h->impl->solve(tol, maxit, hist, hist_cap, n_hist, per_cycle); });
}
int mg_solve_lockstep(mg_handle h, double tol, int maxit, const int *coarse_counts, int n_counts,
                      double *hist, int hist_cap, int *n_hist, mg_cycle_stats *per_cycle)
{
    MG_H(h);
    if (maxit < 0 || n_counts < 0 || (n_counts > 0 && !coarse_counts)) return bad("mg_solve_lockstep: bad argument");
    for (int i = 0; i < n_counts; i++)
        if (coarse_counts[i] < 0) return bad("mg_solve_lockstep: negative sweep count");
    return guarded([&] { return h->impl->solve(tol, maxit, hist, hist_cap, n_hist, per_cycle, coarse_counts, n_counts); });
}
int mg_set_stage_callback(mg_handle h, mg_stage_fn fn, void *user)
{
    MG_H(h);
    return guarded([&] { return h->impl->set_stage_callback(fn, user); });
}
int mg_sync(mg_handle h) { MG_H(h); return guarded([&] { return h->impl->sync(); }); }
int mg_timer_start(mg_handle h) { MG_H(h); return guarded([&] { return h->impl->timer_start(); }); }
int mg_timer_stop(mg_handle h, double *ms) { MG_H(h); return guarded([&] { return h->impl->timer_stop(ms); }); }
int mg_profile_begin(mg_handle h) { MG_H(h); return guarded([&] { return h->impl->profile_begin(); }); }
int mg_profile_end(mg_handle h, double *ms, int *sweeps) { MG_H(h); return guarded([&] { return h->impl->profile_end(ms, sweeps); }); }
int mg_profile_fused(mg_handle h, double *ms, int *sweeps) { MG_H(h); return guarded([&] { return h->impl->profile_fused(ms, sweeps); }); }
int mg_profile_get(mg_handle h, int kind, double *ms, int *launches)
{
    MG_H(h);
    if (kind < 0 || kind >= MG_PROF_KINDS) return bad("mg_profile_get: unknown kind");
    return guarded([&] { return h->impl->profile_get(kind, ms, launches); });
}
int mg_comm_info(mg_handle h, int *rank, int *nranks, int *transport_ranks, const char **transport)
{
    MG_H(h);
    return guarded([&] { return h->impl->comm_info(rank, nranks, transport_ranks, transport); });
}
int mg_comm_stats(mg_handle h, long long *groups, long long *bytes_sent)
{
    MG_H(h);
    if (groups) *groups = h->impl->comm_groups();
    if (bytes_sent) *bytes_sent = h->impl->comm_bytes_sent();
    return MG_OK;
}
int mg_device_bytes(mg_handle h, size_t *bytes)
{
    MG_H(h);
    if (!bytes) return bad("mg_device_bytes: null argument");
    *bytes = h->impl->device_bytes();
    return MG_OK;
}

int mg_plan_slab(const mg_desc *desc, int nranks, int rank, int level, int *z0, int *nz,
                 int *first_gathered_level)
{
    std::string why;
    int rc = mg::validate_desc(desc, &why);
    if (rc) { mg::set_last_error("mg_plan_slab: " + why); return rc; }
    mg::SlabPlan p;
    rc = mg::plan_slab(*desc, nranks, rank, level, &p, &why);
    if (rc) { mg::set_last_error("mg_plan_slab: " + why); return rc; }
    if (z0) *z0 = p.z0;
    if (nz) *nz = p.nz;
    if (first_gathered_level) *first_gathered_level = p.first_gathered_level;
    return MG_OK;
}

int mg_comm_unique_id(void *id128)
{
    if (!id128) return bad("mg_comm_unique_id: null argument");
    std::string why;
    int rc = mg::rccl_unique_id(id128, &why);
    if (rc) mg::set_last_error("mg_comm_unique_id: " + why);
    return rc;
}

int mg_comm_selftest(size_t bytes)
{
    return guarded([&]() -> int {
        char id[MG_COMM_ID_BYTES];
        std::string why;
        int rc = mg::rccl_unique_id(id, &why);
        if (rc) { mg::set_last_error("mg_comm_selftest: " + why); return rc; }
        mg::Comm *c = mg::make_rccl_comm(0, 1, id, &why);
        if (!c) { mg::set_last_error("mg_comm_selftest: " + why); return MG_ERR_COMM; }
        struct Guard { mg::Comm *c; ~Guard() { delete c; } } g{c};
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return bad("stream");
        char *a = nullptr, *b = nullptr;
        double *d = nullptr;
        if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess ||
            hipMalloc((void **)&d, sizeof(double)) != hipSuccess) return bad("alloc");
        (void)hipMemsetAsync(a, 0x5a, bytes, s);
        (void)hipMemsetAsync(b, 0, bytes, s);
        double one = 1.25;
        (void)hipMemcpyAsync(d, &one, sizeof(double), hipMemcpyHostToDevice, s);
        mg::P2POp ops[2] = {{0, true, a, bytes}, {0, false, b, bytes}};
        rc = c->batch(ops, 2, s);
        if (!rc) rc = c->allreduce_sum(d, 1, s);
        std::string hb(bytes, 0);
        double out = 0;
        (void)hipMemcpyAsync(&hb[0], b, bytes, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(&out, d, sizeof(double), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(d); (void)hipStreamDestroy(s);
        if (rc) { mg::set_last_error("mg_comm_selftest: transport call failed"); return rc; }
        for (size_t i = 0; i < bytes; i++)
            if (hb[i] != 0x5a) { mg::set_last_error("mg_comm_selftest: payload mismatch"); return MG_ERR_COMM; }
        if (out != 1.25) { mg::set_last_error("mg_comm_selftest: all-reduce mismatch"); return MG_ERR_COMM; }
        return MG_OK;
    });
}

static int create_with_comm(const mg_desc *desc, int device, int rank, int nranks, mg::Comm *comm, mg_handle *out)
{
    (void)rank; (void)nranks;
    mg::Solver *s = new mg::Solver(*desc, device, comm);  // owns comm
    int rc = s->init();
    if (rc) { delete s; return rc; }
    *out = new mg_solver{s};
    return MG_OK;
}

int mg_create_distributed(const mg_desc *desc, int device, int rank, int nranks, const void *id128,
                          mg_handle *out)
{
    return guarded([&]() -> int {
        if (!out) return bad("mg_create_distributed: null output handle");
        *out = nullptr;
        if (nranks == 1 && rank == 0) return mg_create(desc, device, out);
        std::string why;
        int rc = mg::validate_desc(desc, &why);
        if (rc) { mg::set_last_error("mg_create_distributed: " + why); return rc; }
        mg::SlabPlan p;
        rc = mg::plan_slab(*desc, nranks, rank, 0, &p, &why);
        if (rc) { mg::set_last_error("mg_create_distributed: " + why); return rc; }
        if (device >= 0 && hipSetDevice(device) != hipSuccess) return bad("mg_create_distributed: bad device");
        mg::Comm *c = mg::make_rccl_comm(rank, nranks, id128, &why);
        if (!c) { mg::set_last_error("mg_create_distributed: " + why); return MG_ERR_COMM; }
        return create_with_comm(desc, device, rank, nranks, c, out);
    });
}

int mg_create_distributed_hostcomm(const mg_desc *desc, int device, int rank, int nranks,
                                   const mg_host_comm *comm, mg_handle *out)
{
    return guarded([&]() -> int {
        if (!out || !comm) return bad("mg_create_distributed_hostcomm: null argument");
        *out = nullptr;
        std::string why;
        int rc = mg::validate_desc(desc, &why);
        if (rc) { mg::set_last_error("mg_create_distributed_hostcomm: " + why); return rc; }
        mg::SlabPlan p;
        rc = mg::plan_slab(*desc, nranks, rank, 0, &p, &why);
        if (rc) { mg::set_last_error("mg_create_distributed_hostcomm: " + why); return rc; }
        mg::Comm *c = mg::make_host_comm(rank, nranks, *comm, &why);
        if (!c) { mg::set_last_error("mg_create_distributed_hostcomm: " + why); return MG_ERR_COMM; }
        return create_with_comm(desc, device, rank, nranks, c, out);
    });
}

int mg_create_distributed_dryrun(const mg_desc *desc, int device, int rank, int nranks, mg_handle *out)
{
    return guarded([&]() -> int {
        if (!out) return bad("mg_create_distributed_dryrun: null output handle");
        *out = nullptr;
        std::string why;
        int rc = mg::validate_desc(desc, &why);
        if (rc) { mg::set_last_error("mg_create_distributed_dryrun: " + why); return rc; }
        mg::SlabPlan p;
        rc = mg::plan_slab(*desc, nranks, rank, 0, &p, &why);
        if (rc) { mg::set_last_error("mg_create_distributed_dryrun: " + why); return rc; }
        return create_with_comm(desc, device, rank, nranks, mg::make_dry_comm(rank, nranks), out);
    });
}

}  // extern "C"
