// mg_solver.cpp -- grid hierarchy in HBM + stream-ordered cycle drivers.
// Reference call structure being replaced: SawtoothMGIteration
// (include/multigrid.hpp:108-145) and the outer loop of src/main.cpp:72-116.
#include "mg_solver.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mg_kernels.h"

namespace mg {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

#define MG_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return MG_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

int validate_desc(const mg_desc *d, std::string *why)
{
    auto fail = [&](const char *m) { if (why) *why = m; return (int)MG_ERR_INVALID_DESC; };
    if (!d) return fail("null descriptor");
    if (d->dim != 2 && d->dim != 3) return fail("dim must be 2 or 3");
    if (d->levels < 1 || d->levels > 16) return fail("levels must be in 1..16");
    if (d->n < 3) return fail("n must be >= 3");
    long step = 1L << (d->levels - 1);
    // The reference accepts any n (e.g. its defaults n=200, levels=2) and then reads
    // out of range on the coarse grid; we refuse instead.
    if ((d->n - 1) % step != 0) return fail("n-1 must be a multiple of 2^(levels-1)");
    if ((d->n - 1) / step + 1 < 3) return fail("coarsest grid would have fewer than 3 nodes per side");
    if (d->dtype != MG_F64 && d->dtype != MG_F32) return fail("dtype must be MG_F64 or MG_F32");
    if (d->smoother < MG_SMOOTH_GS_LEX || d->smoother > MG_SMOOTH_ZEBRA_X) return fail("unknown smoother");
    if (d->cycle != MG_CYCLE_SAWTOOTH && d->cycle != MG_CYCLE_V) return fail("unknown cycle kind");
    if (d->restriction != MG_RESTRICT_INJECT && d->restriction != MG_RESTRICT_FULLW) return fail("unknown restriction");
    if (d->coarse_mode != MG_COARSE_TOL && d->coarse_mode != MG_COARSE_FIXED) return fail("unknown coarse mode");
    if (!(d->length > 0) || !(d->alpha > 0)) return fail("length and alpha must be positive");
    if (d->coarse_maxit < 0 || d->nu_pre < 0 || d->nu_post < 0 || d->outer_pre_gs < 0)
        return fail("negative sweep count");
    for (int a = 0; a < 3; a++)
        if (!(d->aniso[a] > 0)) return fail("aniso multipliers must be positive");
    if (d->semi_xy < 0 || d->semi_xy > d->levels - 1) return fail("semi_xy must be in 0..levels-1");
    if (d->semi_xy && d->dim != 3) return fail("semi-coarsening (semi_xy) needs dim == 3");
    return MG_OK;
}

static bool is_zebra(int smoother) { return smoother == MG_SMOOTH_ZEBRA_Y || smoother == MG_SMOOTH_ZEBRA_X; }
// the coarsest-grid solver of a zebra hierarchy smooths with red-black Gauss-Seidel (mg_desc.h)
static int coarse_smoother_of(int smoother) { return is_zebra(smoother) ? MG_SMOOTH_RBGS : smoother; }

int level_n(const mg_desc &d, int level)
{
    int n = d.n;
    for (int l = 0; l < level; l++) n = (n + 1) / 2;  // reference src/domain.cpp:9-12
    return n;
}

void level_coefficients(const mg_desc &d, int level, double out[4])
{
    // reference: m_h = length/(size-1) (domain.cpp:5); h() = m_h*step (domain.hpp:90);
    // k = h*h (linear_system.hpp:17); -alpha/k and 4*alpha/k (linear_system.hpp:27-28,37-38)
    double m_h = d.length / (double)(d.n - 1);
    double step = (double)(1L << level);
    double h = m_h * step;
    double k = h * h;
    out[0] = -(d.alpha * d.aniso[0]) / k;
    out[1] = -(d.alpha * d.aniso[1]) / k;
    out[2] = -(d.alpha * d.aniso[2]) / k;
    double s = (d.dim == 3) ? (d.aniso[0] + d.aniso[1] + d.aniso[2]) : (d.aniso[0] + d.aniso[1]);
    out[3] = ((2.0 * s) * d.alpha) / k;
    if (d.dim == 3 && d.semi_xy) {  // z is coarsened only after the first semi_xy transitions
        int lz = level > d.semi_xy ? level - d.semi_xy : 0;
        double hz = m_h * (double)(1L << lz);
        double kz = hz * hz;
        out[2] = -(d.alpha * d.aniso[2]) / kz;
        out[3] = 2.0 * ((d.alpha * d.aniso[0]) / k + (d.alpha * d.aniso[1]) / k + (d.alpha * d.aniso[2]) / kz);
    }
}

int level_nz(const mg_desc &d, int level)
{
    if (d.dim != 3) return 1;
    int nz = d.n;  // the first semi_xy transitions keep z, the later ones halve it
    for (int l = d.semi_xy; l < level; l++) nz = (nz + 1) / 2;
    return nz;
}

int plan_slab(const mg_desc &d, int nranks, int rank, int level, SlabPlan *out, std::string *why)
{
    auto fail = [&](const char *m) { if (why) *why = m; return (int)MG_ERR_BAD_ARG; };
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank / nranks");
    if (level < 0 || level >= d.levels) return fail("bad level");
    if (nranks > 1 && d.dim != 3) return fail("domain decomposition needs dim == 3");
    // z-cells of level l: the first semi_xy transitions do not coarsen z
    auto zshift = [&](int l) { return (d.dim == 3 && l > d.semi_xy) ? l - d.semi_xy : 0; };
    auto zcells = [&](int l) { return (d.n - 1) >> zshift(l); };
    // Distributed levels 0..Ld-1: every rank keeps >= 2 z-cells and the level holds at least
    // dist_min_n^3 points (below that a halo exchange costs more than the sweep it feeds);
    // coarser levels are agglomerated on rank 0. Level 0 is always distributed.
    const long long min_n = d.dist_min_n > 0 ? d.dist_min_n : 257;
    int Ld = d.levels;
    if (nranks > 1) {
        Ld = 0;
        for (int l = 0; l < d.levels; l++) {
            long long nl = level_n(d, l);
            long long pts = nl * nl * (zcells(l) + 1);
            if (zcells(l) >= 2 * nranks && (l == 0 || pts >= min_n * min_n * min_n)) Ld = l + 1; else break;
        }
        if (Ld == 0) return fail("grid too small for this many ranks");
    }
    out->first_gathered_level = Ld;
    if (d.dim == 2) { out->z0 = 0; out->nz = 1; return MG_OK; }
    if (level >= Ld) {  // gathered: rank 0 owns everything
        out->z0 = 0;
        out->nz = (rank == 0) ? level_nz(d, level) : 0;
        return MG_OK;
    }
    // Split the z-cells of the first gathered level (of the coarsest level if none is gathered);
    // finer levels inherit the split, so a coarse plane K always lives with the fine plane it
    // coincides with (2K, or K when z is kept) -- including the planes of the first gathered
    // level each rank restricts into before they are gathered (plan_stage).
    const int base = (Ld < d.levels) ? Ld : Ld - 1;
    int cellsC = zcells(base);
    int shift = zshift(base) - zshift(level);
    long s0 = ((long)cellsC * rank) / nranks, s1 = ((long)cellsC * (rank + 1)) / nranks;
    out->z0 = (int)(s0 << shift);
    out->nz = (int)((s1 - s0) << shift) + ((rank == nranks - 1) ? 1 : 0);
    return MG_OK;
}

int plan_stage(const mg_desc &d, int nranks, int rank, SlabPlan *out, std::string *why)
{
    SlabPlan p0;
    int rc = plan_slab(d, nranks, rank, 0, &p0, why);
    if (rc) return rc;
    const int S = p0.first_gathered_level;
    out->first_gathered_level = S;
    if (S >= d.levels || nranks == 1) { out->z0 = 0; out->nz = 0; return MG_OK; }
    auto zshift = [&](int l) { return (d.dim == 3 && l > d.semi_xy) ? l - d.semi_xy : 0; };
    const int cellsC = (d.n - 1) >> zshift(S);
    long s0 = ((long)cellsC * rank) / nranks, s1 = ((long)cellsC * (rank + 1)) / nranks;
    out->z0 = (int)s0;
    out->nz = (int)(s1 - s0) + ((rank == nranks - 1) ? 1 : 0);
    return MG_OK;
}

Solver::Solver(const mg_desc &d, int device, Comm *comm) : d_(d), device_(device), comm_(comm)
{
    if (comm_) { rank_ = comm_->rank; nranks_ = comm_->nranks; }
}

Solver::~Solver()
{
    if (device_ >= 0) (void)hipSetDevice(device_);
    for (auto &L : lv_)
        for (auto &b : L.base)
            if (b) (void)hipFree(b);
    for (auto &L : lv_) if (L.zebra) (void)hipFree(L.zebra);
    for (auto &f : full_) if (f) (void)hipFree(f);
    for (auto &b : stage_base_) if (b) (void)hipFree(b);
    delete comm_;
    if (ev_ready_) (void)hipEventDestroy(ev_ready_);
    if (ev_halo_) (void)hipEventDestroy(ev_halo_);
    if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
    if (d_partials_) (void)hipFree(d_partials_);
    if (d_scal_) (void)hipFree(d_scal_);
    if (d_coarse_) (void)hipFree(d_coarse_);
    if (h_scal_) (void)hipHostFree(h_scal_);
    if (h_coarse_) (void)hipHostFree(h_coarse_);
    if (h_fixed_) (void)hipHostFree(h_fixed_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    for (auto &e : prof_ev_) (void)hipEventDestroy(e);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    for (hipEvent_t e : ev_stage_) if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
}

int Solver::init()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_last_error("no HIP device: libmg_hip has no CPU fallback");
        return MG_ERR_NO_DEVICE;
    }
    if (device_ < 0) MG_HIP(hipGetDevice(&device_));
    if (device_ >= ndev) { set_last_error("device index out of range"); return MG_ERR_BAD_ARG; }
    MG_HIP(hipSetDevice(device_));
    MG_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    MG_HIP(hipEventCreate(&ev0_));
    MG_HIP(hipEventCreate(&ev1_));
    for (hipEvent_t &e : ev_stage_) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (nranks_ > 1) {
        // the communication stream outranks the main one: the exchange kernels and the boundary pieces behind them are
        // dispatched ahead of the interior launch they run beside (MG_COMM_PRIORITY=0: same priority)
        int prio_least = 0, prio_greatest = 0;
        const char *pe = getenv("MG_COMM_PRIORITY");
        if (!(pe && pe[0] == '0') && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_greatest != prio_least)
            MG_HIP(hipStreamCreateWithPriority(&comm_stream_, hipStreamNonBlocking, prio_greatest));
        else
            MG_HIP(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
        MG_HIP(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
        MG_HIP(hipEventCreateWithFlags(&ev_halo_, hipEventDisableTiming));
        const char *ov = getenv("MG_OVERLAP");
        overlap_ = !(ov && ov[0] == '0');
    }

    const int epl = (int)(128 / esize());  // elements per 128-byte line
    lv_.resize(d_.levels);
    int max_partials = 1;
    for (int l = 0; l < d_.levels; l++) {
        Level &L = lv_[l];
        int n = level_n(d_, l);
        L.g.dim = d_.dim;
        L.g.nx = n; L.g.ny = n; L.g.nz = level_nz(d_, l);
        L.g.pitch = ((n + epl - 1) / epl) * epl;
        L.g.plane = (long long)L.g.ny * L.g.pitch;
        L.g.gz0 = 0; L.g.gnz = L.g.nz;
        level_coefficients(d_, l, L.coef);
        if (nranks_ > 1) {
            SlabPlan p;
            std::string why;
            int rc = plan_slab(d_, nranks_, rank_, l, &p, &why);
            if (rc) { set_last_error("mg_create_distributed: " + why); return rc; }
            T_ = p.first_gathered_level - 1;
            L.dist = l < p.first_gathered_level;
            // Gathered levels: by default EVERY rank holds them and runs them redundantly (same bits everywhere) after an
            // all-gather of the restricted right-hand side -- nobody waits for rank 0 to scatter the correction back, and
            // rank 0 is no longer the one rank with more work. MG_REPLICATE_TAIL=0: rank 0 alone, gather + scatter.
            replicate_ = [] { const char *e = getenv("MG_REPLICATE_TAIL"); return !(e && e[0] == '0'); }();
            L.present = L.dist || rank_ == 0 || replicate_;
            if (L.dist) {
                if (l == T_ && rank_ == 0 && T_ == d_.levels - 1) {  // coarsest level still distributed: rank 0 solves it gathered
                    gfull_ = L.g;
                    for (auto &f : full_) {
                        size_t nbytes = (size_t)(gfull_.nz + 2) * (size_t)gfull_.plane * esize();
                        MG_HIP(hipMalloc(&f, nbytes));
                        MG_HIP(hipMemsetAsync(f, 0, nbytes, stream_));
                        bytes_ += nbytes;
                    }
                    max_partials = std::max(max_partials, reduce_partials_capacity(gfull_));
                }
                if (l == T_) {
                    planT_.resize(nranks_);
                    for (int r = 0; r < nranks_; r++) plan_slab(d_, nranks_, r, l, &planT_[r], &why);
                }
                L.g.nz = p.nz; L.g.gz0 = p.z0;
                L.nz_min = p.nz;
                for (int r = 0; r < nranks_; r++) {
                    SlabPlan q;
                    if (plan_slab(d_, nranks_, r, l, &q, &why) == MG_OK) L.nz_min = std::min(L.nz_min, q.nz);
                }
            }
        }
        if (nranks_ > 1 && l == T_ + 1 && l < d_.levels) {
            // staging slab of the first gathered level: the planes that coincide with this rank's
            // slab of level T_ (restriction target / prolongation source around the gather)
            std::string why;
            planS_.resize(nranks_);
            for (int r = 0; r < nranks_; r++) {
                int rc = plan_stage(d_, nranks_, r, &planS_[r], &why);
                if (rc) { set_last_error("mg_create_distributed: " + why); return rc; }
            }
            stage_g_ = L.g;
            stage_g_.nz = planS_[rank_].nz; stage_g_.gz0 = planS_[rank_].z0;
            for (auto &b : stage_base_) {
                size_t nbytes = (size_t)(stage_g_.nz + 2) * (size_t)stage_g_.plane * esize();
                MG_HIP(hipMalloc(&b, nbytes));
                MG_HIP(hipMemsetAsync(b, 0, nbytes, stream_));
                bytes_ += nbytes;
            }
        }
        if (!L.dist) L.nz_min = L.g.nz;
        {
            // Interior / boundary split with the exchange on the communication stream: two cross-stream waits (6-20 us each on
            // this chip) and a boundary launch per operation. It pays while the interior launch is longer than the exchange;
            // on a slab of a few MB (257^3 on 8 ranks: 17 MB per array, interior pair 20 us, halo 1 MB per neighbour) it does
            // not, and exchange-then-one-launch takes the same time with free communication 13-20 us less per operation.
            const char *e = getenv("MG_OVERLAP_MIN_MB");   // read per handle (tests switch it between handles of one process)
            const double min_mb = e ? atof(e) : 32.0;
            L.overlap = overlap_ && L.dist && (double)L.nz_min * (double)L.g.plane * (double)esize() >= min_mb * 1048576.0;
        }
        L.gh = L.dist ? 2 : 1;
        L.alloc_elems = (size_t)(L.g.nz + 2 * L.gh) * (size_t)L.g.plane;
        if (!L.present) continue;
        for (int a = 0; a < NUM_ARR; a++) {
            if (a == MG_ARR_RES && l > 0) continue;
            size_t nbytes = L.alloc_elems * esize();
            MG_HIP(hipMalloc(&L.base[a], nbytes));
            MG_HIP(hipMemsetAsync(L.base[a], 0, nbytes, stream_));
            bytes_ += nbytes;
        }
        if (is_zebra(d_.smoother)) {  // elimination factors of the line solve: along y (cy, ny) or along x (cx, nx)
            const bool alongx = d_.smoother == MG_SMOOTH_ZEBRA_X;
            const int nl = alongx ? L.g.nx : L.g.ny;
            const double cl = alongx ? L.coef[0] : L.coef[1];
            const size_t nb = 2 * (size_t)nl * esize();
            MG_HIP(hipMalloc(&L.zebra, nb));
            if (d_.dtype == MG_F64) {
                std::vector<double> f(2 * (size_t)nl);
                zebra_line_factors<double>(cl, L.coef[3], nl, f.data());
                MG_HIP(hipMemcpy(L.zebra, f.data(), nb, hipMemcpyHostToDevice));
            } else {
                std::vector<float> f(2 * (size_t)nl);
                zebra_line_factors<float>((float)cl, (float)L.coef[3], nl, f.data());
                MG_HIP(hipMemcpy(L.zebra, f.data(), nb, hipMemcpyHostToDevice));
            }
            bytes_ += nb;
        }
        int cap = reduce_partials_capacity(L.g);
        if (cap > max_partials) max_partials = cap;
    }
    MG_HIP(hipMalloc((void **)&d_partials_, sizeof(double) * (size_t)max_partials));
    MG_HIP(hipMalloc((void **)&d_scal_, sizeof(double) * 8));
    MG_HIP(hipMalloc((void **)&d_coarse_, sizeof(CoarseOut)));
    MG_HIP(hipMemsetAsync(d_scal_, 0, sizeof(double) * 8, stream_));
    MG_HIP(hipMemsetAsync(d_coarse_, 0, sizeof(CoarseOut), stream_));
    MG_HIP(hipHostMalloc((void **)&h_scal_, sizeof(double) * 8));
    MG_HIP(hipHostMalloc((void **)&h_coarse_, sizeof(CoarseOut)));
    MG_HIP(hipHostMalloc((void **)&h_fixed_, sizeof(CoarseOut)));
    bytes_ += sizeof(double) * ((size_t)max_partials + 8) + sizeof(CoarseOut);
    MG_HIP(hipStreamSynchronize(stream_));
    return MG_OK;
}

bool Solver::check_arr(int which, int level, const char *fn) const
{
    if (level < 0 || level >= d_.levels || which < 0 || which >= NUM_ARR || !lv_[level].base[which]) {
        set_last_error(std::string(fn) + ": no such array/level");
        return false;
    }
    return true;
}

template <typename T>
T *Solver::ptr(int which, int level) const
{
    const Level &L = lv_[level];
    return reinterpret_cast<T *>(L.base[which]) + L.gh * L.g.plane;  // skip the lower ghost plane(s)
}

// Host <-> device copies of a level's array (dense rows on the host, 128-byte-pitched rows on the device)
// go through a pinned staging buffer in whole padded planes: one contiguous DMA per chunk and the row
// (un)packing on the host. hipMemcpy2DAsync from pageable memory took 5 ms for a 257^2 array (0.1 GB/s),
// a third of the whole solve of the reference's own case.
// Round 3: the staging buffer has two halves and the row (un)packing runs on a few host threads, so the DMA of one chunk
// overlaps the packing of the next (1.08 GB at 513^3: 16-19 GB/s up and 8-10 GB/s down before, when chunk k's packing, its DMA
// and the wait for it ran one after the other on one thread).
namespace {
// fn(first_row, last_row) over [0, nrows) on up to `nthreads` threads (the calling thread takes the first share)
template <typename F>
void parallel_rows(size_t nrows, int nthreads, F &&fn)
{
    nthreads = (int)std::max<size_t>(1, std::min<size_t>((size_t)nthreads, nrows / 64));
    if (nthreads == 1) { fn((size_t)0, nrows); return; }
    std::vector<std::thread> th;
    const size_t per = (nrows + nthreads - 1) / nthreads;
    for (int t = 1; t < nthreads; t++) {
        const size_t lo = std::min(nrows, t * per), hi = std::min(nrows, lo + per);
        if (lo < hi) th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
    }
    fn((size_t)0, std::min(nrows, per));
    for (auto &t : th) t.join();
}
int stage_threads()
{
    static const int n = [] {
        const char *e = getenv("MG_STAGE_THREADS");
        if (e) return std::max(1, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(8u, hc ? hc / 2 : 4u));
    }();
    return n;
}
}  // namespace

int Solver::stage_rows(int which, int level, void *host, bool to_device)
{
    pair_on_comm_level_ = -1;
    MG_HIP(hipSetDevice(device_));
    const Level &L = lv_[level];
    const size_t es = esize(), row = (size_t)L.g.nx * es, prow = (size_t)L.g.pitch * es;
    const size_t pbytes = (size_t)L.g.plane * es;
    const size_t want = std::max<size_t>((size_t)64 << 20, 2 * pbytes);   // two halves, each at least one plane
    if (!h_stage_ || h_stage_bytes_ < want) {
        if (h_stage_) { (void)hipHostFree(h_stage_); h_stage_ = nullptr; }
        h_stage_bytes_ = want;
        MG_HIP(hipHostMalloc(&h_stage_, h_stage_bytes_));
    }
    const size_t half = h_stage_bytes_ / 2;
    const int per = (int)std::max<size_t>(1, half / pbytes);  // planes per chunk
    char *dev = reinterpret_cast<char *>(L.base[which]) + (size_t)L.gh * pbytes;      // local plane 0
    if (to_device && which == MG_ARR_RHS) lv_[level].rhs_halo_ok = false;
    char *hp = reinterpret_cast<char *>(host);
    // (threads only where there is something to share: a 257^2 array is 0.5 MB, a thread costs ~30 us to start)
    const int ny = L.g.ny, nthr = (pbytes * (size_t)std::min(per, L.g.nz) >= ((size_t)8 << 20)) ? stage_threads() : 1;
    const int nchunks = (L.g.nz + per - 1) / per;
    auto half_ptr = [&](int k) { return reinterpret_cast<char *>(h_stage_) + (size_t)(k & 1) * half; };
    auto pack = [&](int k) {        // host rows of chunk k -> its staging half, padding columns zeroed
        const int z0 = k * per, nzc = std::min(per, L.g.nz - z0);
        char *st = half_ptr(k);
        parallel_rows((size_t)nzc * ny, nthr, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; r++) {
                char *d = st + r * prow;      // plane pitch == ny * row pitch: the rows of a chunk are equally spaced
                std::memcpy(d, hp + ((size_t)z0 * ny + r) * row, row);
                std::memset(d + row, 0, prow - row);
            }
        });
    };
    auto unpack = [&](int k) {
        const int z0 = k * per, nzc = std::min(per, L.g.nz - z0);
        const char *st = half_ptr(k);
        parallel_rows((size_t)nzc * ny, nthr, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; r++) std::memcpy(hp + ((size_t)z0 * ny + r) * row, st + r * prow, row);
        });
    };
    auto dma = [&](int k) -> int {
        const int z0 = k * per, nzc = std::min(per, L.g.nz - z0);
        if (to_device) MG_HIP(hipMemcpyAsync(dev + (size_t)z0 * pbytes, half_ptr(k), (size_t)nzc * pbytes, hipMemcpyHostToDevice, stream_));
        else MG_HIP(hipMemcpyAsync(half_ptr(k), dev + (size_t)z0 * pbytes, (size_t)nzc * pbytes, hipMemcpyDeviceToHost, stream_));
        MG_HIP(hipEventRecord(ev_stage_[k & 1], stream_));
        return MG_OK;
    };
    auto wait_half = [&](int k) -> int { MG_HIP(hipEventSynchronize(ev_stage_[k & 1])); return MG_OK; };
    if (to_device) {
        for (int k = 0; k < nchunks; k++) {
            if (k >= 2) { int rc = wait_half(k); if (rc) return rc; }   // the half's previous DMA has read it
            pack(k);
            int rc = dma(k); if (rc) return rc;
        }
        MG_HIP(hipStreamSynchronize(stream_));
    } else {
        { int rc = dma(0); if (rc) return rc; }
        for (int k = 0; k < nchunks; k++) {
            if (k + 1 < nchunks) { int rc = dma(k + 1); if (rc) return rc; }   // the other half: unpacked two chunks ago
            int rc = wait_half(k); if (rc) return rc;
            unpack(k);
        }
    }
    return MG_OK;
}

int Solver::set_array(int which, int level, const void *host)
{
    if (!host || !check_arr(which, level, "mg_set_array")) return MG_ERR_BAD_ARG;
    return stage_rows(which, level, const_cast<void *>(host), true);
}

int Solver::get_array(int which, int level, void *host)
{
    if (!host || !check_arr(which, level, "mg_get_array")) return MG_ERR_BAD_ARG;
    return stage_rows(which, level, host, false);
}

int Solver::zero_array(int which, int level)
{
    pair_on_comm_level_ = -1;
    if (!check_arr(which, level, "mg_zero_array")) return MG_ERR_BAD_ARG;
    if (which == MG_ARR_RHS) lv_[level].rhs_halo_ok = false;
    MG_HIP(hipMemsetAsync(lv_[level].base[which], 0, lv_[level].alloc_elems * esize(), stream_));
    return MG_OK;
}

#define MG_TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

int Solver::post(const P2POp *ops, int n, hipStream_t s)
{
    if (n > 0) comm_groups_++;
    for (int i = 0; i < n; i++) if (ops[i].send) comm_bytes_ += (long long)ops[i].bytes;
    return comm_->batch(ops, n, s);
}

// The halo operations of `depth` planes with the z-neighbours: my first / last `depth` owned planes go to their upper / lower
// ghost planes, theirs come into mine. Owned and ghost planes are contiguous, so each is one message. Returns the op count.
int Solver::halo_ops(int which, int level, int depth, P2POp *ops)
{
    Level &L = lv_[level];
    const size_t pb = (size_t)L.g.plane * esize();
    char *b = reinterpret_cast<char *>(L.base[which]);
    const size_t own0 = (size_t)L.gh * pb, nzb = (size_t)L.g.nz * pb, db = (size_t)depth * pb;
    int n = 0;
    if (rank_ > 0) {
        ops[n++] = P2POp{rank_ - 1, true, b + own0, db};                   // my first planes -> their upper ghosts
        ops[n++] = P2POp{rank_ - 1, false, b + own0 - db, db};             // their last planes -> my lower ghosts
    }
    if (rank_ < nranks_ - 1) {
        ops[n++] = P2POp{rank_ + 1, true, b + own0 + nzb - db, db};        // my last planes -> their lower ghosts
        ops[n++] = P2POp{rank_ + 1, false, b + own0 + nzb, db};
    }
    return n;
}

int Solver::exchange(int which, int level, int depth)
{
    Level &L = lv_[level];
    if (!L.dist) return MG_OK;
    if (depth > L.gh || depth > L.nz_min) { set_last_error("halo exchange deeper than the ghost planes / the thinnest slab"); return MG_ERR_BAD_ARG; }
    P2POp ops[4];
    const int n = halo_ops(which, level, depth, ops);
    int rc = post(ops, n, stream_);
    if (rc) set_last_error("halo exchange failed");
    return rc;
}

int Solver::exchange_begin(int which, int level, int depth, bool record)
{
    Level &L = lv_[level];
    if (depth > L.gh || depth > L.nz_min) { set_last_error("halo exchange deeper than the ghost planes / the thinnest slab"); return MG_ERR_BAD_ARG; }
    P2POp ops[4];
    const int n = halo_ops(which, level, depth, ops);
    MG_HIP(hipEventRecord(ev_ready_, stream_));
    MG_HIP(hipStreamWaitEvent(comm_stream_, ev_ready_, 0));
    int rc = post(ops, n, comm_stream_);
    if (rc) { set_last_error("halo exchange failed"); return rc; }
    if (record) MG_HIP(hipEventRecord(ev_halo_, comm_stream_));
    return MG_OK;
}

int Solver::halo_work_done()
{
    MG_HIP(hipEventRecord(ev_halo_, comm_stream_));
    return MG_OK;
}

// The boundary pieces of a slab operation (the planes that need the halo) run on the COMMUNICATION stream, right behind
// the exchange, while the interior piece runs on the main stream: measured on one rank's schedule (tools/dry_ranks.sh) the
// boundary launch after the interior one cost 14-18 us plus a cross-stream wait per operation, five operations per cycle.
static bool boundary_on_comm_stream()
{
    static const bool e = [] { const char *v = getenv("MG_BOUNDARY_ON_COMM"); return !(v && v[0] == '0'); }();
    return e;
}

int Solver::exchange_end()
{
    MG_HIP(hipStreamWaitEvent(stream_, ev_halo_, 0));
    return MG_OK;
}

template <typename F>
int Solver::overlapped(int level, int arr_x, F &&launch)
{
    Level &L = lv_[level];
    if (!L.dist) { launch(L.g, (long long)0); return MG_OK; }
    if (!L.overlap || L.g.nz < 4) {
        MG_TRY(exchange(arr_x, level));
        launch(L.g, (long long)0);
        return MG_OK;
    }
    MG_TRY(exchange_begin(arr_x, level));
    Geom gi = L.g;                      // interior planes 1 .. nz-2 never touch a ghost plane
    gi.nz = L.g.nz - 2; gi.gz0 = L.g.gz0 + 1;
    launch(gi, L.g.plane);
    MG_TRY(exchange_end());
    Geom g0 = L.g; g0.nz = 1;           // first and last owned plane, now that the halo is in
    launch(g0, (long long)0);
    Geom g1 = L.g; g1.nz = 1; g1.gz0 = L.g.gz0 + L.g.nz - 1;
    launch(g1, (long long)(L.g.nz - 1) * L.g.plane);
    return MG_OK;
}

int Solver::gather_S(int arr)
{
    const Level &L = lv_[T_ + 1];
    const size_t pb = (size_t)stage_g_.plane * esize();
    char *sb = reinterpret_cast<char *>(stage_base_[0]);
    int rc;
    if (replicate_) {  // all-gather: my staging slab to everybody, everybody's into my full copy of the level
        char *f = reinterpret_cast<char *>(L.base[arr]);
        MG_HIP(hipMemcpyAsync(f + (size_t)(1 + planS_[rank_].z0) * pb, sb + pb, (size_t)stage_g_.nz * pb, hipMemcpyDeviceToDevice, stream_));
        std::vector<P2POp> ops;
        for (int r = 0; r < nranks_; r++) {
            if (r == rank_) continue;
            ops.push_back(P2POp{r, true, sb + pb, (size_t)stage_g_.nz * pb});
            ops.push_back(P2POp{r, false, f + (size_t)(1 + planS_[r].z0) * pb, (size_t)planS_[r].nz * pb});
        }
        rc = post(ops.data(), (int)ops.size(), stream_);
        if (rc) set_last_error("all-gather of the coarse right-hand side failed");
        return rc;
    }
    if (rank_ == 0) {
        char *f = reinterpret_cast<char *>(L.base[arr]);
        MG_HIP(hipMemcpyAsync(f + pb, sb + pb, (size_t)stage_g_.nz * pb, hipMemcpyDeviceToDevice, stream_));
        std::vector<P2POp> ops;
        for (int r = 1; r < nranks_; r++)
            ops.push_back(P2POp{r, false, f + (size_t)(1 + planS_[r].z0) * pb, (size_t)planS_[r].nz * pb});
        rc = post(ops.data(), (int)ops.size(), stream_);
    } else {
        P2POp op{0, true, sb + pb, (size_t)stage_g_.nz * pb};
        rc = post(&op, 1, stream_);
    }
    if (rc) set_last_error("gather to rank 0 failed");
    return rc;
}

int Solver::scatter_S(int arr)
{
    // every rank receives its planes plus the next one (upper ghost of the prolongation)
    const Level &L = lv_[T_ + 1];
    const size_t pb = (size_t)stage_g_.plane * esize();
    char *sb = reinterpret_cast<char *>(stage_base_[1]);
    auto planes = [&](int r) { return (size_t)planS_[r].nz + (r < nranks_ - 1 ? 1 : 0); };
    int rc;
    if (replicate_) {  // every rank computed the level: its own planes (+ the upper ghost) are a local copy away
        char *f = reinterpret_cast<char *>(L.base[arr]);
        MG_HIP(hipMemcpyAsync(sb + pb, f + (size_t)(1 + planS_[rank_].z0) * pb, planes(rank_) * pb, hipMemcpyDeviceToDevice, stream_));
        return MG_OK;
    }
    if (rank_ == 0) {
        char *f = reinterpret_cast<char *>(L.base[arr]);
        MG_HIP(hipMemcpyAsync(sb + pb, f + pb, planes(0) * pb, hipMemcpyDeviceToDevice, stream_));
        std::vector<P2POp> ops;
        for (int r = 1; r < nranks_; r++)
            ops.push_back(P2POp{r, true, f + (size_t)(1 + planS_[r].z0) * pb, planes(r) * pb});
        rc = post(ops.data(), (int)ops.size(), stream_);
    } else {
        P2POp op{0, false, sb + pb, planes(rank_) * pb};
        rc = post(&op, 1, stream_);
    }
    if (rc) set_last_error("scatter from rank 0 failed");
    return rc;
}

int Solver::gather_T(int which, int fullk)
{
    Level &L = lv_[T_];
    const size_t pb = (size_t)L.g.plane * esize();
    char *b = reinterpret_cast<char *>(L.base[which]);
    int rc;
    if (rank_ == 0) {
        char *f = reinterpret_cast<char *>(full_[fullk]);
        MG_HIP(hipMemcpyAsync(f + pb, b + (size_t)L.gh * pb, (size_t)L.g.nz * pb, hipMemcpyDeviceToDevice, stream_));
        std::vector<P2POp> ops;
        for (int r = 1; r < nranks_; r++)
            ops.push_back(P2POp{r, false, f + (size_t)(1 + planT_[r].z0) * pb, (size_t)planT_[r].nz * pb});
        rc = post(ops.data(), (int)ops.size(), stream_);
    } else {
        P2POp op{0, true, b + (size_t)L.gh * pb, (size_t)L.g.nz * pb};
        rc = post(&op, 1, stream_);
    }
    if (rc) set_last_error("gather to rank 0 failed");
    return rc;
}

int Solver::scatter_T(int fullk, int which)
{
    pair_on_comm_level_ = -1;   // this call writes a level's arrays on the main stream: no exchange may skip its wait for that stream (every writer says so)
    Level &L = lv_[T_];
    const size_t pb = (size_t)L.g.plane * esize();
    char *b = reinterpret_cast<char *>(L.base[which]);
    int rc;
    if (rank_ == 0) {
        char *f = reinterpret_cast<char *>(full_[fullk]);
        MG_HIP(hipMemcpyAsync(b + (size_t)L.gh * pb, f + pb, (size_t)L.g.nz * pb, hipMemcpyDeviceToDevice, stream_));
        std::vector<P2POp> ops;
        for (int r = 1; r < nranks_; r++)
            ops.push_back(P2POp{r, true, f + (size_t)(1 + planT_[r].z0) * pb, (size_t)planT_[r].nz * pb});
        rc = post(ops.data(), (int)ops.size(), stream_);
    } else {
        P2POp op{0, false, b + (size_t)L.gh * pb, (size_t)L.g.nz * pb};
        rc = post(&op, 1, stream_);
    }
    if (rc) set_last_error("scatter from rank 0 failed");
    return rc;
}

int Solver::allreduce(double *dptr, int n)
{
    if (nranks_ == 1) return MG_OK;
    comm_groups_++; comm_bytes_ += 8LL * n;
    int rc = comm_->allreduce_sum(dptr, n, stream_);
    if (rc) set_last_error("allreduce failed");
    return rc;
}

template <typename T>
static Coef<T> coef_of(const Level &L)
{
    static const bool fast_div = [] { const char *e = getenv("MG_FAST_DIV"); return !(e && e[0] == '0'); }();
    Coef<T> c = make_coef<T>(L.coef[0], L.coef[1], L.coef[2], L.coef[3]);
    if (!fast_div) c.win = 0;  // A/B switch: hardware division everywhere
    return c;
}

// true when the first pre-smoothing sweep of `level` can consume an implicit zero guess
template <typename T>
bool Solver::can_skip_zeroing(int level) const
{
    if (d_.nu_pre <= 0 || level >= d_.levels - 1) return false;
    if (d_.smoother == MG_SMOOTH_JACOBI) return fast_path_ok<T>(lv_[level].g);
    // red-black: the one-pass sweep of an undistributed level (smooth_t takes exactly this branch for it)
    static const bool rb_zero = [] { const char *e = getenv("MG_RB_ZERO_GUESS"); return !(e && e[0] == '0'); }();
    return rb_zero && d_.smoother == MG_SMOOTH_RBGS && !lv_[level].dist && rb_fused_ok<T>(lv_[level].g);
}

// true when the V-cycle's prolong-add into `level` can be folded into the first post-smoothing pair
// (k_jacobi2<CORR>): out = J(J(u + P e)) without ever storing u + P e
static Geom slab_gate_geom(const Level &L);
static bool depth2_enabled();

// the same on two consecutive distributed levels (Jacobi, two ghost planes): every piece of the slab pair folds P e in
template <typename T>
bool Solver::can_fold_prolong_slab(int level) const
{
    static const bool enabled = [] { const char *e = getenv("MG_FUSED_PROLONG"); return !(e && e[0] == '0'); }();
    if (!enabled || level + 1 >= d_.levels || !lv_[level].dist || !lv_[level + 1].dist) return false;
    if (d_.smoother != MG_SMOOTH_JACOBI || d_.nu_post < 2 || !depth2_enabled() || lv_[level + 1].gh < 2 || lv_[level + 1].nz_min < 2) return false;
    // the same answer on every rank: the gate looks at the thinnest slabs, whose z-relation holds like everybody's
    Geom gf = slab_gate_geom(lv_[level]), gc = slab_gate_geom(lv_[level + 1]);
    gf.nz = 2 * gc.nz;
    return jacobi2_slab_ok<T>(slab_gate_geom(lv_[level])) && jacobi2_corr_slab_ok<T>(gf, gc) &&
           jacobi2_corr_slab_ok<T>(lv_[level].g, lv_[level + 1].g);
}

// the last distributed level over the first replicated one (MG_REPLICATE_TAIL): every rank holds the whole coarse correction, the
// kernels address coarse planes by global index, so the slab pair folds P e in straight from the replicated array -- no copy
// of the rank's planes into the staging slab, no prolongation launch (13-15 us per cycle at 513^3 on 8 ranks)
template <typename T>
bool Solver::can_fold_prolong_replicated(int level) const
{
    static const bool enabled = [] {
        const char *e = getenv("MG_FUSED_PROLONG"), *r = getenv("MG_FUSED_PROLONG_REPLICATED");
        return !(e && e[0] == '0') && !(r && r[0] == '0');
    }();
    if (!enabled || !replicate_ || level + 1 >= d_.levels || !lv_[level].dist || lv_[level + 1].dist || !lv_[level + 1].present) return false;
    if (d_.smoother != MG_SMOOTH_JACOBI || d_.nu_post < 2 || !depth2_enabled()) return false;
    const Geom &gf = lv_[level].g, &gc = lv_[level + 1].g;
    return jacobi2_slab_ok<T>(slab_gate_geom(lv_[level])) && gc.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 &&
           gf.gnz == 2 * gc.gnz - 1 && gc.gz0 == 0 && gc.gnz == gc.nz && lv_[level].nz_min >= 4;
}

template <typename T>
bool Solver::can_fold_prolong(int level) const
{
    static const bool enabled = [] { const char *e = getenv("MG_FUSED_PROLONG"); return !(e && e[0] == '0'); }();
    const bool sm = (d_.smoother == MG_SMOOTH_JACOBI && d_.nu_post >= 2) ||
                    (d_.smoother == MG_SMOOTH_RBGS && d_.nu_post >= 1 && rb_fused_ok<T>(lv_[level].g));
    return enabled && sm && level + 1 < d_.levels &&
           lv_[level].present && !lv_[level].dist && lv_[level + 1].present && !lv_[level + 1].dist &&
           jacobi2_corr_ok<T>(lv_[level].g, lv_[level + 1].g);
}

static bool rb_slab_enabled()
{
    static const bool e = [] { const char *v = getenv("MG_FUSED_RB"); return !(v && v[0] == '0'); }();
    return e;
}

// Two Jacobi sweeps (or, rb, the two colour passes of one red-black sweep) of U on a z-slab with the fused kernel (depth-1 ghost planes are enough):
//   a. halo of u;
//   b. sweep 1 on the two outermost planes of either end -> v in the level's E array (free in a
//      V-cycle), and the halo of v starts moving on the communication stream;
//   c. meanwhile the fused pair computes the output planes 1 .. nz-2 (its internal first sweep needs
//      u on planes -1 .. nz, i.e. only the ghosts of step a);
//   d. sweep 2 on planes 0 and nz-1 from v once its ghosts are in.
// Same arithmetic per point as two exchanged single sweeps => same bits as one GPU.
template <typename T>
int Solver::pair_on_slab_t(int level, bool rb)
{
    Level &L = lv_[level];
    const Geom &g = L.g;
    Coef<T> c = coef_of<T>(L);
    const T om = (T)d_.omega;
    T *px = ptr<T>(MG_ARR_U, level), *pr = ptr<T>(MG_ARR_RHS, level), *pt = ptr<T>(MG_ARR_TMP, level),
      *pv = ptr<T>(MG_ARR_E, level);
    const long long pl = g.plane;
    // rb: the "pair" is one red-black sweep -- first pass = red half-sweep, second pass = black half-sweep
    auto pass = [&](int which, const Geom &gs, const T *in, const T *rhs, T *out) {
        if (rb) launch_rb_fast<T>(stream_, gs, c, which, in, rhs, out);
        else launch_jacobi<T>(stream_, gs, c, om, in, rhs, out, false);
    };
    MG_TRY(exchange(MG_ARR_U, level));
    Geom glo = g; glo.nz = 2;
    pass(0, glo, px, pr, pv);
    Geom ghi = g; ghi.nz = 2; ghi.gz0 = g.gz0 + g.nz - 2;
    const long long ohi = (long long)(g.nz - 2) * pl;
    pass(0, ghi, px + ohi, pr + ohi, pv + ohi);
    MG_TRY(exchange_begin(MG_ARR_E, level));
    Geom gb = g; gb.nz = g.nz - 2; gb.gz0 = g.gz0 + 1;
    if (rb) launch_rb_fused<T>(stream_, gb, c, px + pl, pr + pl, pt + pl, (const T *)nullptr, gb);
    else launch_jacobi2<T>(stream_, gb, c, om, px + pl, pr + pl, pt + pl, false);
    MG_TRY(exchange_end());
    Geom g0 = g; g0.nz = 1;
    pass(1, g0, pv, pr, pt);
    Geom g1 = g; g1.nz = 1; g1.gz0 = g.gz0 + g.nz - 1;
    const long long o1 = (long long)(g.nz - 1) * pl;
    pass(1, g1, pv + o1, pr + o1, pt + o1);
    MG_HIP(hipGetLastError());
    std::swap(L.base[MG_ARR_U], L.base[MG_ARR_TMP]);
    return MG_OK;
}

// Every decision that changes WHICH messages a rank posts must come out the same on all ranks: the gates that look at
// the slab's thickness are evaluated on the thinnest slab of the level, not on the local one (the last rank owns one
// plane more, uneven splits differ by a coarse cell).
static Geom slab_gate_geom(const Level &L)
{
    Geom g = L.g;
    g.nz = L.nz_min;
    return g;
}

static bool depth2_enabled()
{
    static const bool e = [] { const char *v = getenv("MG_DEPTH2"); return !(v && v[0] == '0'); }();
    return e;
}

int Solver::refresh_rhs_halo(int level)
{
    Level &L = lv_[level];
    if (!L.dist || L.rhs_halo_ok) return MG_OK;
    MG_TRY(exchange(MG_ARR_RHS, level, 1));
    L.rhs_halo_ok = true;
    return MG_OK;
}

// The same pair (or red-black sweep) with TWO ghost planes: one exchange of two planes of u, then the fused kernel on the
// whole slab -- its first sweep is evaluated on the ghost planes -1 and nz as well (redundantly with the neighbour, from
// u on planes -2 .. nz+1 and the neighbour's rhs plane, which is exchanged once per right-hand side). One message pair
// per neighbour and pair instead of two, three launches instead of five; the interior output planes 2 .. nz-3 need no
// ghost plane at all and run while the halo moves. Same arithmetic per point => same bits as one GPU.
template <typename T>
// u_halo_ok: U's two ghost planes either side already hold the neighbours' current planes (the residual + restriction of this
// cycle fetched them and nothing has written U since -- true for the folding pair, which reads the uncorrected u): no exchange,
// and with nothing to hide behind an interior launch the whole slab is ONE launch.
int Solver::pair_on_slab2_t(int level, bool rb, int corr_level, bool u_halo_ok, double *norm_partials, int *norm_np)
{
    Level &L = lv_[level];
    const Geom &g = L.g;
    Coef<T> c = coef_of<T>(L);
    const T om = (T)d_.omega;
    T *px = ptr<T>(MG_ARR_U, level), *pr = ptr<T>(MG_ARR_RHS, level), *pt = ptr<T>(MG_ARR_TMP, level);
    const long long pl = g.plane;
    // dup > 0: the same piece once more, `dup` planes further up, in the same launch
    // corr_level >= 0 (Jacobi): the pair also applies the coarse correction, out = J(J(u + P e)): e's two ghost planes either
    // side come first (the separate prolongation fetched one), then every piece is the folding kernel
    const T *pe = corr_level >= 0 ? ptr<T>(MG_ARR_U, corr_level) : (const T *)nullptr;
    if (corr_level >= 0 && lv_[corr_level].dist) MG_TRY(exchange(MG_ARR_U, corr_level, 2));   // (a replicated level is whole on every rank)
    auto fused = [&](const Geom &gs, long long off, int dup = 0, hipStream_t st = nullptr) {
        if (!st) st = stream_;
        if (pe) launch_jacobi2_corr<T>(st, gs, lv_[corr_level].g, c, om, px + off, pe, pr + off, pt + off, dup);
        else if (rb) launch_rb_fused<T>(st, gs, c, px + off, pr + off, pt + off, (const T *)nullptr, gs, dup);
        else launch_jacobi2<T>(st, gs, c, om, px + off, pr + off, pt + off, false, dup);
    };
    static const bool one_boundary_launch = [] { const char *e = getenv("MG_MERGE_BOUNDARY"); return !(e && e[0] == '0'); }();
    MG_TRY(refresh_rhs_halo(level));
    static const bool reuse_halo = [] { const char *e = getenv("MG_REUSE_HALO"); return !(e && e[0] == '0'); }();
    if (norm_partials && !pe) {
        // Solver::solve's speculative pair: the whole slab in ONE wide-tile launch that also sums (rhs - A u)^2 of its input over
        // the owned planes (no interior / boundary split: the thin boundary pieces have no norm variant; the separate residual
        // norm this replaces needed the same exchange and a pass over the slab of its own)
        MG_TRY(exchange(MG_ARR_U, level, 2));
        const int np = rb ? launch_rb_fused<T>(stream_, g, c, px, pr, pt, (const T *)nullptr, g, 0, false, norm_partials)
                          : launch_jacobi2<T>(stream_, g, c, om, px, pr, pt, false, 0, norm_partials);
        if (norm_np) *norm_np = np;
    } else if (u_halo_ok && reuse_halo) {
        fused(g, 0);
    } else if (!L.overlap || g.nz < 8) {
        MG_TRY(exchange(MG_ARR_U, level, 2));
        fused(g, 0);
    } else {
        const bool on_comm = one_boundary_launch && boundary_on_comm_stream();
        MG_TRY(exchange_begin(MG_ARR_U, level, 2, !on_comm));
        Geom gi = g; gi.nz = g.nz - 4; gi.gz0 = g.gz0 + 2;     // reads u on planes 0 .. nz-1 only
        Geom glo = g; glo.nz = 2;
        if (on_comm) {                       // the boundary planes follow the halo on its own stream, beside the interior launch
            fused(glo, 0, g.nz - 2, comm_stream_);
            MG_TRY(halo_work_done());
            pair_on_comm_level_ = level;     // the planes the next exchange sends were written on the communication stream
        }
        fused(gi, 2 * pl);
        MG_TRY(exchange_end());
        if (on_comm) {
        } else if (one_boundary_launch) {    // output planes 0, 1 and nz-2, nz-1 in ONE launch (two were 2 x 15 us at 513^2)
            fused(glo, 0, g.nz - 2);
        } else {
            fused(glo, 0);
            Geom ghi = g; ghi.nz = 2; ghi.gz0 = g.gz0 + g.nz - 2;
            fused(ghi, (long long)(g.nz - 2) * pl);
        }
    }
    MG_HIP(hipGetLastError());
    std::swap(L.base[MG_ARR_U], L.base[MG_ARR_TMP]);
    return MG_OK;
}

// Residual + full weighting of a distributed level in one kernel: coarse = R (rhs - A u) for the coarse planes that coincide
// with this rank's fine planes. The lowest of them needs the residual on the ghost plane -1, i.e. u on planes -2 .. 0 and
// the neighbour's rhs plane: one exchange of two planes of u (none of the residual, which is never stored).
template <typename T>
int Solver::resid_restrict_on_slab_t(int level, const Geom &gc, T *coarse_rhs)
{
    Level &L = lv_[level];
    const Geom &gf = L.g;
    const Coef<T> c = coef_of<T>(L);
    T *pu = ptr<T>(MG_ARR_U, level), *pr = ptr<T>(MG_ARR_RHS, level);
    MG_TRY(refresh_rhs_halo(level));
    const int pair_level = pair_on_comm_level_;
    pair_on_comm_level_ = -1;
    const bool semi = gf.gnz == gc.gnz;          // planes map one to one
    if (!L.overlap || gc.nz < 4 || semi) {
        MG_TRY(exchange(MG_ARR_U, level, 2));
        launch_resid_restrict_fw<T>(stream_, gf, gc, c, pu, pr, coarse_rhs);
    } else {
        // coarse planes 1 .. nzc-2 read u on owned planes only: they run while the halo moves
        static const bool one_boundary_launch = [] { const char *e = getenv("MG_MERGE_BOUNDARY"); return !(e && e[0] == '0'); }();
        const bool on_comm = one_boundary_launch && boundary_on_comm_stream();
        static const bool early = [] { const char *e = getenv("MG_EARLY_EXCHANGE"); return !(e && e[0] == '0'); }();
        if (on_comm && early && pair_level == level) {
            // The planes to send (0, 1, nz-2, nz-1 of the pair's output) were written by the pair's boundary piece on the
            // communication stream itself: the exchange starts at once, while the pair's interior launch is still running on
            // the main stream; only the boundary piece below needs that launch's planes.
            if (2 > L.gh || 2 > L.nz_min) { set_last_error("halo exchange deeper than the ghost planes / the thinnest slab"); return MG_ERR_BAD_ARG; }
            P2POp ops[4];
            const int n = halo_ops(MG_ARR_U, level, 2, ops);
            int rc = post(ops, n, comm_stream_);
            if (rc) { set_last_error("halo exchange failed"); return rc; }
            MG_HIP(hipEventRecord(ev_ready_, stream_));
            MG_HIP(hipStreamWaitEvent(comm_stream_, ev_ready_, 0));
        } else {
            MG_TRY(exchange_begin(MG_ARR_U, level, 2, !on_comm));
        }
        Geom gc0 = gc; gc0.nz = 1;
        Geom gf0 = gf; gf0.nz = 2;
        const int kl = gc.nz - 1;                 // last coarse plane: fine planes 2 kl (and 2 kl + 1 unless it is the grid's top plane)
        if (on_comm) {                            // first and last coarse plane behind the halo on its own stream
            launch_resid_restrict_fw<T>(comm_stream_, gf0, gc0, c, pu, pr, coarse_rhs, kl, gf.nz - 2 * kl);
            MG_TRY(halo_work_done());
        }
        Geom gci = gc; gci.nz = gc.nz - 2; gci.gz0 = gc.gz0 + 1;
        Geom gfi = gf; gfi.nz = 2 * gci.nz; gfi.gz0 = gf.gz0 + 2;
        launch_resid_restrict_fw<T>(stream_, gfi, gci, c, pu + 2 * gf.plane, pr + 2 * gf.plane, coarse_rhs + gc.plane);
        MG_TRY(exchange_end());
        if (on_comm) {
        } else if (one_boundary_launch) {         // first and last coarse plane in ONE launch
            launch_resid_restrict_fw<T>(stream_, gf0, gc0, c, pu, pr, coarse_rhs, kl, gf.nz - 2 * kl);
        } else {
            launch_resid_restrict_fw<T>(stream_, gf0, gc0, c, pu, pr, coarse_rhs);
            Geom gc1 = gc; gc1.nz = 1; gc1.gz0 = gc.gz0 + kl;
            Geom gf1 = gf; gf1.nz = gf.nz - 2 * kl; gf1.gz0 = gf.gz0 + 2 * kl;
            launch_resid_restrict_fw<T>(stream_, gf1, gc1, c, pu + (long long)2 * kl * gf.plane, pr + (long long)2 * kl * gf.plane,
                                        coarse_rhs + (long long)kl * gc.plane);
        }
    }
    MG_HIP(hipGetLastError());
    return MG_OK;
}

// corr_level >= 0: x is still missing the coarse-grid correction P u_{corr_level}; the first fused
// pair applies it on the fly (caller checked can_fold_prolong)
// e_scratch: the level's E array may be used as scratch (true only inside the V-cycle, where E is idle): the
// fused pair on a z-slab keeps its boundary planes' first sweep there. The public mg_smooth never passes it, so a
// caller's E array is left alone (distributed levels then take exchanged single sweeps).
template <typename T>
int Solver::smooth_t(int level, int smoother, int sweeps, int ax, int ar, bool x_zero, int corr_level, bool e_scratch, bool u_halo_ok)
{
    Level &L = lv_[level];
    pair_on_comm_level_ = -1;
    Coef<T> c = coef_of<T>(L);
    const bool prof = profiling_ && level == 0 && sweeps > 0 && smoother != MG_SMOOTH_GS_LEX;
    int launches = 0;  // kernel launches of this call (a fused pair is one)
    if (prof) MG_TRY(prof_begin(level));
    switch (smoother) {
    case MG_SMOOTH_JACOBI:
        for (int s = 0; s < sweeps; s++) {
            pair_on_comm_level_ = -1;
            if (L.dist && depth2_enabled() && ax == MG_ARR_U && ar == MG_ARR_RHS && s + 1 < sweeps && jacobi2_slab_ok<T>(slab_gate_geom(L))) {
                if (x_zero && s == 0) {  // zero guess on every rank: no halo of u to fetch at all, only the neighbours' rhs planes
                    MG_TRY(refresh_rhs_halo(level));
                    launch_jacobi2<T>(stream_, L.g, c, (T)d_.omega, ptr<T>(ax, level), ptr<T>(ar, level), ptr<T>(MG_ARR_TMP, level), true);
                    std::swap(L.base[ax], L.base[MG_ARR_TMP]);
                } else {
                    const bool norm = want_pair_norm_ && level == 0 && s == 0 && corr_level < 0;
                    int np = 0;
                    MG_TRY(pair_on_slab2_t<T>(level, false, s == 0 ? corr_level : -1, s == 0 && u_halo_ok, norm ? d_partials_ : (double *)nullptr, &np));
                    if (norm && np > 0) {   // sum r^2 of the pair's input over this rank's planes, then over the ranks -> d_scal_[0]
                        launch_reduce_final(stream_, d_partials_, np, d_scal_);
                        MG_TRY(allreduce(d_scal_, 1));
                        pair_norm_done_ = true;
                    }
                }
                s++; launches += 1;   // counted as ONE segment: exchange + interior + boundary launches
                continue;
            }
            if (L.dist && L.overlap && e_scratch && ax == MG_ARR_U && ar == MG_ARR_RHS && s + 1 < sweeps &&
                !(x_zero && s == 0) && jacobi2_slab_ok<T>(slab_gate_geom(L))) {  // E is free in a V-cycle: scratch for the boundary planes' first sweep
                MG_TRY(pair_on_slab_t<T>(level, false));
                s++; launches += 1;
                continue;
            }
            if (!L.dist && s + 1 < sweeps && jacobi2_ok<T>(L.g)) {
                // two sweeps in one pass over HBM; the pair lands in TMP like a single sweep would
                if (s == 0 && corr_level >= 0)
                    launch_jacobi2_corr<T>(stream_, L.g, lv_[corr_level].g, c, (T)d_.omega, ptr<T>(ax, level),
                                           ptr<T>(ax, corr_level), ptr<T>(ar, level), ptr<T>(MG_ARR_TMP, level));
                else {  // x_zero: the pair starts from an implicit zero guess (nothing is read for x)
                    // The variant of the wide-tile pair that also sums (rhs - A u)^2 costs 1.5 % over the plain one (0.636 against 0.626 ms
                    // at 513^3, alternating bench runs on one box): level 0's pre-smoothing pair takes it when the outer loop asks
                    // for the norm (MG_ALWAYS_NORM=1: always, as in the first half of round 3 when it was the faster of the two).
                    static const bool always_norm = [] { const char *e = getenv("MG_ALWAYS_NORM"); return e && e[0] == '1'; }();
                    const bool norm = (want_pair_norm_ || always_norm) && level == 0 && s == 0 && !x_zero && ax == MG_ARR_U && ar == MG_ARR_RHS;
                    const int np = launch_jacobi2<T>(stream_, L.g, c, (T)d_.omega, ptr<T>(ax, level), ptr<T>(ar, level),
                                                     ptr<T>(MG_ARR_TMP, level), x_zero && s == 0, 0, norm ? d_partials_ : (double *)nullptr);
                    if (norm && np > 0 && want_pair_norm_) {   // sum r^2 of the pair's input -> d_scal_[0]
                        launch_reduce_final(stream_, d_partials_, np, d_scal_);
                        pair_norm_done_ = true;
                    }
                }
                std::swap(L.base[ax], L.base[MG_ARR_TMP]);
                s++; launches++;
                continue;
            }
            launches++;
            const bool zero_now = x_zero && s == 0;
            T *px = ptr<T>(ax, level), *pr = ptr<T>(ar, level), *pt = ptr<T>(MG_ARR_TMP, level);
            if (zero_now) {
                launch_jacobi<T>(stream_, L.g, c, (T)d_.omega, px, pr, pt, true);
            } else {
                MG_TRY(overlapped(level, ax, [&](const Geom &gs, long long off) {
                    launch_jacobi<T>(stream_, gs, c, (T)d_.omega, px + off, pr + off, pt + off, false);
                }));
            }
            // the reference swaps the std::vector buffers (solvers.hpp:82); so do we
            std::swap(L.base[ax], L.base[MG_ARR_TMP]);
        }
        break;
    case MG_SMOOTH_RBGS:
        for (int s = 0; s < sweeps; s++) {
            pair_on_comm_level_ = -1;
            if (L.dist && depth2_enabled() && ax == MG_ARR_U && ar == MG_ARR_RHS && jacobi2_slab_ok<T>(slab_gate_geom(L)) && rb_slab_enabled()) {
                const bool norm = want_pair_norm_ && level == 0 && s == 0 && corr_level < 0;
                int np = 0;
                MG_TRY(pair_on_slab2_t<T>(level, true, -1, false, norm ? d_partials_ : (double *)nullptr, &np));   // one-pass red-black sweep on the whole slab, two ghost planes
                if (norm && np > 0) {
                    launch_reduce_final(stream_, d_partials_, np, d_scal_);
                    MG_TRY(allreduce(d_scal_, 1));
                    pair_norm_done_ = true;
                }
                launches += 1;
                continue;
            }
            if (L.dist && L.overlap && e_scratch && ax == MG_ARR_U && ar == MG_ARR_RHS &&
                jacobi2_slab_ok<T>(slab_gate_geom(L)) && rb_slab_enabled()) {  // one-pass red-black sweep on the slab's inner planes
                MG_TRY(pair_on_slab_t<T>(level, true));
                launches += 1;
                continue;
            }
            if (!L.dist && rb_fused_ok<T>(L.g)) {  // both colours in one pass over HBM; the sweep lands in TMP
                const bool corr = (s == 0 && corr_level >= 0);
                // (Solver::solve: the first pre-smoothing sweep of level 0 also sums (rhs - A u)^2 of its input)
                const bool norm = want_pair_norm_ && level == 0 && s == 0 && !corr && !x_zero && ax == MG_ARR_U && ar == MG_ARR_RHS;
                const int np = launch_rb_fused<T>(stream_, L.g, c, ptr<T>(ax, level), ptr<T>(ar, level), ptr<T>(MG_ARR_TMP, level),
                                                  corr ? ptr<T>(ax, corr_level) : (const T *)nullptr, lv_[corr ? corr_level : level].g, 0,
                                                  x_zero && s == 0, norm ? d_partials_ : (double *)nullptr);
                if (norm && np > 0) {
                    launch_reduce_final(stream_, d_partials_, np, d_scal_);
                    pair_norm_done_ = true;
                }
                std::swap(L.base[ax], L.base[MG_ARR_TMP]);
                launches++;
                continue;
            }
            launches += 2;
            if (fast_path_ok<T>(L.g)) {  // vectorised, out of place: red x -> tmp, black tmp -> x
                T *px = ptr<T>(ax, level), *pr = ptr<T>(ar, level), *pt = ptr<T>(MG_ARR_TMP, level);
                MG_TRY(overlapped(level, ax, [&](const Geom &gs, long long off) {
                    launch_rb_fast<T>(stream_, gs, c, 0, px + off, pr + off, pt + off);
                }));
                MG_TRY(overlapped(level, MG_ARR_TMP, [&](const Geom &gs, long long off) {
                    launch_rb_fast<T>(stream_, gs, c, 1, pt + off, pr + off, px + off);
                }));
                continue;
            }
            MG_TRY(exchange(ax, level));
            launch_rbgs_colour<T>(stream_, L.g, c, 0, ptr<T>(ax, level), ptr<T>(ar, level));
            MG_TRY(exchange(ax, level));
            launch_rbgs_colour<T>(stream_, L.g, c, 1, ptr<T>(ax, level), ptr<T>(ar, level));
        }
        break;
    case MG_SMOOTH_ZEBRA_Y:
    case MG_SMOOTH_ZEBRA_X:
        if (sweeps > 0 && (!L.zebra || smoother != d_.smoother)) {
            set_last_error("zebra line smoother: the handle was not created with this smoother (its line factors are tabulated at creation)");
            return MG_ERR_BAD_ARG;
        }
        launches += 2 * sweeps;
        for (int s = 0; s < sweeps; s++)
            for (int colour = 0; colour < 2; colour++) {
                MG_TRY(exchange(ax, level));  // the other colour's ghost planes (z-slabs; lines run along y)
                if (smoother == MG_SMOOTH_ZEBRA_X)
                    launch_zebra_x<T>(stream_, L.g, c, colour, ptr<T>(ax, level), ptr<T>(ar, level), ptr<T>(MG_ARR_TMP, level),
                                      static_cast<const T *>(L.zebra));
                else
                    launch_zebra_y<T>(stream_, L.g, c, colour, ptr<T>(ax, level), ptr<T>(ar, level), ptr<T>(MG_ARR_TMP, level),
                                      static_cast<const T *>(L.zebra));
            }
        break;
    default:
        if (L.dist && sweeps > 0) {
            set_last_error("lexicographic Gauss-Seidel is sequential across slabs: not available on a distributed level");
            return MG_ERR_BAD_ARG;
        }
        if (sweeps > 0) launch_gs_lex<T>(stream_, L.g, c, sweeps, ptr<T>(ax, level), ptr<T>(ar, level));
        break;
    }
    if (prof) MG_TRY(prof_end(level, corr_level >= 0 ? MG_PROF_SMOOTH_PROLONG : MG_PROF_SMOOTH, sweeps, launches));
    MG_HIP(hipGetLastError());
    return MG_OK;
}

int Solver::smooth(int level, int smoother, int sweeps, int arr_x, int arr_rhs)
{
    if (!check_arr(arr_x, level, "mg_smooth") || !check_arr(arr_rhs, level, "mg_smooth")) return MG_ERR_BAD_ARG;
    if (arr_x == MG_ARR_TMP || arr_rhs == MG_ARR_TMP || arr_x == arr_rhs || sweeps < 0 ||
        smoother < MG_SMOOTH_GS_LEX || smoother > MG_SMOOTH_ZEBRA_X) {
        set_last_error("mg_smooth: bad array / smoother / sweeps");
        return MG_ERR_BAD_ARG;
    }
    MG_HIP(hipSetDevice(device_));
    if (arr_x == MG_ARR_RHS) lv_[level].rhs_halo_ok = false;
    return d_.dtype == MG_F64 ? smooth_t<double>(level, smoother, sweeps, arr_x, arr_rhs)
                              : smooth_t<float>(level, smoother, sweeps, arr_x, arr_rhs);
}

template <typename T>
int Solver::residual_t(int level, int ax, int ar, int arr_r, bool want_norm)
{
    Level &L = lv_[level];
    if (!want_norm && arr_r >= 0) {  // vector only: the exchange hides behind the interior planes
        T *px = ptr<T>(ax, level), *pr = ptr<T>(ar, level), *po = ptr<T>(arr_r, level);
        Coef<T> c = coef_of<T>(L);
        MG_TRY(overlapped(level, ax, [&](const Geom &gs, long long off) {
            launch_residual<T>(stream_, gs, c, px + off, pr + off, po + off, d_partials_, (double *)nullptr);
        }));
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    MG_TRY(exchange(ax, level));
    launch_residual<T>(stream_, L.g, coef_of<T>(L), ptr<T>(ax, level), ptr<T>(ar, level),
                       arr_r >= 0 ? ptr<T>(arr_r, level) : (T *)nullptr, d_partials_,
                       want_norm ? d_scal_ : (double *)nullptr);
    MG_HIP(hipGetLastError());
    if (want_norm && L.dist) MG_TRY(allreduce(d_scal_, 1));
    return MG_OK;
}

int Solver::residual(int level, int arr_x, int arr_rhs, int arr_r, double *sumsq_out)
{
    if (!check_arr(arr_x, level, "mg_residual") || !check_arr(arr_rhs, level, "mg_residual")) return MG_ERR_BAD_ARG;
    if (arr_r >= 0 && (!check_arr(arr_r, level, "mg_residual") || arr_r == arr_x || arr_r == arr_rhs)) {
        set_last_error("mg_residual: bad output array");
        return MG_ERR_BAD_ARG;
    }
    MG_HIP(hipSetDevice(device_));
    if (arr_r == MG_ARR_RHS) lv_[level].rhs_halo_ok = false;
    int rc = d_.dtype == MG_F64 ? residual_t<double>(level, arr_x, arr_rhs, arr_r, true)
                                : residual_t<float>(level, arr_x, arr_rhs, arr_r, true);
    if (rc) return rc;
    if (sumsq_out) {
        MG_HIP(hipMemcpyAsync(h_scal_, d_scal_, sizeof(double), hipMemcpyDeviceToHost, stream_));
        MG_HIP(hipStreamSynchronize(stream_));
        *sumsq_out = h_scal_[0];
    }
    return MG_OK;
}

template <typename T>
int Solver::sumsq_t(int level, int arr)
{
    launch_sumsq<T>(stream_, lv_[level].g, ptr<T>(arr, level), d_partials_, d_scal_ + 1);
    MG_HIP(hipGetLastError());
    if (lv_[level].dist) MG_TRY(allreduce(d_scal_ + 1, 1));
    return MG_OK;
}

int Solver::sumsq(int level, int arr, double *out)
{
    if (!out || !check_arr(arr, level, "mg_sumsq")) return MG_ERR_BAD_ARG;
    MG_HIP(hipSetDevice(device_));
    int rc = d_.dtype == MG_F64 ? sumsq_t<double>(level, arr) : sumsq_t<float>(level, arr);
    if (rc) return rc;
    MG_HIP(hipMemcpyAsync(h_scal_ + 1, d_scal_ + 1, sizeof(double), hipMemcpyDeviceToHost, stream_));
    MG_HIP(hipStreamSynchronize(stream_));
    *out = h_scal_[1];
    return MG_OK;
}

template <typename T>
int Solver::restrict_t(int fl, int kind, int as, int ad)
{
    if (lv_[fl].dist != lv_[fl + 1].dist) {
        set_last_error("transfer across the gather level is only available inside mg_cycle");
        return MG_ERR_BAD_ARG;
    }
    if (ad == MG_ARR_RHS) lv_[fl + 1].rhs_halo_ok = false;
    if (kind == MG_RESTRICT_FULLW) MG_TRY(exchange(as, fl));  // needs r on the lower ghost plane
    if (kind == MG_RESTRICT_FULLW)
        launch_restrict_fw<T>(stream_, lv_[fl].g, lv_[fl + 1].g, ptr<T>(as, fl), ptr<T>(ad, fl + 1));
    else
        launch_inject<T>(stream_, lv_[fl].g, lv_[fl + 1].g, ptr<T>(as, fl), ptr<T>(ad, fl + 1));
    MG_HIP(hipGetLastError());
    return MG_OK;
}

int Solver::restrict_to(int fine_level, int kind, int arr_src, int arr_dst)
{
    if (fine_level + 1 >= d_.levels || !check_arr(arr_src, fine_level, "mg_restrict") ||
        !check_arr(arr_dst, fine_level + 1, "mg_restrict")) {
        set_last_error("mg_restrict: bad level / array");
        return MG_ERR_BAD_ARG;
    }
    MG_HIP(hipSetDevice(device_));
    return d_.dtype == MG_F64 ? restrict_t<double>(fine_level, kind, arr_src, arr_dst)
                              : restrict_t<float>(fine_level, kind, arr_src, arr_dst);
}

template <typename T>
int Solver::prolong_t(int cl, int add, int as, int ad)
{
    pair_on_comm_level_ = -1;   // this call writes a level's arrays on the main stream: no exchange may skip its wait for that stream (every writer says so)
    if (lv_[cl].dist != lv_[cl - 1].dist) {
        set_last_error("transfer across the gather level is only available inside mg_cycle");
        return MG_ERR_BAD_ARG;
    }
    if (ad == MG_ARR_RHS) lv_[cl - 1].rhs_halo_ok = false;
    MG_TRY(exchange(as, cl));  // odd fine planes read the coarse upper ghost plane
    launch_prolong<T>(stream_, lv_[cl].g, lv_[cl - 1].g, ptr<T>(as, cl), ptr<T>(ad, cl - 1), add != 0);
    MG_HIP(hipGetLastError());
    return MG_OK;
}

int Solver::prolong(int coarse_level, int add, int arr_src, int arr_dst)
{
    if (coarse_level < 1 || !check_arr(arr_src, coarse_level, "mg_prolong") ||
        !check_arr(arr_dst, coarse_level - 1, "mg_prolong")) {
        set_last_error("mg_prolong: bad level / array");
        return MG_ERR_BAD_ARG;
    }
    MG_HIP(hipSetDevice(device_));
    return d_.dtype == MG_F64 ? prolong_t<double>(coarse_level, add, arr_src, arr_dst)
                              : prolong_t<float>(coarse_level, add, arr_src, arr_dst);
}

template <typename T>
int Solver::correct_t(int level, int au, int ae)
{
    pair_on_comm_level_ = -1;   // this call writes a level's arrays on the main stream: no exchange may skip its wait for that stream (every writer says so)
    launch_correct<T>(stream_, lv_[level].g, ptr<T>(au, level), ptr<T>(ae, level));
    MG_HIP(hipGetLastError());
    return MG_OK;
}

int Solver::correct(int arr_u, int arr_e)
{
    if (!check_arr(arr_u, 0, "mg_correct") || !check_arr(arr_e, 0, "mg_correct") || arr_u == arr_e)
        return MG_ERR_BAD_ARG;
    if (arr_u == MG_ARR_RHS || arr_e == MG_ARR_RHS) lv_[0].rhs_halo_ok = false;
    MG_HIP(hipSetDevice(device_));
    return d_.dtype == MG_F64 ? correct_t<double>(0, arr_u, arr_e) : correct_t<float>(0, arr_u, arr_e);
}

template <typename T>
int Solver::coarse_ex_t(int level, int ax, int ar, int smoother, int maxit, double tol, int fixed, bool x_zero)
{
    pair_on_comm_level_ = -1;   // this call writes a level's arrays on the main stream: no exchange may skip its wait for that stream (every writer says so)
    Level &L = lv_[level];
    launch_coarse_solve<T>(stream_, L.g, coef_of<T>(L), (T)d_.omega, smoother, ptr<T>(ax, level),
                           ptr<T>(MG_ARR_TMP, level), ptr<T>(ar, level), maxit, tol, fixed, d_coarse_, x_zero);
    MG_HIP(hipGetLastError());
    return MG_OK;
}

template <typename T>
int Solver::coarse_t(int level, int ax, int ar, bool x_zero)
{
    // the coarsest-grid solver of a zebra hierarchy smooths with red-black Gauss-Seidel (mg_desc.h)
    const int sm = coarse_smoother_of(d_.smoother);
    if (lock_iters_ >= 0) return coarse_ex_t<T>(level, ax, ar, sm, lock_iters_, d_.coarse_tol, 1, x_zero);
    return coarse_ex_t<T>(level, ax, ar, sm, d_.coarse_maxit, d_.coarse_tol,
                          d_.coarse_mode == MG_COARSE_FIXED ? 1 : 0, x_zero);
}

int Solver::coarse_solve(int level, int arr_x, int arr_rhs, mg_cycle_stats *st)
{
    return coarse_solve_ex(level, arr_x, arr_rhs, coarse_smoother_of(d_.smoother),
                           d_.coarse_maxit, d_.coarse_tol, d_.coarse_mode == MG_COARSE_FIXED ? 1 : 0, st);
}

int Solver::coarse_solve_ex(int level, int arr_x, int arr_rhs, int smoother, int maxit, double tol, int fixed,
                            mg_cycle_stats *st)
{
    if (!check_arr(arr_x, level, "mg_coarse_solve") || !check_arr(arr_rhs, level, "mg_coarse_solve") ||
        arr_x == MG_ARR_TMP || arr_rhs == MG_ARR_TMP || arr_x == arr_rhs || maxit < 0 ||
        smoother < MG_SMOOTH_GS_LEX || smoother > MG_SMOOTH_RBGS) {
        set_last_error("mg_coarse_solve: bad array / smoother / maxit");
        return MG_ERR_BAD_ARG;
    }
    MG_HIP(hipSetDevice(device_));
    int rc = d_.dtype == MG_F64 ? coarse_ex_t<double>(level, arr_x, arr_rhs, smoother, maxit, tol, fixed)
                                : coarse_ex_t<float>(level, arr_x, arr_rhs, smoother, maxit, tol, fixed);
    if (rc) return rc;
    MG_HIP(hipMemcpyAsync(h_coarse_, d_coarse_, sizeof(CoarseOut), hipMemcpyDeviceToHost, stream_));
    MG_HIP(hipStreamSynchronize(stream_));
    if (st) {
        st->coarse_iters = h_coarse_->iters;
        st->coarse_flag = h_coarse_->flag;
        st->coarse_relres = h_coarse_->relres;
        st->fine_sumsq_r = 0;
    }
    return MG_OK;
}

// Coarse solve of a coarsest level that is still distributed (few levels, many ranks):
// gather its rhs on rank 0, solve there in the full-size copies, scatter the solution.
// full_[0] must already hold the gathered rhs.
template <typename T>
int Solver::coarse_full_t()
{
    if (rank_ != 0) return MG_OK;
    Level &L = lv_[T_];
    MG_HIP(hipMemsetAsync(full_[1], 0, (size_t)(gfull_.nz + 2) * (size_t)gfull_.plane * esize(), stream_));
    launch_coarse_solve<T>(stream_, gfull_, coef_of<T>(L), (T)d_.omega,
                           coarse_smoother_of(d_.smoother), fullptr<T>(1), fullptr<T>(2),
                           fullptr<T>(0), lock_iters_ >= 0 ? lock_iters_ : d_.coarse_maxit, d_.coarse_tol,
                           (lock_iters_ >= 0 || d_.coarse_mode == MG_COARSE_FIXED) ? 1 : 0, d_coarse_);
    MG_HIP(hipGetLastError());
    return MG_OK;
}

// x_zero: the initial guess is zero and the array has NOT been cleared (the LDS Jacobi solver takes that as a flag)
template <typename T>
int Solver::coarse_level_t(int l, int ax, int ar, bool x_zero)
{
    Level &L = lv_[l];
    if (!L.present) return MG_OK;  // this rank holds nothing of the level (gathered on rank 0)
    const long long pts = (long long)L.g.nx * L.g.ny * L.g.gnz;
    const bool big = pts > 32768;  // e.g. the 17 x 17 x 513 coarsest grid of a semi-coarsened hierarchy
    if (x_zero && ((big && d_.coarse_mode == MG_COARSE_FIXED) || L.dist)) { MG_TRY(zero_array(ax, l)); x_zero = false; }
    if (big && d_.coarse_mode == MG_COARSE_FIXED) {
        MG_TRY(smooth_t<T>(l, coarse_smoother_of(d_.smoother), d_.coarse_maxit, ax, ar, false, -1,
                           d_.cycle == MG_CYCLE_V));
        h_fixed_->iters = d_.coarse_maxit; h_fixed_->flag = 0;
        h_fixed_->relres = 0; h_fixed_->sumsq_rhs = 0; h_fixed_->sumsq_r = 0;  // not evaluated on this path
        MG_HIP(hipMemcpyAsync(d_coarse_, h_fixed_, sizeof(CoarseOut), hipMemcpyHostToDevice, stream_));
        return MG_OK;
    }
    if (L.dist) {  // small but distributed (few levels, many ranks): gather, solve on rank 0, scatter
        MG_TRY(gather_T(ar, 0));
        MG_TRY(coarse_full_t<T>());
        MG_TRY(scatter_T(1, ax));
        return MG_OK;
    }
    return coarse_t<T>(l, ax, ar, x_zero);
}

// Standard V(nu_pre, nu_post) (extension, BASELINE configs 2-4). Slab-decomposed runs: levels
// 0..T_ are distributed (halo exchanges happen inside smooth_t / residual_t / restrict_t /
// prolong_t), the residual of level T_ is gathered on rank 0, which runs the deeper levels
// alone and scatters the prolonged correction back.
template <typename T>
int Solver::vcycle_rec_t(int l, bool u_zero)
{
    const int L = d_.levels;
    const bool mine = lv_[l].present;
    bool fold = false;  // prolong-add folded into the post-smoothing pair
    bool fold_slab = false;  // ... on the pieces of a slab: u itself stays as the residual + restriction saw it
    if (l == L - 1) return coarse_level_t<T>(l, MG_ARR_U, MG_ARR_RHS, u_zero);
    // fused residual + full weighting when both levels live whole on this rank
    const bool fuse_rr = mine && d_.restriction == MG_RESTRICT_FULLW && !lv_[l].dist && lv_[l + 1].present &&
                         resid_restrict_fast_ok<T>(lv_[l].g, lv_[l + 1].g);
    // distributed level: the coarse slab is the next level's, or the staging slab when the next level is gathered
    const Geom &gc_slab = lv_[l + 1].dist ? lv_[l + 1].g : stage_g_;
    const bool fuse_rr_slab = mine && lv_[l].dist && d_.restriction == MG_RESTRICT_FULLW && depth2_enabled() &&
                              resid_restrict_slab_ok<T>(lv_[l].g, gc_slab);
    const bool prof = profiling_ && l == 0 && mine;
    // launch-bound levels (65^3 and below; rows too narrow for the fused pair): V(2,2) Jacobi with full weighting runs as ONE
    // launch either side of the coarser levels -- J(J(0)) + residual + restriction, and J(J(u + P e)) (mg_small_levels.hip)
    // levels up to this size take the brick kernels even where the row-wide fused pair would run (MG_SMALL_MAX_N)
    static const int small_max_n = [] { const char *e = getenv("MG_SMALL_MAX_N"); return e ? atoi(e) : 0; }();
    const bool small = mine && !prof && !stage_fn_ && !lv_[l].dist && lv_[l + 1].present && !lv_[l + 1].dist &&
                       d_.smoother == MG_SMOOTH_JACOBI && d_.nu_pre == 2 && d_.nu_post == 2 && d_.restriction == MG_RESTRICT_FULLW &&
                       (!jacobi2_ok<T>(lv_[l].g) || lv_[l].g.nx <= small_max_n) && small_fused_ok<T>(lv_[l].g, lv_[l + 1].g);
    const bool small_pre = small && u_zero;   // the zero guess is part of the fused kernel's contract
    if (small_pre) {
        lv_[l + 1].rhs_halo_ok = false;
        launch_small_pre_rr<T>(stream_, lv_[l].g, lv_[l + 1].g, coef_of<T>(lv_[l]), (T)d_.omega, ptr<T>(MG_ARR_RHS, l),
                               ptr<T>(MG_ARR_U, l), ptr<T>(MG_ARR_RHS, l + 1));
        MG_HIP(hipGetLastError());
        const bool skip0 = can_skip_zeroing<T>(l + 1) || l + 1 == L - 1;   // the coarsest-grid solver takes the zero guess as a flag
        if (!skip0) MG_TRY(zero_array(MG_ARR_U, l + 1));
        MG_TRY(vcycle_rec_t<T>(l + 1, skip0));
    } else if (mine) {
        if (l == 0 && fine_pre_done_ > 0) {   // Solver::solve ran (the first sweeps of) this level's pre-smoothing with the norm
            const int left = d_.nu_pre - fine_pre_done_;
            fine_pre_done_ = 0;
            if (left > 0) MG_TRY(smooth_t<T>(l, d_.smoother, left, MG_ARR_U, MG_ARR_RHS, false, -1, true));
        } else MG_TRY(smooth_t<T>(l, d_.smoother, d_.nu_pre, MG_ARR_U, MG_ARR_RHS, u_zero, -1, true));
        if (prof) MG_TRY(prof_begin(l));
        if (!fuse_rr && !fuse_rr_slab) MG_TRY(residual_t<T>(l, MG_ARR_U, MG_ARR_RHS, MG_ARR_TMP, false));
    }
    if (small_pre) {
        // residual, restriction and the coarser levels are done
    } else if (lv_[l].dist && !lv_[l + 1].dist) {  // l == T_: restrict locally, gather the coarse rhs on rank 0
        if (fuse_rr_slab) {
            MG_TRY(resid_restrict_on_slab_t<T>(l, stage_g_, stageptr<T>(0)));
        } else if (d_.restriction == MG_RESTRICT_FULLW) {
            MG_TRY(exchange(MG_ARR_TMP, l));
            launch_restrict_fw<T>(stream_, lv_[l].g, stage_g_, ptr<T>(MG_ARR_TMP, l), stageptr<T>(0));
        } else {
            launch_inject<T>(stream_, lv_[l].g, stage_g_, ptr<T>(MG_ARR_TMP, l), stageptr<T>(0));
        }
        MG_HIP(hipGetLastError());
        if (prof) MG_TRY(prof_end(l, MG_PROF_RESID_RESTRICT, 1, fuse_rr_slab ? 1 : 2));
        MG_TRY(gather_S(MG_ARR_RHS));
        if (lv_[l + 1].present) {
            const bool skip0 = can_skip_zeroing<T>(l + 1) || l + 1 == L - 1;
            if (!skip0) MG_TRY(zero_array(MG_ARR_U, l + 1));
            MG_TRY(vcycle_rec_t<T>(l + 1, skip0));
        }
        if (fuse_rr_slab && can_fold_prolong_replicated<T>(l)) {
            fold = fold_slab = true;   // the post-smoothing pair reads the correction from the replicated level's own array
        } else {
            MG_TRY(scatter_S(MG_ARR_U));
            if (prof) MG_TRY(prof_begin(l));
            launch_prolong<T>(stream_, stage_g_, lv_[l].g, stageptr<T>(1), ptr<T>(MG_ARR_U, l), true);
            MG_HIP(hipGetLastError());
            if (prof) MG_TRY(prof_end(l, MG_PROF_PROLONG, 1, 1));
        }
    } else if (mine) {
        if (fuse_rr_slab) {
            lv_[l + 1].rhs_halo_ok = false;
            MG_TRY(resid_restrict_on_slab_t<T>(l, lv_[l + 1].g, ptr<T>(MG_ARR_RHS, l + 1)));
        } else if (fuse_rr) {
            launch_resid_restrict_fw<T>(stream_, lv_[l].g, lv_[l + 1].g, coef_of<T>(lv_[l]), ptr<T>(MG_ARR_U, l),
                                        ptr<T>(MG_ARR_RHS, l), ptr<T>(MG_ARR_RHS, l + 1));
            MG_HIP(hipGetLastError());
        } else {
            MG_TRY(restrict_t<T>(l, d_.restriction, MG_ARR_TMP, MG_ARR_RHS));
        }
        if (prof) MG_TRY(prof_end(l, MG_PROF_RESID_RESTRICT, 1, (fuse_rr || fuse_rr_slab) ? 1 : 2));   // the slab-fused form is ONE segment (exchange + interior + boundary launches), like the pair
        const bool skip0 = can_skip_zeroing<T>(l + 1) || l + 1 == L - 1;   // the coarsest-grid solver takes the zero guess as a flag
        if (!skip0) MG_TRY(zero_array(MG_ARR_U, l + 1));
        MG_TRY(vcycle_rec_t<T>(l + 1, skip0));
        fold_slab = can_fold_prolong_slab<T>(l);
        fold = can_fold_prolong<T>(l) || fold_slab;
        if (!fold && !small) {
            if (prof) MG_TRY(prof_begin(l));
            MG_TRY(prolong_t<T>(l + 1, 1, MG_ARR_U, MG_ARR_U));
            if (prof) MG_TRY(prof_end(l, MG_PROF_PROLONG, 1, 1));
        }
    }
    if (small) {   // prolong-add and both post-smoothing sweeps in one launch; the result lands in TMP like a sweep's
        launch_small_prolong_post<T>(stream_, lv_[l].g, lv_[l + 1].g, coef_of<T>(lv_[l]), (T)d_.omega, ptr<T>(MG_ARR_U, l),
                                     ptr<T>(MG_ARR_U, l + 1), ptr<T>(MG_ARR_RHS, l), ptr<T>(MG_ARR_TMP, l));
        MG_HIP(hipGetLastError());
        std::swap(lv_[l].base[MG_ARR_U], lv_[l].base[MG_ARR_TMP]);
    } else if (mine) {
        // the folding pair on a slab reads the uncorrected u, whose ghost planes this cycle's residual + restriction fetched
        MG_TRY(smooth_t<T>(l, d_.smoother, d_.nu_post, MG_ARR_U, MG_ARR_RHS, false, fold ? l + 1 : -1, true, fold_slab && fuse_rr_slab));
    }
    return MG_OK;
}

// Enqueues one cycle; no host synchronisation inside (RCCL transport).
template <typename T>
int Solver::cycle_enqueue_t()
{
    pair_on_comm_level_ = -1;
    const int L = d_.levels;
    if (d_.cycle == MG_CYCLE_V) return vcycle_rec_t<T>(0);
    // --- reference sawtooth, include/multigrid.hpp:126-145 ---
    // :127  sol * RES : fine residual into `res`, sum r^2
    MG_TRY(residual_t<T>(0, MG_ARR_U, MG_ARR_RHS, MG_ARR_RES, true));
    MG_HIP(hipMemcpyAsync(d_scal_ + 2, d_scal_, sizeof(double), hipMemcpyDeviceToDevice, stream_));
    // every level's rhs is the fine residual seen through mask() (:113,121) = injection
    int src = MG_ARR_RES;
    for (int l = 0; l + 1 < L; l++) {
        if (lv_[l].dist && !lv_[l + 1].dist) {
            launch_inject<T>(stream_, lv_[l].g, stage_g_, ptr<T>(src, l), stageptr<T>(0));
            MG_HIP(hipGetLastError());
            MG_TRY(gather_S(MG_ARR_RHS));
        } else if (lv_[l].present) {
            MG_TRY(restrict_t<T>(l, MG_RESTRICT_INJECT, src, MG_ARR_RHS));
        }
        src = MG_ARR_RHS;
    }
    const int rhsL = (L == 1) ? MG_ARR_RES : MG_ARR_RHS;
    if (stage_fn_) MG_TRY(dump_stage(L - 1, false));
    // :128-131 coarse solve from err == 0
    if (lv_[L - 1].present) MG_TRY(zero_array(MG_ARR_E, L - 1));
    MG_TRY(coarse_level_t<T>(L - 1, MG_ARR_E, rhsL));
    if (stage_fn_) MG_TRY(dump_stage(L - 1, true));
    // :134-139 prolong (overwrite) + nu sweeps, coarse to fine
    for (int l = L - 2; l >= 0; l--) {
        if (lv_[l].dist && !lv_[l + 1].dist) {
            MG_TRY(scatter_S(MG_ARR_E));
            launch_prolong<T>(stream_, stage_g_, lv_[l].g, stageptr<T>(1), ptr<T>(MG_ARR_E, l), false);
            MG_HIP(hipGetLastError());
        } else if (lv_[l].present) {
            MG_TRY(prolong_t<T>(l + 1, 0, MG_ARR_E, MG_ARR_E));
        }
        if (stage_fn_) MG_TRY(dump_stage(l, true));
        if (lv_[l].present)
            MG_TRY(smooth_t<T>(l, d_.smoother, d_.nu_post, MG_ARR_E, l == 0 ? MG_ARR_RES : MG_ARR_RHS));
        if (stage_fn_) MG_TRY(dump_stage(l, true));
    }
    // :141-144
    MG_TRY(correct_t<T>(0, MG_ARR_U, MG_ARR_E));
    if (stage_fn_) MG_TRY(dump_stage(0, false));
    return MG_OK;
}

int Solver::cycle_enqueue()
{
    return d_.dtype == MG_F64 ? cycle_enqueue_t<double>() : cycle_enqueue_t<float>();
}

int Solver::cycle(mg_cycle_stats *st)
{
    MG_HIP(hipSetDevice(device_));
    MG_TRY(cycle_enqueue());
    MG_HIP(hipMemcpyAsync(h_coarse_, d_coarse_, sizeof(CoarseOut), hipMemcpyDeviceToHost, stream_));
    MG_HIP(hipMemcpyAsync(h_scal_ + 2, d_scal_ + 2, sizeof(double), hipMemcpyDeviceToHost, stream_));
    MG_HIP(hipStreamSynchronize(stream_));
    if (st) {
        st->coarse_iters = h_coarse_->iters;
        st->coarse_flag = h_coarse_->flag;
        st->coarse_relres = h_coarse_->relres;
        st->fine_sumsq_r = (d_.cycle == MG_CYCLE_SAWTOOTH) ? h_scal_[2] : 0.0;
    }
    return MG_OK;
}

int Solver::cycle_async(int count)
{
    MG_HIP(hipSetDevice(device_));
    for (int i = 0; i < count; i++) MG_TRY(cycle_enqueue());
    return MG_OK;
}

// The residual norm of main.cpp:86 can ride on the next cycle's first launch when that launch is the wide-tile Jacobi pair on
// an undistributed finest level (its first sweep holds every operand of rhs - A u): V-cycle, exactly two pre-smoothing sweeps,
// nothing between the norm and the cycle (no outer Gauss-Seidel sweeps, no stage dumps, no profiling brackets).
template <typename T>
bool Solver::pair_norm_ok() const
{
    static const bool enabled = [] { const char *e = getenv("MG_PAIR_NORM"); return !(e && e[0] == '0'); }();
    const Level &L = lv_[0];
    if (!enabled || d_.cycle != MG_CYCLE_V || d_.levels <= 1 || d_.outer_pre_gs != 0 || stage_fn_ || profiling_ || !L.present) return false;
    if (L.dist) {   // z-slabs: the Jacobi pair on the whole slab (two ghost planes), wide enough on the thinnest slab of all ranks
        static const bool slab_enabled = [] { const char *e = getenv("MG_PAIR_NORM_SLAB"); return !(e && e[0] == '0'); }();
        const Geom gs = slab_gate_geom(L);
        const bool sm = (d_.smoother == MG_SMOOTH_JACOBI && d_.nu_pre == 2) || (d_.smoother == MG_SMOOTH_RBGS && d_.nu_pre >= 1 && rb_slab_enabled());
        return slab_enabled && sm && depth2_enabled() && jacobi2_slab_ok<T>(gs) && pair_wide_ok<T>(gs);
    }
    const bool sm = (d_.smoother == MG_SMOOTH_JACOBI && d_.nu_pre == 2 && jacobi2_ok<T>(L.g)) ||
                    (d_.smoother == MG_SMOOTH_RBGS && d_.nu_pre >= 1 && rb_fused_ok<T>(L.g));   // red-black: the first sweep carries it
    return sm && nranks_ == 1 && pair_wide_ok<T>(L.g);
}

// Outer loop of src/main.cpp:72-116.
int Solver::solve(double tol, int maxit, double *hist, int hist_cap, int *n_hist,
                  mg_cycle_stats *per_cycle, const int *lock_counts, int n_lock)
{
    MG_HIP(hipSetDevice(device_));
    double nb = 0, nr = 0;
    MG_TRY(sumsq(0, MG_ARR_RHS, &nb));                       // Residual ctor, solvers.hpp:237-242
    const bool fused_norm = !lock_counts && (d_.dtype == MG_F64 ? pair_norm_ok<double>() : pair_norm_ok<float>());
    if (fused_norm) {
        // Same loop, same history: entry k is the norm after k cycles. It is computed by the first pre-smoothing pair of cycle
        // k + 1, which runs before the test; when the test says stop (or maxit is reached) that pair's output is dropped -- it
        // was written out of place, U still holds the iterate the norm belongs to.
        Level &L0 = lv_[0];
        int nh = 0;
        for (int it = 0; it <= maxit; it++) {
            void *const base_u = L0.base[MG_ARR_U], *const base_t = L0.base[MG_ARR_TMP];
            want_pair_norm_ = true; pair_norm_done_ = false;
            // the speculative launch: the Jacobi pair (both pre-smoothing sweeps), or the first red-black sweep -- ONE out-of-place
            // launch either way, so the iterate the norm belongs to is still whole when the test says stop
            const int spec = d_.smoother == MG_SMOOTH_JACOBI ? d_.nu_pre : 1;
            const int rc = d_.dtype == MG_F64 ? smooth_t<double>(0, d_.smoother, spec, MG_ARR_U, MG_ARR_RHS, false, -1, true)
                                              : smooth_t<float>(0, d_.smoother, spec, MG_ARR_U, MG_ARR_RHS, false, -1, true);
            want_pair_norm_ = false;
            MG_TRY(rc);
            if (!pair_norm_done_) { set_last_error("mg_solve: the pre-smoothing pair did not deliver the residual norm"); return MG_ERR_HIP; }
            MG_HIP(hipMemcpyAsync(h_scal_, d_scal_, sizeof(double), hipMemcpyDeviceToHost, stream_));
            MG_HIP(hipStreamSynchronize(stream_));
            nr = h_scal_[0];
            if (per_cycle && it > 0) {                       // the coarse solver's record of the cycle that has just finished
                mg_cycle_stats &st = per_cycle[it - 1];
                st.coarse_iters = h_coarse_->iters;
                st.coarse_flag = h_coarse_->flag;
                st.coarse_relres = h_coarse_->relres;
                st.fine_sumsq_r = 0.0;
            }
            const double rel = std::sqrt(nr / nb);
            if (hist && nh < hist_cap) hist[nh] = rel;
            nh++;
            if ((it > 0 && rel <= tol) || it == maxit) {     // main.cpp:88-89 / the loop bound: drop the speculative pair
                L0.base[MG_ARR_U] = base_u; L0.base[MG_ARR_TMP] = base_t;
                break;
            }
            fine_pre_done_ = spec;
            const int crc = cycle_enqueue();
            fine_pre_done_ = 0;
            MG_TRY(crc);
            if (per_cycle) MG_HIP(hipMemcpyAsync(h_coarse_, d_coarse_, sizeof(CoarseOut), hipMemcpyDeviceToHost, stream_));
        }
        MG_HIP(hipStreamSynchronize(stream_));
        if (n_hist) *n_hist = nh;
        return MG_OK;
    }
    MG_TRY(residual(0, MG_ARR_U, MG_ARR_RHS, -1, &nr));      // main.cpp:73-74
    int nh = 0;
    if (hist && nh < hist_cap) hist[nh] = std::sqrt(nr / nb);
    nh++;
    for (int it = 0; it < maxit; it++) {
        if (d_.outer_pre_gs > 0)                             // `u * GS * GS` main.cpp:85
            MG_TRY(smooth(0, MG_SMOOTH_GS_LEX, d_.outer_pre_gs, MG_ARR_U, MG_ARR_RHS));
        lock_iters_ = (lock_counts && it < n_lock) ? lock_counts[it] : -1;
        const int crc = cycle(per_cycle ? &per_cycle[it] : nullptr);  // `* MGx`
        lock_iters_ = -1;
        MG_TRY(crc);
        MG_TRY(residual(0, MG_ARR_U, MG_ARR_RHS, -1, &nr));  // main.cpp:86
        double rel = std::sqrt(nr / nb);
        if (hist && nh < hist_cap) hist[nh] = rel;
        nh++;
        if (rel <= tol) break;                               // main.cpp:88-89
    }
    if (n_hist) *n_hist = nh;
    return MG_OK;
}

int Solver::set_stage_callback(mg_stage_fn fn, void *user)
{
    if (fn && nranks_ > 1) { set_last_error("stage dumps are single-GPU only"); return MG_ERR_BAD_ARG; }
    stage_fn_ = fn; stage_user_ = user; stage_count_ = 0;
    return MG_OK;
}

// `sol` sampled on `level` (+ `err` of that level): what the reference's CREATE_GIF twin saves
// with formatVector(temp, A_level) after temp[id] += err[id] (multigrid.hpp:217-299)
int Solver::dump_stage(int level, bool add_err)
{
    const Level &L0 = lv_[0], &L = lv_[level];
    const size_t es = esize();
    const size_t cnt0 = (size_t)L0.g.nx * L0.g.ny * L0.g.nz, cnt = (size_t)L.g.nx * L.g.ny * L.g.nz;
    stage_u_.resize(cnt0 * es); stage_e_.resize(cnt * es);
    MG_TRY(get_array(MG_ARR_U, 0, stage_u_.data()));
    if (add_err) MG_TRY(get_array(MG_ARR_E, level, stage_e_.data()));
    const int sx = (L0.g.nx - 1) / (L.g.nx - 1);
    const int sz = (L.g.nz > 1) ? (L0.g.nz - 1) / (L.g.nz - 1) : 1;
    auto sample = [&](auto *u, auto *e) {
        for (int k = 0; k < L.g.nz; k++)
            for (int j = 0; j < L.g.ny; j++)
                for (int i = 0; i < L.g.nx; i++) {
                    size_t c = ((size_t)k * L.g.ny + j) * L.g.nx + i;
                    size_t f = ((size_t)(k * sz) * L0.g.ny + (size_t)j * sx) * L0.g.nx + (size_t)i * sx;
                    e[c] = add_err ? u[f] + e[c] : u[f];
                }
    };
    if (d_.dtype == MG_F64) sample(reinterpret_cast<double *>(stage_u_.data()), reinterpret_cast<double *>(stage_e_.data()));
    else sample(reinterpret_cast<float *>(stage_u_.data()), reinterpret_cast<float *>(stage_e_.data()));
    stage_fn_(stage_user_, stage_count_++, level, L.g.nx, L.g.nz, stage_e_.data());
    return MG_OK;
}

int Solver::sync()
{
    MG_HIP(hipSetDevice(device_));
    MG_HIP(hipStreamSynchronize(stream_));
    return MG_OK;
}

int Solver::prof_begin(int)
{
    if (prof_used_ + 2 > prof_ev_.size()) {
        size_t old = prof_ev_.size();
        prof_ev_.resize(old + 256);
        for (size_t i = old; i < prof_ev_.size(); i++) MG_HIP(hipEventCreate(&prof_ev_[i]));
    }
    MG_HIP(hipEventRecord(prof_ev_[prof_used_], stream_));
    return MG_OK;
}

int Solver::prof_end(int, int kind, int units, int launches)
{
    MG_HIP(hipEventRecord(prof_ev_[prof_used_ + 1], stream_));
    prof_used_ += 2;
    prof_kind_.push_back(kind);
    prof_units_.push_back(units);
    prof_launches_.push_back(launches);
    return MG_OK;
}

int Solver::profile_begin()
{
    profiling_ = true;
    prof_used_ = 0;
    prof_sweeps_ = 0;
    prof_kind_.clear(); prof_units_.clear(); prof_launches_.clear();
    return MG_OK;
}

int Solver::profile_end(double *ms, int *sweeps)
{
    MG_HIP(hipSetDevice(device_));
    MG_HIP(hipStreamSynchronize(stream_));
    profiling_ = false;
    double tot = 0;
    prof_sweeps_ = 0; prof_fused_ms_ = 0; prof_fused_sweeps_ = 0;
    for (int k = 0; k < MG_PROF_KINDS; k++) { prof_ms_[k] = 0; prof_n_[k] = 0; }
    for (size_t i = 0; i + 1 < prof_used_; i += 2) {
        float f = 0;
        MG_HIP(hipEventElapsedTime(&f, prof_ev_[i], prof_ev_[i + 1]));
        const int k = prof_kind_[i / 2];
        prof_ms_[k] += f; prof_n_[k] += prof_launches_[i / 2];
        if (k == MG_PROF_SMOOTH_PROLONG) { prof_fused_ms_ += f; prof_fused_sweeps_ += prof_units_[i / 2]; }
        else if (k == MG_PROF_SMOOTH) { tot += f; prof_sweeps_ += prof_units_[i / 2]; }
    }
    if (ms) *ms = tot;
    if (sweeps) *sweeps = prof_sweeps_;
    return MG_OK;
}

int Solver::profile_fused(double *ms, int *sweeps) const
{
    if (ms) *ms = prof_fused_ms_;
    if (sweeps) *sweeps = prof_fused_sweeps_;
    return MG_OK;
}

int Solver::profile_get(int kind, double *ms, int *launches) const
{
    if (ms) *ms = prof_ms_[kind];
    if (launches) *launches = prof_n_[kind];
    return MG_OK;
}

int Solver::comm_info(int *rank, int *nranks, int *transport_ranks, const char **transport) const
{
    if (rank) *rank = rank_;
    if (nranks) *nranks = nranks_;
    if (transport_ranks) *transport_ranks = comm_ ? comm_->transport_ranks() : 1;
    if (transport) *transport = comm_ ? comm_->name() : "none";
    return MG_OK;
}

int Solver::timer_start()
{
    MG_HIP(hipSetDevice(device_));
    MG_HIP(hipEventRecord(ev0_, stream_));
    return MG_OK;
}

int Solver::timer_stop(double *ms)
{
    MG_HIP(hipSetDevice(device_));
    MG_HIP(hipEventRecord(ev1_, stream_));
    MG_HIP(hipEventSynchronize(ev1_));
    float f = 0;
    MG_HIP(hipEventElapsedTime(&f, ev0_, ev1_));
    if (ms) *ms = (double)f;
    return MG_OK;
}

}  // namespace mg
