// mg_solver.h -- host-side owner of the HBM-resident grid hierarchy and the
// stream-ordered cycle drivers behind the C-ABI of include/mg_hip.h.
#ifndef MG_SOLVER_H
#define MG_SOLVER_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mg_hip.h"
#include "mg_comm.h"
#include "mg_geom.h"

namespace mg {

void set_last_error(const std::string &msg);
const std::string &last_error();
int validate_desc(const mg_desc *d, std::string *why);
int level_n(const mg_desc &d, int level);
int level_nz(const mg_desc &d, int level);  // planes of a level (== n unless semi-coarsening)
void level_coefficients(const mg_desc &d, int level, double out[4]);

struct SlabPlan {
    int z0 = 0, nz = 0;           // owned planes of this rank on the level
    int first_gathered_level = 0;  // levels >= this live on rank 0 only
};
// host-only partition arithmetic (no HIP call), shared by the library and the CPU tests
int plan_slab(const mg_desc &d, int nranks, int rank, int level, SlabPlan *out, std::string *why);
// the planes of the FIRST GATHERED level (level == first_gathered_level) that coincide with this
// rank's slab of the last distributed level: what the rank restricts into / prolongs from before
// the level is gathered on / after it is scattered from rank 0
int plan_stage(const mg_desc &d, int nranks, int rank, SlabPlan *out, std::string *why);

constexpr int NUM_ARR = 5;

struct Level {
    Geom g{};
    size_t alloc_elems = 0;      // (nz + 2) * plane
    void *base[NUM_ARR] = {};    // allocations (ghost plane first)
    double coef[4] = {};
    bool present = true;         // false: level not held by this rank (gathered on rank 0)
    bool dist = false;           // true: z-slab of a level distributed over all ranks
    int nz_min = 0;              // thinnest slab of the level over all ranks (== g.nz when the level is not distributed)
    int gh = 1;                  // ghost planes either side of the owned ones: 2 on distributed levels (one exchange then
                                 // feeds a fused sweep pair / residual + restriction), 1 elsewhere (unused, zero)
    bool overlap = false;        // distributed level whose slabs are big enough for the interior launch to hide the halo exchange
                                 // (MG_OVERLAP_MIN_MB, decided on the thinnest slab: the same answer on every rank)
    bool rhs_halo_ok = false;    // distributed level: the RHS array's first ghost planes hold the neighbours' planes
    void *zebra = nullptr;       // MG_SMOOTH_ZEBRA_Y / _X: cp(j), den(j) of the line solve (2 * ny or 2 * nx values, device)
};

class Solver {
public:
    // takes ownership of `comm` (nullptr: single GPU)
    Solver(const mg_desc &d, int device, Comm *comm = nullptr);
    ~Solver();
    int init();  // allocates; returns mg_status

    int set_array(int which, int level, const void *host);
    int stage_rows(int which, int level, void *host, bool to_device);
    int get_array(int which, int level, void *host);
    int zero_array(int which, int level);

    int smooth(int level, int smoother, int sweeps, int arr_x, int arr_rhs);
    int residual(int level, int arr_x, int arr_rhs, int arr_r, double *sumsq);
    int sumsq(int level, int arr, double *out);
    int restrict_to(int fine_level, int kind, int arr_src, int arr_dst);
    int prolong(int coarse_level, int add, int arr_src, int arr_dst);
    int correct(int arr_u, int arr_e);
    int coarse_solve(int level, int arr_x, int arr_rhs, mg_cycle_stats *st);
    int coarse_solve_ex(int level, int arr_x, int arr_rhs, int smoother, int maxit, double tol, int fixed,
                        mg_cycle_stats *st);
    int cycle(mg_cycle_stats *st);
    int cycle_async(int count);
    // lock_counts != nullptr: outer iteration i < n_lock runs its coarse solve for exactly
    // lock_counts[i] sweeps (mg_solve_lockstep)
    int solve(double tol, int maxit, double *hist, int hist_cap, int *n_hist,
              mg_cycle_stats *per_cycle, const int *lock_counts = nullptr, int n_lock = 0);
    int set_stage_callback(mg_stage_fn fn, void *user);
    int sync();
    int timer_start();
    int timer_stop(double *ms);
    int profile_begin();
    int profile_end(double *ms, int *sweeps);
    int profile_fused(double *ms, int *sweeps) const;
    int profile_get(int kind, double *ms, int *launches) const;
    int comm_info(int *rank, int *nranks, int *transport_ranks, const char **transport) const;
    size_t device_bytes() const { return bytes_; }
    long long comm_groups() const { return comm_groups_; }
    long long comm_bytes_sent() const { return comm_bytes_; }

    const mg_desc &desc() const { return d_; }
    int nlevels() const { return d_.levels; }
    const Level &level(int l) const { return lv_[l]; }

private:
    template <typename T> T *ptr(int which, int level) const;  // local plane 0
    // x_zero: the caller knows x == 0 (fresh coarse-level guess): the first Jacobi sweep of a
    // fast-path level then skips reading x (and the caller skips the memset)
    template <typename T> int smooth_t(int level, int smoother, int sweeps, int ax, int ar, bool x_zero = false,
                                       int corr_level = -1, bool e_scratch = false, bool u_halo_ok = false);
    template <typename T> bool can_fold_prolong(int level) const;
    template <typename T> bool can_fold_prolong_slab(int level) const;   // both levels distributed: the slab pair folds P e in
    template <typename T> bool can_fold_prolong_replicated(int level) const;   // slab level over a level every rank holds whole
    template <typename T> int pair_on_slab_t(int level, bool rb);
    template <typename T> int pair_on_slab2_t(int level, bool rb, int corr_level = -1, bool u_halo_ok = false, double *norm_partials = nullptr, int *norm_np = nullptr);   // depth-2 ghosts: one exchange, the fused kernel on the whole slab
    template <typename T> int resid_restrict_on_slab_t(int level, const Geom &gc, T *coarse_rhs);
    int refresh_rhs_halo(int level);
    template <typename T> bool can_skip_zeroing(int level) const;
    template <typename T> int residual_t(int level, int ax, int ar, int arr_r, bool want_norm);
    template <typename T> int sumsq_t(int level, int arr);
    template <typename T> int restrict_t(int fl, int kind, int as, int ad);
    template <typename T> int prolong_t(int cl, int add, int as, int ad);
    template <typename T> int correct_t(int level, int au, int ae);
    template <typename T> int coarse_full_t();
    // coarse "solve" of level l: persistent one-workgroup kernel, or -- when the level is too big
    // for one workgroup and the mode is MG_COARSE_FIXED -- coarse_maxit regular sweeps
    template <typename T> int coarse_level_t(int l, int ax, int ar, bool x_zero = false);  // coarse solve of a still-distributed coarsest level, gathered
    int exchange(int which, int level, int depth = 1);        // `depth` ghost planes <-> z-neighbours (on the main stream)
    // the same on the comm stream, after the main stream's work so far; record = false: the caller puts more work that needs the halo
    // on the comm stream (the boundary pieces of a slab operation) and records the event itself with halo_work_done()
    int exchange_begin(int which, int level, int depth = 1, bool record = true);
    int halo_work_done();
    // level whose U boundary planes were last written on the COMMUNICATION stream (by the boundary piece of a slab pair) with
    // nothing touching U since: the next exchange of those planes needs no wait for the main stream. -1: none.
    int pair_on_comm_level_ = -1;
    int halo_ops(int which, int level, int depth, P2POp *ops);
    int exchange_end();                          // main stream waits for the halo
    // Runs a stencil launch over a distributed level with the halo exchange of `arr_x` hidden
    // behind the interior planes: launch(sub-slab geometry, element offset of its first plane)
    template <typename F> int overlapped(int level, int arr_x, F &&launch);
    int gather_S(int arr);                       // staging rhs slabs -> lv_[T_+1].base[arr] on rank 0
    int scatter_S(int arr);                      // lv_[T_+1].base[arr] on rank 0 -> staging u slabs (+ upper ghost)
    template <typename T> T *stageptr(int k) const { return reinterpret_cast<T *>(stage_base_[k]) + stage_g_.plane; }
    int gather_T(int which, int fullk);          // slabs of level T_ -> full_[fullk] on rank 0
    int scatter_T(int fullk, int which);         // full_[fullk] on rank 0 -> slabs of level T_
    int allreduce(double *dptr, int n);
    template <typename T> T *fullptr(int k) const { return reinterpret_cast<T *>(full_[k]) + gfull_.plane; }
    template <typename T> int coarse_t(int level, int ax, int ar, bool x_zero = false);
    template <typename T> int coarse_ex_t(int level, int ax, int ar, int smoother, int maxit, double tol, int fixed, bool x_zero = false);
    template <typename T> int cycle_enqueue_t();
    template <typename T> int vcycle_rec_t(int l, bool u_zero = false);
    int cycle_enqueue();
    bool check_arr(int which, int level, const char *fn) const;
    size_t esize() const { return d_.dtype == MG_F64 ? 8 : 4; }

    mg_desc d_;
    int device_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    hipEvent_t ev_stage_[2] = {nullptr, nullptr};   // one per half of the host staging buffer (stage_rows)
    std::vector<Level> lv_;
    double *d_partials_ = nullptr;  // per-block partial sums
    double *d_scal_ = nullptr;      // [0] sum r^2, [1] sum b^2, [2] cycle's fine sum r^2
    CoarseOut *d_coarse_ = nullptr;
    double *h_scal_ = nullptr;      // pinned mirrors
    CoarseOut *h_coarse_ = nullptr;
    CoarseOut *h_fixed_ = nullptr;  // pinned: stats reported when the coarse level is swept, not solved
    size_t bytes_ = 0;
    // slab decomposition (mg_create_distributed*): levels 0..T_ are distributed, deeper
    // levels live on rank 0; full_[] are rank 0's gathered copies of level T_
    Comm *comm_ = nullptr;
    int rank_ = 0, nranks_ = 1, T_ = -1;
    hipStream_t comm_stream_ = nullptr;
    hipEvent_t ev_ready_ = nullptr, ev_halo_ = nullptr;
    void *h_stage_ = nullptr;      // pinned staging buffer of set_array / get_array
    size_t h_stage_bytes_ = 0;
    bool overlap_ = true;  // MG_OVERLAP=0 disables (debugging)
    bool replicate_ = false;  // gathered levels are held and run by every rank (all-gather in, no scatter out)
    long long comm_groups_ = 0, comm_bytes_ = 0;   // message groups posted / bytes sent by this rank (mg_comm_stats)
    int post(const P2POp *ops, int n, hipStream_t s);  // comm_->batch + the counters
    int lock_iters_ = -1;  // >= 0: the next coarse solve runs exactly this many sweeps (lock-step parity mode)
    // Outer loop with the residual norm taken inside the next cycle's first pre-smoothing pair (Solver::solve): want_pair_norm_
    // asks smooth_t for it, pair_norm_done_ says the launch delivered it into d_scal_[0], fine_pre_done_ tells vcycle_rec_t how
    // many of level 0's pre-smoothing sweeps have already run (Jacobi: the pair = 2; red-black: the first sweep = 1).
    bool want_pair_norm_ = false, pair_norm_done_ = false;
    int fine_pre_done_ = 0;
    template <typename T> bool pair_norm_ok() const;
    Geom gfull_{};
    void *full_[3] = {nullptr, nullptr, nullptr};
    std::vector<SlabPlan> planT_;
    // staging slab of the first gathered level (every rank): [0] restricted rhs, [1] correction
    Geom stage_g_{};
    void *stage_base_[2] = {nullptr, nullptr};
    std::vector<SlabPlan> planS_;
    // CREATE_GIF-style stage dumps (mg_set_stage_callback)
    int dump_stage(int level, bool add_err);
    mg_stage_fn stage_fn_ = nullptr;
    void *stage_user_ = nullptr;
    int stage_count_ = 0;
    std::vector<char> stage_u_, stage_e_;
    // in-region timing of the finest-grid smoother (mg_profile_begin/end)
    bool profiling_ = false;
    std::vector<hipEvent_t> prof_ev_;
    size_t prof_used_ = 0;
    int prof_sweeps_ = 0;
    std::vector<int> prof_kind_;   // per event pair: mg_prof_kind
    std::vector<int> prof_units_;  // per event pair: sweeps (smoother kinds) or 1
    std::vector<int> prof_launches_;
    double prof_fused_ms_ = 0;
    int prof_fused_sweeps_ = 0;
    double prof_ms_[MG_PROF_KINDS] = {};
    int prof_n_[MG_PROF_KINDS] = {};
    // brackets [begin, end) with HIP events when profiling the finest level
    int prof_begin(int level);
    int prof_end(int level, int kind, int units, int launches);
};

}  // namespace mg
#endif
