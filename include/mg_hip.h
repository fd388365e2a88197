/*
 * mg_hip.h -- C-ABI of the MI355X-native geometric-multigrid hot path (libmg_hip.so).
 *
 * The reference (Stefo01/multigrid_prj, GeometricMultigrid/) has no FFI: its
 * boundary is a set of C++ operator classes applied to std::vector<double> with
 * `x * Op`.  This header is the plain-C boundary a maintainer binds
 * instead; each entry point names the reference interface it replaces (paths
 * relative to /root/reference/GeometricMultigrid/).  The C++ mirror of the
 * reference classes that sits on top of it is include/multigrid_hip.hpp, the
 * ctypes binding is multigrid_prj_amd/capi.py.
 *
 * Conventions: every function returns 0 on success or a negative mg_status and
 * records a message retrievable with mg_last_error(); no exception crosses the
 * ABI.  The library owns all device memory; the caller owns all host buffers.
 * One handle <-> one host thread. Work is enqueued on a HIP stream owned by the
 * handle; functions that return scalars or copy to host synchronise that stream.
 * There is NO CPU fallback: without a HIP device mg_create fails with
 * MG_ERR_NO_DEVICE.
 *
 * Host arrays are dense and contiguous: 2-D a[j*n+i] (j = reference row, i.e.
 * y = length - j*h; i = column), 3-D a[(k*n+j)*n+i]; element type = desc.dtype.
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stddef.h>
#include "mg_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_solver *mg_handle;

enum mg_status {
    MG_OK = 0,
    MG_ERR_INVALID_DESC = -1,  /* see mg_last_error(); includes n/levels mismatch the
                                  reference silently mis-handles            */
    MG_ERR_NO_DEVICE = -2,
    MG_ERR_HIP = -3,
    MG_ERR_BAD_ARG = -4,
    MG_ERR_COMM = -5
};

/* device arrays of one level, addressable by tests and by the C++ mirror */
enum mg_array {
    MG_ARR_U = 0,   /* level 0: solution `u` (main.cpp:49); V-cycle: u_l on every level  */
    MG_ARR_E = 1,   /* sawtooth error `err` (multigrid.hpp:96), one dense array per level */
    MG_ARR_RHS = 2, /* level 0: `fvec` (main.cpp:45); l>0: restricted residual           */
    MG_ARR_TMP = 3, /* Jacobi `temp` (solvers.hpp:58) / residual scratch                 */
    MG_ARR_RES = 4  /* level 0 only: `res` (multigrid.hpp:95)                            */
};

const char *mg_last_error(void);
int mg_device_count(int *count);

/* SquareDomain + PoissonMatrix hierarchy (main.cpp:32-41) and the operator objects of
 * SawtoothMGIteration's constructor (multigrid.hpp:108-124), resident in HBM.
 * device < 0 selects the current HIP device. */
int mg_create(const mg_desc *desc, int device, mg_handle *out);
int mg_destroy(mg_handle h);

/* SquareDomain::getWidth() of level l (domain.hpp:82, domain.cpp:9-12) */
int mg_level_n(mg_handle h, int level, int *n);
/* number of z-planes this rank holds of level l (1 in 2-D; the local slab when distributed;
 * with semi-coarsening every level keeps the finest grid's z resolution) */
int mg_level_nz(mg_handle h, int level, int *nz);
/* PoissonMatrix coefficients of level l: out = {cx, cy, cz, cd}
 * (linear_system.hpp:17,27-28,37-38) */
int mg_level_coefficients(mg_handle h, int level, double out[4]);

/* DataVector (linear_system.hpp:85-92) is assembled by the caller on the host;
 * these move dense host arrays to/from the padded device layout. */
int mg_set_rhs(mg_handle h, const void *host_b);            /* == set_array(RHS, 0) */
int mg_set_solution(mg_handle h, const void *host_u);       /* == set_array(U, 0)   */
int mg_get_solution(mg_handle h, void *host_u);
int mg_set_array(mg_handle h, int which, int level, const void *host);
int mg_get_array(mg_handle h, int which, int level, void *host);
int mg_zero_array(mg_handle h, int which, int level);

/* `x * smoother` `sweeps` times on level l:  A_l x = rhs
 *   MG_SMOOTH_JACOBI -> Jacobi_iteration::apply_iteration_to_vec  solvers.hpp:64-83
 *   MG_SMOOTH_GS_LEX -> Gauss_Seidel_iteration::…                 solvers.hpp:33-48
 *   MG_SMOOTH_RBGS   -> red-black GS (extension)
 *   MG_SMOOTH_ZEBRA_Y / MG_SMOOTH_ZEBRA_X -> zebra line GS along y / along x (extension; only on a handle created
 *                        with that smoother, which tabulates the line factors per level)
 * arr_x / arr_rhs name which device arrays play x and rhs. */
int mg_smooth(mg_handle h, int level, int smoother, int sweeps, int arr_x, int arr_rhs);

/* `x * RES`: Residual::apply_iteration_to_vec solvers.hpp:257-295. arr_r < 0 is the
 * non-saving branch (:277-294). *sumsq_r = sum r^2 (the member `norm`). */
int mg_residual(mg_handle h, int level, int arr_x, int arr_rhs, int arr_r, double *sumsq_r);
/* Residual::refresh_normalization_constant solvers.hpp:244-254 */
int mg_sumsq(mg_handle h, int level, int arr, double *sumsq);

/* level l -> l+1. kind MG_RESTRICT_INJECT is what the reference does implicitly by
 * building every level on `res` through mask() (multigrid.hpp:113,121; domain.hpp:78-80) */
int mg_restrict(mg_handle h, int fine_level, int kind, int arr_src, int arr_dst);
/* level l -> l-1: InterpolationClass::interpolate src/multigrid.cpp:3-27 (add == 0,
 * overwrite); add != 0 is the V-cycle's fine += P coarse (extension) */
int mg_prolong(mg_handle h, int coarse_level, int add, int arr_src, int arr_dst);
/* sol += err; err = 0 on the finest grid, multigrid.hpp:141-144 */
int mg_correct(mg_handle h, int arr_u, int arr_e);
/* Solver::Solve on `level` (solvers.hpp:324-342 as instantiated at multigrid.hpp:123),
 * one persistent workgroup; honours desc.coarse_{mode,maxit,tol}, desc.smoother */
int mg_coarse_solve(mg_handle h, int level, int arr_x, int arr_rhs, mg_cycle_stats *st);
/* the same with the Solver constructor's arguments (smoother, maxit, tol) given per call
 * instead of taken from the descriptor; fixed != 0 runs exactly maxit sweeps */
int mg_coarse_solve_ex(mg_handle h, int level, int arr_x, int arr_rhs, int smoother, int maxit,
                       double tol, int fixed, mg_cycle_stats *st);

/* SawtoothMGIteration::apply_iteration_to_vec multigrid.hpp:126-145 (or the V-cycle
 * extension, per desc.cycle) applied to the solution array U of level 0 */
int mg_cycle(mg_handle h, mg_cycle_stats *st);
/* enqueue `count` cycles without any host synchronisation (benchmark path) */
int mg_cycle_async(mg_handle h, int count);
/* outer loop of main.cpp:72-116; hist[0] = initial relative residual; *n_hist = entries
 * produced (also counted when hist_cap is too small); per_cycle may be NULL */
int mg_solve(mg_handle h, double tol, int maxit, double *hist, int hist_cap, int *n_hist,
             mg_cycle_stats *per_cycle);

/* Lock-step parity mode: mg_solve with the coarse Solver's stopping test taken out of the
 * comparison. Solver::Solve (solvers.hpp:324-342) stops on `Norm() > 0.1`, so a last-bit difference in a
 * sum of squares can move the stop by one sweep and everything after it in the 3rd-4th digit. Here the
 * coarse solve of outer iteration i spends exactly coarse_counts[i] sweeps -- the counts a reference run
 * spent (tests/golden/ref_solve.json) -- so that every kernel of the solve can be held to the
 * reference's numbers tightly. Iterations beyond n_counts run free, like mg_solve. */
int mg_solve_lockstep(mg_handle h, double tol, int maxit, const int *coarse_counts, int n_counts,
                      double *hist, int hist_cap, int *n_hist, mg_cycle_stats *per_cycle);

/* Debug stage dumps of the sawtooth cycle -- the reference's CREATE_GIF twin
 * (multigrid.hpp:160-316) writes `sol + err` sampled on the level being worked on after every
 * stage: before and after the coarse solve, after each interpolation, after each level's
 * sweeps, and the corrected solution. With a callback installed mg_cycle does the same
 * (synchronising after every stage): `values` is a dense host array of n*n*nz elements of the
 * descriptor's dtype, `stage` counts the calls since the callback was installed (the
 * reference's frame counter). fn == NULL removes it. Single-GPU handles only. */
typedef void (*mg_stage_fn)(void *user, int stage, int level, int n, int nz, const void *values);
int mg_set_stage_callback(mg_handle h, mg_stage_fn fn, void *user);

int mg_sync(mg_handle h);
/* HIP-event timing on the handle's own stream (torch.cuda.Event would not see it) */
int mg_timer_start(mg_handle h);
int mg_timer_stop(mg_handle h, double *milliseconds);
/* In-region kernel timing for bench.py: between begin and end every finest-grid
 * smoother call made by mg_cycle/mg_cycle_async/mg_smooth is bracketed by HIP events on
 * the handle's stream. end synchronises and returns the summed time and the number of
 * sweeps (kernel launches; a red-black sweep counts once, both colours included). */
int mg_profile_begin(mg_handle h);
int mg_profile_end(mg_handle h, double *smoother_ms, int *smoother_sweeps);
/* A V-cycle folds the prolongation into the first post-smoothing pair when it can (one
 * launch computing J(J(u + P e))); those segments are not smoother-only work, so they are
 * left out of mg_profile_end's totals and reported here (valid after mg_profile_end). */
int mg_profile_fused(mg_handle h, double *fused_ms, int *fused_sweeps);
/* The same events, by kind of finest-level launch (valid after mg_profile_end): summed milliseconds and
 * number of launches of  MG_PROF_SMOOTH          smoother launches (a fused pair is ONE launch of two sweeps)
 *                        MG_PROF_SMOOTH_PROLONG  the post-smoothing launch that also applies P e
 *                        MG_PROF_RESID_RESTRICT  residual + restriction of level 0 (fused or as two kernels)
 *                        MG_PROF_PROLONG         separate prolongation into level 0 */
enum mg_prof_kind { MG_PROF_SMOOTH = 0, MG_PROF_SMOOTH_PROLONG = 1, MG_PROF_RESID_RESTRICT = 2, MG_PROF_PROLONG = 3, MG_PROF_KINDS = 4 };
int mg_profile_get(mg_handle h, int kind, double *ms, int *launches);
/* bytes of HBM held by the handle */
int mg_device_bytes(mg_handle h, size_t *bytes);

/* ---- multi-GPU (z-slab domain decomposition, RCCL halo exchange) ---- */
#define MG_COMM_ID_BYTES 128
/* rank 0 creates the RCCL unique id; the caller ships it to the other ranks by any
 * means (bench.py: torch.distributed broadcast) */
int mg_comm_unique_id(void *id128);
/* single-process smoke test of the RCCL transport on the current device: communicator of
 * one rank, grouped send/recv to self of `bytes` bytes, all-reduce of one double */
int mg_comm_selftest(size_t bytes);
/* rank / size of the handle's decomposition and what the transport itself reports (RCCL: ncclCommCount;
 * 1 for a single-GPU handle); transport = "none" | "rccl" | "host-callbacks" (static string) */
int mg_comm_info(mg_handle h, int *rank, int *nranks, int *transport_ranks, const char **transport);
/* cumulative communication of this rank since creation: message groups posted (halo exchanges, gathers, scatters,
 * all-reduces: one ncclGroup / one host batch each) and bytes sent in them */
int mg_comm_stats(mg_handle h, long long *groups, long long *bytes_sent);
/* like mg_create, for rank `rank` of `nranks` (one process per GPU) */
int mg_create_distributed(const mg_desc *desc, int device, int rank, int nranks,
                          const void *id128, mg_handle *out);
/* The same distributed solver with the exchanges delegated to the HOST (test transport:
 * the library stages halo planes through pinned host buffers and calls back; tests wire the
 * callbacks to torch.distributed/gloo so two processes sharing one GPU can exercise the
 * whole multi-rank path). `batch` must post every op of the list concurrently and return
 * when all have completed; `allreduce_sum` sums n doubles over all ranks in place. Both
 * return 0 on success. */
typedef struct mg_p2p_op {
    int32_t peer;     /* rank to send to / receive from */
    int32_t is_send;  /* 1 = send, 0 = receive          */
    void   *buf;      /* host buffer                    */
    size_t  bytes;
} mg_p2p_op;
typedef struct mg_host_comm {
    void *ctx;
    int (*batch)(void *ctx, const mg_p2p_op *ops, int nops);
    int (*allreduce_sum)(void *ctx, double *vals, int n);
} mg_host_comm;
int mg_create_distributed_hostcomm(const mg_desc *desc, int device, int rank, int nranks,
                                   const mg_host_comm *comm, mg_handle *out);
/* MEASUREMENT ONLY: rank `rank` of `nranks` with no peers -- every exchange and all-reduce is a no-op, so the
 * results are meaningless; the handle runs one rank's launch schedule (slab kernels, boundary launches, replicated
 * coarse levels) so that its compute time can be measured on a single GPU (bench.py --transport dry). */
int mg_create_distributed_dryrun(const mg_desc *desc, int device, int rank, int nranks, mg_handle *out);
/* host-only partition plan (no GPU needed): z-planes [z0, z0+nz) of level l owned by
 * rank r, and the first level that is agglomerated on rank 0 */
int mg_plan_slab(const mg_desc *desc, int nranks, int rank, int level, int *z0, int *nz,
                 int *first_gathered_level);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */
