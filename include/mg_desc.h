/*
 * mg_desc.h -- problem / cycle descriptor shared by the C-ABI (mg_hip.h) and by
 * the CPU oracle (oracle/gmg_oracle.h).  Plain C, no dependencies.
 *
 * The fields restate the knobs of the reference GeometricMultigrid program
 * (citations relative to /root/reference/GeometricMultigrid/):
 *   n, length, alpha, levels   <- CLI -n -w -a -ml         src/utilities.cpp:3-132
 *   smoother                   <- CLI -smt, enum SMOOTHERS include/utilities.hpp:9-14
 *   nu_post = 5                <- SawtoothMGIteration::nu  include/multigrid.hpp:105
 *   coarse_maxit/tol           <- Solver(…,2000,1e-1,1)    include/multigrid.hpp:123
 *   outer_pre_gs = 2           <- `u * GS * GS * MGx`      src/main.cpp:85,95,106
 * Everything the reference does not have (3-D, fp32, omega, red-black, V(nu1,nu2),
 * full weighting, fixed coarse sweeps) is an extension whose oracle is our own
 * CPU restatement (oracle/), itself pinned to the reference on the 2-D cases.
 */
#ifndef MG_DESC_H
#define MG_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum mg_dtype       { MG_F64 = 0, MG_F32 = 1 };
/* numbering of the first two mirrors -smt 0 / 1 (include/utilities.hpp:9-14) */
/* MG_SMOOTH_ZEBRA_Y (EXTENSION): zebra line Gauss-Seidel, lines along y solved exactly
 * (Thomas), coloured by the parity of x (+ z): for operators whose y-coupling dominates. The coarsest-grid
 * solver of such a hierarchy smooths with red-black Gauss-Seidel.
 * MG_SMOOTH_ZEBRA_X: the same with lines along x (the fast axis), coloured by the parity of y (+ z): for a dominant
 * x-coupling (aniso[0] >> 1). */
enum mg_smoother    { MG_SMOOTH_GS_LEX = 0, MG_SMOOTH_JACOBI = 1, MG_SMOOTH_RBGS = 2, MG_SMOOTH_ZEBRA_Y = 3,
                      MG_SMOOTH_ZEBRA_X = 4 };
enum mg_cycle_kind  { MG_CYCLE_SAWTOOTH = 0,   /* reference cycle, multigrid.hpp:126-145 */
                      MG_CYCLE_V        = 1 }; /* standard V(nu_pre,nu_post), extension  */
enum mg_restriction { MG_RESTRICT_INJECT = 0,  /* reference: aliasing via mask()         */
                      MG_RESTRICT_FULLW  = 1 };/* 9/27-point full weighting, extension   */
enum mg_coarse_mode { MG_COARSE_TOL   = 0,     /* reference Solver::Solve, solvers.hpp:324-342 */
                      MG_COARSE_FIXED = 1 };   /* exactly coarse_maxit sweeps, extension */

typedef struct mg_desc {
    int32_t dim;          /* 2 (reference) or 3                                         */
    int32_t n;            /* nodes per side on the finest grid, boundary included;
                             must satisfy (n-1) % 2^(levels-1) == 0 and coarsest n >= 3  */
    int32_t levels;       /* number of grids (reference -ml)                             */
    int32_t dtype;        /* enum mg_dtype                                               */
    double  length;       /* side of the square / cube (reference -w)                    */
    double  alpha;        /* diffusion constant (reference -a)                           */
    int32_t cycle;        /* enum mg_cycle_kind                                          */
    int32_t smoother;     /* enum mg_smoother, used on every level of the cycle          */
    double  omega;        /* Jacobi damping, 1.0 == reference (undamped)                 */
    int32_t nu_pre;       /* V-cycle pre-smoothing sweeps (ignored by the sawtooth)      */
    int32_t nu_post;      /* post-smoothing sweeps per level (reference: 5)              */
    int32_t restriction;  /* enum mg_restriction (V-cycle only; sawtooth always injects) */
    int32_t coarse_mode;  /* enum mg_coarse_mode                                         */
    int32_t coarse_maxit; /* reference 2000 (or the fixed sweep count)                   */
    int32_t outer_pre_gs; /* lexicographic GS sweeps on the finest grid before each
                             cycle inside mg_solve (reference: 2)                        */
    double  coarse_tol;   /* reference 1e-1                                              */
    double  aniso[3];     /* per-axis multipliers of alpha, order {x(fast), y, z(slow)};
                             {1,1,1} == isotropic reference operator                     */
    int32_t dist_min_n;   /* multi-GPU only: levels with fewer nodes per side than this are
                             gathered on rank 0 instead of being slab-decomposed.
                             0 = default (257: below that a halo exchange costs more than the
                             sweep it feeds)                                */
    int32_t semi_xy;      /* 3-D only: number k of leading SEMI-coarsenings. The first k level
                             transitions coarsen x and y only (levels 0..k keep the finest grid's z
                             resolution), the remaining ones coarsen all three axes. For
                             aniso[2] = eps << 1 (strong coupling inside the x-y planes, BASELINE
                             config 5) choose k ~ log4(1/eps): after k semi-coarsenings the
                             operator is roughly isotropic again. 0 = standard coarsening.          */
} mg_desc;

/* Fills *d with the reference defaults for a 2-D run (`Multigrid -n n -ml levels …`). */
static inline void mg_desc_reference_defaults(mg_desc *d, int n, int levels,
                                              double length, double alpha, int smoother)
{
    d->dim = 2; d->n = n; d->levels = levels; d->dtype = MG_F64;
    d->length = length; d->alpha = alpha;
    d->cycle = MG_CYCLE_SAWTOOTH; d->smoother = smoother; d->omega = 1.0;
    d->nu_pre = 0; d->nu_post = 5; d->restriction = MG_RESTRICT_INJECT;
    d->coarse_mode = MG_COARSE_TOL; d->coarse_maxit = 2000; d->outer_pre_gs = 2;
    d->coarse_tol = 1e-1;
    d->aniso[0] = d->aniso[1] = d->aniso[2] = 1.0;
    d->dist_min_n = 0; d->semi_xy = 0;
}

/* per-cycle statistics returned by mg_cycle / orc_mg_cycle */
typedef struct mg_cycle_stats {
    int32_t coarse_iters;    /* smoother sweeps spent by the coarse solve               */
    int32_t coarse_flag;     /* Solver::Status(): 1 == hit maxit, 0 == converged        */
    double  coarse_relres;   /* "Achieved residual on coarse grid" multigrid.hpp:131    */
    double  fine_sumsq_r;    /* sum r^2 of the fine residual the cycle started from
                                (sawtooth only; 0 for the V-cycle)                       */
} mg_cycle_stats;

#ifdef __cplusplus
}
#endif
#endif /* MG_DESC_H */
