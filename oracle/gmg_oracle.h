/*
 * gmg_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement (plain C) of the
 * reference GeometricMultigrid hot path (Stefo01/multigrid_prj,
 * /root/reference/GeometricMultigrid/{include/solvers.hpp, include/multigrid.hpp,
 * src/multigrid.cpp, include/linear_system.hpp, include/domain.hpp, src/domain.cpp,
 * src/main.cpp}).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (multigrid_prj_amd/, libmg_hip.so)
 * never links or calls it.
 *
 * PARITY PIN: the 2-D fp64 sawtooth path of this oracle is checked against
 *   (a) the reference's own fixtures  GeometricMultigrid/test/MGGS4.txt,
 *       WebInterface/MGGS4.txt (+ x.mtx), and
 *   (b) outputs of the reference itself compiled here (oracle/_ref, recipe in
 *       oracle/Makefile), committed as tests/golden/ref_*.json,
 * by tests/test_oracle_vs_reference.py.  The 3-D / fp32 / omega / red-black /
 * V-cycle / full-weighting extensions have NO reference counterpart ("parity
 * unpinned" w.r.t. the reference): for them this file IS the definition, tied
 * to the reference only through the shared 2-D operator bodies.
 */
#ifndef GMG_ORACLE_H
#define GMG_ORACLE_H

#include <stddef.h>
#include "../include/mg_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* 0 if the descriptor is valid, otherwise a negative code (same codes as mg_hip.h). */
int orc_validate(const mg_desc *d);

/* nodes per side (x, y) of level l (src/domain.cpp:9-12) and its number of planes */
int orc_level_n(const mg_desc *d, int level);
int orc_level_nz(const mg_desc *d, int level);

/* out[0..3] = {cx, cy, cz, cd} of level l in double precision
 * (include/linear_system.hpp:17,27-28,37-38; include/domain.hpp:90; src/domain.cpp:5). */
void orc_level_coefficients(const mg_desc *d, int level, double out[4]);

/* DataVector<T> (include/linear_system.hpp:85-92) with the (f,g) table of
 * src/utilities.cpp:138-147: b = g on boundary nodes, f inside; node (j,i) sits at
 * x = i*h, y = length - j*h (include/domain.hpp:68). test outside 0..2 -> pair 0. */
void orc_fill_rhs_2d(int n, double length, int test, double *b);

/* 3-D right-hand sides (EXTENSION): kind 0 = manufactured
 * f = 3*pi^2*alpha/len^2 * sin(pi x/len) sin(pi y/len) sin(pi z/len), g = 0;
 * kind 1 = counter-based hash noise in [-1,1) of the linear index, g = 0. */
void orc_fill_rhs_3d(int n, double length, double alpha, int kind, unsigned long long seed,
                     double *b);
/* the exact solution of kind 0, for O(h^2) checks */
void orc_exact_3d(int n, double length, double *u);

/* ---- single operators, both precisions (bodies: gmg_ops.inc) ---- */
typedef struct orc_coef_f64 { double cx, cy, cz, cd; } orc_coef_f64;
typedef struct orc_coef_f32 { float cx, cy, cz, cd; } orc_coef_f32;

/* every level is (dim, n, nz): n nodes per side in x/y, nz planes (1 in 2-D); transfers take the
 * COARSE level's (nc, nzc) and `semi` (1 = z not coarsened) */
#define ORC_DECL_OPS(REAL, SUF)                                                              \
    void orc_jacobi_##SUF(int dim, int n, int nz, orc_coef_##SUF c, REAL omega, const REAL *u, \
                          const REAL *rhs, REAL *unew);                                      \
    void orc_gs_lex_##SUF(int dim, int n, int nz, orc_coef_##SUF c, REAL *u, const REAL *rhs); \
    void orc_rbgs_##SUF(int dim, int n, int nz, orc_coef_##SUF c, REAL *u, const REAL *rhs); \
    double orc_residual_##SUF(int dim, int n, int nz, orc_coef_##SUF c, const REAL *u,       \
                              const REAL *rhs, REAL *r);                                     \
    void orc_residual_vec_##SUF(int dim, int n, int nz, orc_coef_##SUF c, const REAL *u,     \
                                const REAL *rhs, REAL *r);                                   \
    double orc_sumsq_##SUF(size_t count, const REAL *v);                                     \
    void orc_inject_##SUF(int dim, int nc, int nzc, int semi, const REAL *fine, REAL *coarse); \
    void orc_restrict_fw_##SUF(int dim, int nc, int nzc, int semi, const REAL *fine, REAL *coarse); \
    void orc_prolong_overwrite_##SUF(int dim, int nc, int nzc, int semi, const REAL *coarse, REAL *fine); \
    void orc_prolong_add_##SUF(int dim, int nc, int nzc, int semi, const REAL *coarse, REAL *fine, \
                               REAL *scratch);                                               \
    void orc_correct_##SUF(size_t count, REAL *u, REAL *e);                                  \
    void orc_smooth_##SUF(int smoother, int dim, int n, int nz, orc_coef_##SUF c, REAL omega, \
                          int sweeps, REAL *u, const REAL *rhs, REAL *tmp);                  \
    int orc_coarse_solve_##SUF(int smoother, int dim, int n, int nz, orc_coef_##SUF c, REAL omega, \
                               REAL *e, const REAL *rhs, REAL *tmp, int maxit, double tol,   \
                               int fixed, int *flag, double *relres);
ORC_DECL_OPS(double, f64)
ORC_DECL_OPS(float, f32)

/* ---- whole solver (hierarchy + cycle + outer loop), dtype taken from desc ---- */
typedef struct orc_mg orc_mg;
orc_mg *orc_mg_create(const mg_desc *d);
void orc_mg_destroy(orc_mg *m);
void orc_mg_set_rhs(orc_mg *m, const void *b);       /* finest grid, desc dtype */
void orc_mg_set_solution(orc_mg *m, const void *u);
void orc_mg_get_solution(const orc_mg *m, void *u);
void orc_mg_get_residual(const orc_mg *m, void *r);  /* `res` of the last sawtooth cycle */
void orc_mg_cycle(orc_mg *m, mg_cycle_stats *st);
/* returns the number of history entries produced (hist[0] = initial relative residual) */
int orc_mg_solve(orc_mg *m, double tol, int maxit, double *hist, int hist_cap,
                 mg_cycle_stats *per_cycle);
/* one smoother sweep / residual on the finest grid (bench cpu_baseline leg) */
void orc_mg_smooth_fine(orc_mg *m, int smoother, int sweeps);
double orc_mg_residual_fine(orc_mg *m);

int orc_omp_threads(void);

#ifdef __cplusplus
}
#endif
#endif
