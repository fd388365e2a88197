// mg_kernels.h -- launchers of the hand-written gfx950 kernels (mg_kernels.hip,
// mg_jacobi_fast.hip). All launchers only enqueue on `s`; none synchronises.
#ifndef MG_KERNELS_H
#define MG_KERNELS_H

#include <hip/hip_runtime.h>
#include "mg_geom.h"

namespace mg {

// number of per-block partial sums a reduction launch over `g` may produce
int reduce_partials_capacity(const Geom &g);

// zero_u: the caller guarantees u == 0 everywhere (fresh coarse-level initial guess); the fast
// path then reads nothing of u. The generic path needs u to really hold zeros.
template <typename T>
void launch_jacobi(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u,
                   const T *rhs, T *out, bool zero_u = false);

// finest-grid fast paths (mg_jacobi_fast.hip); launch_jacobi / launch_residual pick them
// automatically when fast_path_ok<T>(g)
template <typename T> bool fast_path_ok(const Geom &g);
template <typename T> int fast_partials_capacity(const Geom &g);
template <typename T>
void launch_jacobi_fast(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u,
                        const T *rhs, T *out, bool zero_u);
// two Jacobi sweeps in one pass (out = J(J(u))), see mg_jacobi_fast.hip
template <typename T> bool jacobi2_ok(const Geom &g);
// d_partials != nullptr: also sum (rhs - A u)^2 of the INPUT u, where the wide-tile kernel runs (returns the number of partial
// sums written there, 0 = no norm was computed)
template <typename T>
int launch_jacobi2(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u, const T *rhs, T *out,
                   bool zero_u = false, int dup_planes = 0, double *d_partials = nullptr);  // dup_planes > 0: the same geometry once more, that many planes further up, in the same launch
// z-slab of a distributed level: the pair on its inner planes (launch_jacobi2 with g = planes 1 .. nz-2), see pair_on_slab_t
template <typename T> bool jacobi2_slab_ok(const Geom &slab);
// the same with the V-cycle's prolong-add folded in: out = J(J(u + P coarse)); u is not modified
template <typename T> bool jacobi2_corr_ok(const Geom &gf, const Geom &gc);
template <typename T>
void launch_jacobi2_corr(hipStream_t s, const Geom &g, const Geom &gc, const Coef<T> &c, T omega, const T *u,
                         const T *coarse, const T *rhs, T *out, int dup_planes = 0);
// on the pieces of a z-slab: g = the piece, gc = the WHOLE coarse slab, coarse = its local plane 0 (two valid ghost planes either side)
template <typename T> bool jacobi2_corr_slab_ok(const Geom &gf, const Geom &gc);
// wide-tile form of the same pairs (mg_pair_wide.hip): rows of 128 / 256 lanes on levels big enough to fill the chip with
// 1024-thread workgroups; launch_jacobi2 / launch_jacobi2_corr / launch_rb_fused hand over to it when pair_wide_ok
// coarse != nullptr: out = pair(u + P coarse); zero_u: u == 0; rb: one red-black sweep instead of two Jacobi sweeps
template <typename T> bool pair_wide_ok(const Geom &g);
void set_pair_wide(int mode);   // measurement tools only (tools/pairbench.hip): 0 = never, 1 = wherever the shape allows, -1 = default
// d_partials != nullptr (plain Jacobi pair only): the launch also leaves one partial sum of (rhs - A u)^2 per workgroup
// there -- the residual norm of the pair's INPUT; returns how many (0: not computed)
template <typename T>
int launch_pair_wide(hipStream_t s, const Geom &g, const Geom &gc, const Coef<T> &c, T omega, const T *u, const T *coarse,
                     const T *rhs, T *out, bool zero_u, bool rb, int dup_planes, double *d_partials = nullptr);
// sum of n partial sums in a fixed order -> *d_out (mg_kernels.hip: k_reduce_final)
void launch_reduce_final(hipStream_t s, const double *d_partials, long long n, double *d_out);
// zebra line Gauss-Seidel along y: one colour pass; cp_den = the 2*ny factors of zebra_line_factors(cy, cd, ny) (device)
template <typename T>
void launch_zebra_y(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u, const T *rhs, T *dp,
                    const T *cp_den);
// the same with lines along x (k_zebra_x): cp_den = the 2*nx factors of zebra_line_factors(cx, cd, nx)
template <typename T>
void launch_zebra_x(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u, const T *rhs, T *dp,
                    const T *cp_den);
// elimination factors cp(j), den(j) of a line with off-diagonal cl and diagonal cd: out = 2 * n values (host)
template <typename T> void zebra_line_factors(T cl, T cd, int n, T *out);
// one whole red-black sweep in one pass (same gate as the fused double Jacobi sweep)
template <typename T> bool rb_fused_ok(const Geom &g);
template <typename T>
int launch_rb_fused(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs, T *out,
                    const T *coarse, const Geom &gc, int dup_planes = 0, bool zero_u = false, double *d_partials = nullptr);
// out-of-place colour half-sweep (the other colour is copied): red u->tmp, black tmp->u
template <typename T>
void launch_rb_fast(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, const T *u, const T *rhs, T *out);
template <typename T>
int launch_residual_fast(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs,
                         T *r, double *d_partials, bool want_norm);

// 3-D prolongation fast path (mg_transfer_fast.hip)
template <typename T> bool prolong_fast_ok(const Geom &gc, const Geom &gf);
template <typename T>
void launch_prolong_fast(hipStream_t s, const Geom &gc, const Geom &gf, const T *coarse, T *fine, bool add);

// fused residual + full-weighting restriction (non-distributed 3-D levels): coarse = R (rhs - A u)
template <typename T> bool resid_restrict_fast_ok(const Geom &gf, const Geom &gc);
// the same launcher on a z-slab (two ghost planes of u and one of rhs below the slab must be valid)
template <typename T> bool resid_restrict_slab_ok(const Geom &gf, const Geom &gc);
template <typename T>
void launch_resid_restrict_fw(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, const T *u,
                              const T *rhs, T *coarse, int dup_kc = 0, int dup_nzf = 0);
// dup_kc > 0 (gc.nz must be 1): a second single coarse plane dup_kc coarse planes further up (its fine planes start 2 dup_kc
// further up and there are dup_nzf of them) in the same launch -- the two boundary pieces of a z-slab

// wide-tile form of the same operator (mg_rr_wide.hip): rows of 128 / 256 lanes; launch_resid_restrict_fw hands over to it
template <typename T> bool rr_wide_ok(const Geom &gf, const Geom &gc);
template <typename T>
void launch_rr_wide(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, const T *u, const T *rhs, T *coarse,
                    int dup_kc, int dup_nzf);
void set_rr_wide(int mode);   // measurement tools only: 0 = never, 1 = wherever the shape allows, -1 = default

// launch-bound levels (65^3 and below), V(2,2) Jacobi, whole 3-D levels (mg_small_levels.hip): the three launches either side
// of the coarser levels in one each -- u_out = J(J(0)), coarse = R(rhs - A u_out)  /  out = J(J(u + P e))
template <typename T> bool small_fused_ok(const Geom &gf, const Geom &gc);
template <typename T>
void launch_small_pre_rr(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, T omega, const T *rhs, T *u_out, T *coarse);
template <typename T>
void launch_small_prolong_post(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, T omega, const T *u, const T *e,
                               const T *rhs, T *out);

// one colour half-sweep of red-black Gauss-Seidel, in place
template <typename T>
void launch_rbgs_colour(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u,
                        const T *rhs);

// lexicographic Gauss-Seidel, one persistent workgroup, diagonal wavefronts
template <typename T>
void launch_gs_lex(hipStream_t s, const Geom &g, const Coef<T> &c, int sweeps, T *u,
                   const T *rhs);

// r = rhs - A u (r may be null), sum r^2 -> *d_sumsq (device double), two-pass
// deterministic reduction through d_partials
template <typename T>
void launch_residual(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs,
                     T *r, double *d_partials, double *d_sumsq);

template <typename T>
void launch_sumsq(hipStream_t s, const Geom &g, const T *v, double *d_partials, double *d_sumsq);

// coarse(K,J,I) = fine(2K,2J,2I)   (gc = coarse geometry, gf = fine geometry)
template <typename T>
void launch_inject(hipStream_t s, const Geom &gf, const Geom &gc, const T *fine, T *coarse);
template <typename T>
void launch_restrict_fw(hipStream_t s, const Geom &gf, const Geom &gc, const T *fine, T *coarse);

// fine = P coarse (add == false, overwrite) or fine += P coarse
template <typename T>
void launch_prolong(hipStream_t s, const Geom &gc, const Geom &gf, const T *coarse, T *fine,
                    bool add);

// u += e; e = 0
template <typename T>
void launch_correct(hipStream_t s, const Geom &g, T *u, T *e);

// Solver::Solve in one persistent workgroup; result vector ends in x
template <typename T>
void launch_coarse_solve(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, int smoother,
                         T *x, T *tmp, const T *rhs, int maxit, double tol, int fixed,
                         CoarseOut *d_out, bool x_is_zero = false);

}  // namespace mg
#endif
