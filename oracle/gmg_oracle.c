/*
 * gmg_oracle.c -- TEST INFRASTRUCTURE.  See gmg_oracle.h for scope and parity pin.
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off [-fopenmp]).
 * -ffp-contract=off keeps every a*b+c as two roundings, like the reference built
 * with g++ -O3 for baseline x86-64 (no FMA), so outputs are comparable bit for bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "gmg_oracle.h"

int orc_omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_validate(const mg_desc *d)
{
    if (!d) return -1;
    if (d->dim != 2 && d->dim != 3) return -2;
    if (d->levels < 1 || d->levels > ORC_MAX_LEVELS) return -3;
    if (d->n < 3) return -4;
    /* The reference does not validate this and reads out of range;
     * we refuse instead: every level must keep its last row on the boundary. */
    long step = 1L << (d->levels - 1);
    if ((d->n - 1) % step != 0) return -5;
    if ((d->n - 1) / step + 1 < 3) return -6;
    if (d->dtype != MG_F64 && d->dtype != MG_F32) return -7;
    if (d->smoother < MG_SMOOTH_GS_LEX || d->smoother > MG_SMOOTH_ZEBRA_X) return -8;
    if (d->cycle != MG_CYCLE_SAWTOOTH && d->cycle != MG_CYCLE_V) return -9;
    if (!(d->length > 0) || !(d->alpha > 0)) return -10;
    if (d->coarse_maxit < 0 || d->nu_pre < 0 || d->nu_post < 0 || d->outer_pre_gs < 0) return -11;
    for (int a = 0; a < 3; a++) if (!(d->aniso[a] > 0)) return -12;
    if (d->semi_xy < 0 || d->semi_xy > d->levels - 1) return -13;
    if (d->semi_xy && d->dim != 3) return -13;
    return 0;
}

int orc_level_n(const mg_desc *d, int level)
{
    int n = d->n;
    for (int l = 0; l < level; l++) n = (n + 1) / 2; /* src/domain.cpp:9-12 */
    return n;
}

int orc_level_nz(const mg_desc *d, int level)
{
    if (d->dim != 3) return 1;
    int nz = d->n;   /* the first semi_xy transitions keep z, the later ones halve it */
    for (int l = d->semi_xy; l < level; l++) nz = (nz + 1) / 2;
    return nz;
}

void orc_level_coefficients(const mg_desc *d, int level, double out[4])
{
    double m_h = d->length / (double)(d->n - 1); /* src/domain.cpp:5            */
    double step = (double)(1L << level);         /* src/domain.cpp:9-12         */
    double h = m_h * step;                       /* include/domain.hpp:90       */
    double k = h * h;                            /* include/linear_system.hpp:17 */
    double ax = d->aniso[0], ay = d->aniso[1], az = d->aniso[2];
    out[0] = -(d->alpha * ax) / k;               /* linear_system.hpp:37-38     */
    out[1] = -(d->alpha * ay) / k;
    out[2] = -(d->alpha * az) / k;
    double s = (d->dim == 3) ? (ax + ay + az) : (ax + ay);
    out[3] = ((2.0 * s) * d->alpha) / k;         /* linear_system.hpp:27-28: 4.*alpha/k */
    if (d->dim == 3 && d->semi_xy) {
        /* semi-coarsening (EXTENSION): z is coarsened only after the first semi_xy transitions */
        int lz = level > d->semi_xy ? level - d->semi_xy : 0;
        double hz = m_h * (double)(1L << lz);
        double kz = hz * hz;
        out[2] = -(d->alpha * az) / kz;
        out[3] = 2.0 * ((d->alpha * ax) / k + (d->alpha * ay) / k + (d->alpha * az) / kz);
    }
}

/* the (f,g) table of src/utilities.cpp:138-147 */
static double test_f(int t, double x, double y)
{
    switch (t) {
    case 1: return -5.0 * exp(x) * exp(-2.0 * y);
    case 2: { double r = sqrt(x * x + y * y);
              return r != 0.0 ? -30. * (cos(30. * r) / r - 30. * sin(30. * r)) : 0.0; }
    default: return 1.;
    }
}
static double test_g(int t, double x, double y)
{
    switch (t) {
    case 1: return exp(x) * exp(-2.0 * y);
    case 2: return sin(30. * sqrt(x * x + y * y));
    default: return 0.;
    }
}

void orc_fill_rhs_2d(int n, double length, int test, double *b)
{
    if (test < 0 || test > 2) test = 0; /* src/utilities.cpp:149-153 */
    double m_h = length / (double)(n - 1);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            double x = i * m_h, y = length - j * m_h; /* include/domain.hpp:68 */
            int bnd = (i == 0 || j == 0 || i == n - 1 || j == n - 1);
            b[(size_t)j * n + i] = bnd ? test_g(test, x, y) : test_f(test, x, y);
        }
}

static double hash_unit(unsigned long long seed, unsigned long long idx)
{
    unsigned long long z = seed + (idx + 1ULL) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

void orc_fill_rhs_3d(int n, double length, double alpha, int kind, unsigned long long seed,
                     double *b)
{
    const double pi = 3.14159265358979323846;
    double m_h = length / (double)(n - 1);
    double w = pi / length;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                size_t idx = ((size_t)k * n + j) * n + i;
                int bnd = (i == 0 || j == 0 || k == 0 || i == n - 1 || j == n - 1 || k == n - 1);
                if (bnd) { b[idx] = 0.; continue; }
                if (kind == 0)
                    b[idx] = 3.0 * alpha * w * w * sin(w * i * m_h) * sin(w * j * m_h) * sin(w * k * m_h);
                else
                    b[idx] = hash_unit(seed, idx);
            }
}

void orc_exact_3d(int n, double length, double *u)
{
    const double pi = 3.14159265358979323846;
    double m_h = length / (double)(n - 1);
    double w = pi / length;
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
                u[((size_t)k * n + j) * n + i] =
                    sin(w * i * m_h) * sin(w * j * m_h) * sin(w * k * m_h);
}

/* ---- typed operator bodies ---- */
#define REAL double
#define SUF f64
#include "gmg_ops.inc"
#include "gmg_cycle.inc"
#undef REAL
#undef SUF

#define REAL float
#define SUF f32
#include "gmg_ops.inc"
#include "gmg_cycle.inc"
#undef REAL
#undef SUF

/* ---- type-erased solver object ---- */
struct orc_mg {
    mg_desc d;
    hier_f64 *h64;
    hier_f32 *h32;
};

orc_mg *orc_mg_create(const mg_desc *d)
{
    if (orc_validate(d) != 0) return NULL;
    orc_mg *m = (orc_mg *)calloc(1, sizeof(*m));
    m->d = *d;
    if (d->dtype == MG_F64) m->h64 = hier_new_f64(d); else m->h32 = hier_new_f32(d);
    return m;
}

void orc_mg_destroy(orc_mg *m)
{
    if (!m) return;
    if (m->h64) hier_free_f64(m->h64);
    if (m->h32) hier_free_f32(m->h32);
    free(m);
}

void orc_mg_set_rhs(orc_mg *m, const void *b)
{
    if (m->h64) memcpy(m->h64->rhs[0], b, m->h64->cnt[0] * sizeof(double));
    else memcpy(m->h32->rhs[0], b, m->h32->cnt[0] * sizeof(float));
}
void orc_mg_set_solution(orc_mg *m, const void *u)
{
    if (m->h64) memcpy(m->h64->u[0], u, m->h64->cnt[0] * sizeof(double));
    else memcpy(m->h32->u[0], u, m->h32->cnt[0] * sizeof(float));
}
void orc_mg_get_solution(const orc_mg *m, void *u)
{
    if (m->h64) memcpy(u, m->h64->u[0], m->h64->cnt[0] * sizeof(double));
    else memcpy(u, m->h32->u[0], m->h32->cnt[0] * sizeof(float));
}
void orc_mg_get_residual(const orc_mg *m, void *r)
{
    if (m->h64) memcpy(r, m->h64->r0, m->h64->cnt[0] * sizeof(double));
    else memcpy(r, m->h32->r0, m->h32->cnt[0] * sizeof(float));
}
void orc_mg_cycle(orc_mg *m, mg_cycle_stats *st)
{
    if (m->h64) cycle_f64(m->h64, &m->d, st); else cycle_f32(m->h32, &m->d, st);
}
int orc_mg_solve(orc_mg *m, double tol, int maxit, double *hist, int hist_cap,
                 mg_cycle_stats *per_cycle)
{
    if (m->h64) return solve_f64(m->h64, &m->d, tol, maxit, hist, hist_cap, per_cycle);
    return solve_f32(m->h32, &m->d, tol, maxit, hist, hist_cap, per_cycle);
}
void orc_mg_smooth_fine(orc_mg *m, int smoother, int sweeps)
{
    if (m->h64) orc_smooth_f64(smoother, m->d.dim, m->h64->n[0], m->h64->nz[0], m->h64->coef[0], m->d.omega, sweeps,
                               m->h64->u[0], m->h64->rhs[0], m->h64->tmp[0]);
    else orc_smooth_f32(smoother, m->d.dim, m->h32->n[0], m->h32->nz[0], m->h32->coef[0], (float)m->d.omega, sweeps,
                        m->h32->u[0], m->h32->rhs[0], m->h32->tmp[0]);
}
double orc_mg_residual_fine(orc_mg *m)
{
    if (m->h64) return orc_residual_f64(m->d.dim, m->h64->n[0], m->h64->nz[0], m->h64->coef[0], m->h64->u[0],
                                        m->h64->rhs[0], NULL);
    return orc_residual_f32(m->d.dim, m->h32->n[0], m->h32->nz[0], m->h32->coef[0], m->h32->u[0],
                            m->h32->rhs[0], NULL);
}
