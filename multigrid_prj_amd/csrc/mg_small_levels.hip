// mg_small_levels.hip -- the V(2,2) Jacobi cycle on levels too narrow for the row-wide fused kernels (65^3 and below), gfx950.
//
// Those levels are launch-latency bound: at 65^3 a sweep is 5 us of which 2 are arithmetic, and a cycle spends six launches
// on each of them (zero-guess sweep, sweep, residual + restriction; prolongation, sweep, sweep). Here a workgroup owns a
// brick of 2CB x 2CB x 2CB fine points (CB^3 coarse points), stages what the brick needs in LDS once and runs the whole
// chain on it, recomputing the overlap with its neighbours (a halo of 2-3 points: redundant arithmetic is free here):
//   k_small_pre_rr       u = J(J(0)),  coarse_rhs = R(rhs - A u)        (one launch instead of three)
//   k_small_prolong_post out = J(J(u + P e))                            (one launch instead of three)
// Per point the expressions are those of k_sweep3d / k_resid_restrict_fw / k_prolong3d_fast (same operand order, same
// correctly rounded division by the diagonal), so a cycle keeps its bits -- the V-cycle parity tests run through these
// kernels on every hierarchy that has such levels. Reference counterparts: Jacobi_iteration::apply_iteration_to_vec
// (include/solvers.hpp:64-83), Residual (solvers.hpp:265-273), InterpolationClass::interpolate (src/multigrid.cpp:3-27).
#include "mg_kernels.h"

#include <cstdlib>

namespace mg {
namespace {

__device__ __forceinline__ long long gidx(const Geom &g, int z, int y, int x)
{
    return (long long)z * g.plane + (long long)y * g.pitch + x;
}

// one Jacobi update of a point from its six neighbours (k_sweep3d's expression)
template <typename T, bool DAMPED>
__device__ __forceinline__ T jacobi_point(const Coef<T> &c, T omega, T b, T uc, T zm, T ym, T xm, T xp, T yp, T zp)
{
    T sum = 0;
    sum += c.cz * zm;
    sum += c.cy * ym;
    sum += c.cx * xm;
    sum += c.cx * xp;
    sum += c.cy * yp;
    sum += c.cz * zp;
    const T num = b - sum;
    T jac = div_cd<T>(num, c);
    if (DAMPED) jac = uc + omega * (jac - uc);
    return jac;
}

constexpr int SMALL_THREADS = 256;

// u = J(J(0)) on the brick's fine points, coarse = R(rhs - A u) on its coarse points.
// Regions (per axis, f0 = 2 K0 = first owned fine index): v = J(0) on [f0-3, f0+FB+1], w = J(v) on [f0-2, f0+FB],
// r = rhs - A w on [f0-1, f0+FB-1], coarse points K0 .. K0+CB-1.
template <typename T, bool DAMPED, int CB>
__global__ __launch_bounds__(SMALL_THREADS) void k_small_pre_rr(Geom gf, Geom gc, Coef<T> c, T omega, const T *__restrict__ rhs,
                                                                 T *__restrict__ u_out, T *__restrict__ coarse, int nbx, int nby)
{
    constexpr int FB = 2 * CB, NV = FB + 5, NW = FB + 3, NR = FB + 1;
    __shared__ T sb[NV * NV * NV];   // rhs on the v region
    __shared__ T sv[NV * NV * NV];
    __shared__ T sw[NW * NW * NW];
    __shared__ T sr[NR * NR * NR];
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const int fx0 = FB * bx, fy0 = FB * by, fz0 = FB * bz;
    const int tid = threadIdx.x;
    auto inside = [&](int z, int y, int x) { return z >= 0 && z < gf.nz && y >= 0 && y < gf.ny && x >= 0 && x < gf.nx; };
    auto on_bnd = [&](int z, int y, int x) {
        return z == 0 || z == gf.nz - 1 || y == 0 || y == gf.ny - 1 || x == 0 || x == gf.nx - 1;
    };
    // first sweep from the zero guess: v = rhs on Dirichlet nodes, 0 + omega (rhs / cd - 0) inside (k_sweep3d<ZEROU>)
    // (every global load of the phase is issued before the first use: one memory round trip per phase, not one per element)
    constexpr int IT1 = (NV * NV * NV + SMALL_THREADS - 1) / SMALL_THREADS;
    T bl[IT1];
#pragma unroll
    for (int k = 0; k < IT1; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % NV, ly = (i / NV) % NV, lz = i / (NV * NV);
        const int x = fx0 - 3 + lx, y = fy0 - 3 + ly, z = fz0 - 3 + lz;
        bl[k] = (i < NV * NV * NV && inside(z, y, x)) ? rhs[gidx(gf, z, y, x)] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < IT1; k++) {
        const int i = tid + k * SMALL_THREADS;
        if (i >= NV * NV * NV) break;
        const int lx = i % NV, ly = (i / NV) % NV, lz = i / (NV * NV);
        const int x = fx0 - 3 + lx, y = fy0 - 3 + ly, z = fz0 - 3 + lz;
        const T b = bl[k];
        T v = 0;
        if (inside(z, y, x)) {
            const T zero = 0;
            v = on_bnd(z, y, x) ? b : jacobi_point<T, DAMPED>(c, omega, b, zero, zero, zero, zero, zero, zero, zero);
        }
        sb[i] = b; sv[i] = v;
    }
    __syncthreads();
    // second sweep: w = J(v); the brick's own points go to u_out
    for (int i = tid; i < NW * NW * NW; i += SMALL_THREADS) {
        const int lx = i % NW, ly = (i / NW) % NW, lz = i / (NW * NW);
        const int x = fx0 - 2 + lx, y = fy0 - 2 + ly, z = fz0 - 2 + lz;
        T w = 0;
        if (inside(z, y, x)) {
            const int j = ((lz + 1) * NV + (ly + 1)) * NV + (lx + 1);   // the same point in the v region
            const T b = sb[j];
            w = on_bnd(z, y, x) ? b
                                : jacobi_point<T, DAMPED>(c, omega, b, sv[j], sv[j - NV * NV], sv[j - NV], sv[j - 1], sv[j + 1],
                                                          sv[j + NV], sv[j + NV * NV]);
            if (lx >= 2 && lx < 2 + FB && ly >= 2 && ly < 2 + FB && lz >= 2 && lz < 2 + FB) u_out[gidx(gf, z, y, x)] = w;
        }
        sw[i] = w;
    }
    __syncthreads();
    // residual of w (k_sweep3d<OP_RESIDUAL> / k_resid_restrict_fw: diagonal term in the middle of the sum)
    for (int i = tid; i < NR * NR * NR; i += SMALL_THREADS) {
        const int lx = i % NR, ly = (i / NR) % NR, lz = i / (NR * NR);
        const int x = fx0 - 1 + lx, y = fy0 - 1 + ly, z = fz0 - 1 + lz;
        T r = 0;
        if (inside(z, y, x)) {
            const int j = ((lz + 1) * NW + (ly + 1)) * NW + (lx + 1);   // the same point in the w region
            const T b = sb[((lz + 2) * NV + (ly + 2)) * NV + (lx + 2)];
            T sum = 0;
            sum += c.cz * sw[j - NW * NW];
            sum += c.cy * sw[j - NW];
            sum += c.cx * sw[j - 1];
            sum += c.cd * sw[j];
            sum += c.cx * sw[j + 1];
            sum += c.cy * sw[j + NW];
            sum += c.cz * sw[j + NW * NW];
            if (on_bnd(z, y, x)) sum = (T)1 * sw[j];
            r = b - sum;
        }
        sr[i] = r;
    }
    __syncthreads();
    // full weighting, axis by axis (x, then y, then z) like k_resid_restrict_fw; coarse boundary nodes inject
    const T q = (T)0.25, h = (T)0.5;
    for (int i = tid; i < CB * CB * CB; i += SMALL_THREADS) {
        const int cx_ = i % CB, cy_ = (i / CB) % CB, cz_ = i / (CB * CB);
        const int I = CB * bx + cx_, J = CB * by + cy_, K = CB * bz + cz_;
        if (I >= gc.nx || J >= gc.ny || K >= gc.nz) continue;
        const int j = ((2 * cz_ + 1) * NR + (2 * cy_ + 1)) * NR + (2 * cx_ + 1);   // fine point (2K, 2J, 2I) in the r region
        T val;
        if (I == 0 || I == gc.nx - 1 || J == 0 || J == gc.ny - 1 || K == 0 || K == gc.nz - 1) {
            val = sr[j];
        } else {
            T yw[3];
#pragma unroll
            for (int dz = 0; dz < 3; dz++) {
                T xw[3];
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int k = j + (dz - 1) * NR * NR + (dy - 1) * NR;
                    xw[dy] = q * sr[k - 1] + h * sr[k] + q * sr[k + 1];
                }
                yw[dz] = q * xw[0] + h * xw[1] + q * xw[2];
            }
            val = q * yw[0] + h * yw[1] + q * yw[2];
        }
        coarse[gidx(gc, K, J, I)] = val;
    }
}

// out = J(J(u + P e)) on the brick's fine points [f0, f0+FB-1]^3: w0 = u + P e on [f0-2, f0+FB+1], v = J(w0) on
// [f0-1, f0+FB], e (coarse) on [K0-1, K0+CB+1].
template <typename T, bool DAMPED, int CB>
__global__ __launch_bounds__(SMALL_THREADS) void k_small_prolong_post(Geom gf, Geom gc, Coef<T> c, T omega, const T *__restrict__ u,
                                                                       const T *__restrict__ e, const T *__restrict__ rhs,
                                                                       T *__restrict__ out, int nbx, int nby)
{
    constexpr int FB = 2 * CB, NW0 = FB + 4, NV = FB + 2, NE = CB + 3;
    __shared__ T se[NE * NE * NE];
    __shared__ T sw[NW0 * NW0 * NW0];
    __shared__ T sv[NV * NV * NV];
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const int fx0 = FB * bx, fy0 = FB * by, fz0 = FB * bz;
    const int ex0 = CB * bx - 1, ey0 = CB * by - 1, ez0 = CB * bz - 1;
    const int tid = threadIdx.x;
    auto inside = [&](int z, int y, int x) { return z >= 0 && z < gf.nz && y >= 0 && y < gf.ny && x >= 0 && x < gf.nx; };
    auto on_bnd = [&](int z, int y, int x) {
        return z == 0 || z == gf.nz - 1 || y == 0 || y == gf.ny - 1 || x == 0 || x == gf.nx - 1;
    };
    // every global value the brick needs is requested up front: e, u on the w0 region, rhs on the v region and on the brick
    constexpr int ITE = (NE * NE * NE + SMALL_THREADS - 1) / SMALL_THREADS, ITW = (NW0 * NW0 * NW0 + SMALL_THREADS - 1) / SMALL_THREADS;
    constexpr int ITV = (NV * NV * NV + SMALL_THREADS - 1) / SMALL_THREADS, ITO = (FB * FB * FB + SMALL_THREADS - 1) / SMALL_THREADS;
    T el[ITE], ul[ITW], bv[ITV], bo[ITO];
#pragma unroll
    for (int k = 0; k < ITE; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % NE, ly = (i / NE) % NE, lz = i / (NE * NE);
        const int x = ex0 + lx, y = ey0 + ly, z = ez0 + lz;
        el[k] = (i < NE * NE * NE && z >= 0 && z < gc.nz && y >= 0 && y < gc.ny && x >= 0 && x < gc.nx) ? e[gidx(gc, z, y, x)] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < ITW; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % NW0, ly = (i / NW0) % NW0, lz = i / (NW0 * NW0);
        const int x = fx0 - 2 + lx, y = fy0 - 2 + ly, z = fz0 - 2 + lz;
        ul[k] = (i < NW0 * NW0 * NW0 && inside(z, y, x)) ? u[gidx(gf, z, y, x)] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < ITV; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % NV, ly = (i / NV) % NV, lz = i / (NV * NV);
        const int x = fx0 - 1 + lx, y = fy0 - 1 + ly, z = fz0 - 1 + lz;
        bv[k] = (i < NV * NV * NV && inside(z, y, x)) ? rhs[gidx(gf, z, y, x)] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < ITO; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % FB, ly = (i / FB) % FB, lz = i / (FB * FB);
        const int x = fx0 + lx, y = fy0 + ly, z = fz0 + lz;
        bo[k] = (i < FB * FB * FB && inside(z, y, x)) ? rhs[gidx(gf, z, y, x)] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < ITE; k++) {
        const int i = tid + k * SMALL_THREADS;
        if (i < NE * NE * NE) se[i] = el[k];
    }
    __syncthreads();
    // w0 = u + P e: the prolongation's tree (k_prolong3d_fast: z midpoints, then y, then x, each 0.5 * (a + b))
    const T hf = (T)0.5;
#pragma unroll
    for (int k = 0; k < ITW; k++) {
        const int i = tid + k * SMALL_THREADS;
        if (i >= NW0 * NW0 * NW0) break;
        const int lx = i % NW0, ly = (i / NW0) % NW0, lz = i / (NW0 * NW0);
        const int x = fx0 - 2 + lx, y = fy0 - 2 + ly, z = fz0 - 2 + lz;
        T w = 0;
        if (inside(z, y, x)) {
            const int j = (((z >> 1) - ez0) * NE + ((y >> 1) - ey0)) * NE + ((x >> 1) - ex0);
            const bool pz = z & 1, py = y & 1, px = x & 1;
            T Y[2];
#pragma unroll
            for (int bxx = 0; bxx < 2; bxx++) {
                T Z[2];
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int k = j + a * NE + bxx;
                    Z[a] = pz ? hf * (se[k] + se[k + NE * NE]) : se[k];
                }
                Y[bxx] = py ? hf * (Z[0] + Z[1]) : Z[0];
            }
            const T pe = px ? hf * (Y[0] + Y[1]) : Y[0];
            w = ul[k] + pe;
        }
        sw[i] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITV; k++) {
        const int i = tid + k * SMALL_THREADS;
        if (i >= NV * NV * NV) break;
        const int lx = i % NV, ly = (i / NV) % NV, lz = i / (NV * NV);
        const int x = fx0 - 1 + lx, y = fy0 - 1 + ly, z = fz0 - 1 + lz;
        T v = 0;
        if (inside(z, y, x)) {
            const int j = ((lz + 1) * NW0 + (ly + 1)) * NW0 + (lx + 1);
            const T b = bv[k];
            v = on_bnd(z, y, x) ? b
                                : jacobi_point<T, DAMPED>(c, omega, b, sw[j], sw[j - NW0 * NW0], sw[j - NW0], sw[j - 1], sw[j + 1],
                                                          sw[j + NW0], sw[j + NW0 * NW0]);
        }
        sv[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITO; k++) {
        const int i = tid + k * SMALL_THREADS;
        const int lx = i % FB, ly = (i / FB) % FB, lz = i / (FB * FB);
        const int x = fx0 + lx, y = fy0 + ly, z = fz0 + lz;
        if (i >= FB * FB * FB || !inside(z, y, x)) continue;
        const int j = ((lz + 1) * NV + (ly + 1)) * NV + (lx + 1);
        const long long gi = gidx(gf, z, y, x);
        const T b = bo[k];
        out[gi] = on_bnd(z, y, x) ? b
                                  : jacobi_point<T, DAMPED>(c, omega, b, sv[j], sv[j - NV * NV], sv[j - NV], sv[j - 1], sv[j + 1],
                                                            sv[j + NV], sv[j + NV * NV]);
    }
}

constexpr int SMALL_CB = 4;

}  // namespace

// whole (non-distributed) 3-D levels joined by a standard coarsening, small enough to be launch-bound
template <typename T>
bool small_fused_ok(const Geom &gf, const Geom &gc)
{
    static const bool enabled = [] { const char *e = getenv("MG_SMALL_FUSED"); return !(e && e[0] == '0'); }();
    return enabled && gf.dim == 3 && gc.dim == 3 && gf.gz0 == 0 && gf.gnz == gf.nz && gc.gz0 == 0 && gc.gnz == gc.nz &&
           gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 && gf.nz == 2 * gc.nz - 1 && gc.nx >= 3 && gc.ny >= 3 && gc.nz >= 3 &&
           (long long)gf.nx * gf.ny * gf.nz <= 129LL * 129 * 129;
}

template <typename T>
void launch_small_pre_rr(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, T omega, const T *rhs, T *u_out, T *coarse)
{
    constexpr int CB = SMALL_CB;
    const int nbx = (gc.nx + CB - 1) / CB, nby = (gc.ny + CB - 1) / CB, nbz = (gc.nz + CB - 1) / CB;
    const dim3 gr(nbx * nby * nbz), bl(SMALL_THREADS);
    if (omega != (T)1) hipLaunchKernelGGL((k_small_pre_rr<T, true, CB>), gr, bl, 0, s, gf, gc, c, omega, rhs, u_out, coarse, nbx, nby);
    else hipLaunchKernelGGL((k_small_pre_rr<T, false, CB>), gr, bl, 0, s, gf, gc, c, omega, rhs, u_out, coarse, nbx, nby);
}

template <typename T>
void launch_small_prolong_post(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, T omega, const T *u, const T *e,
                               const T *rhs, T *out)
{
    constexpr int CB = SMALL_CB;
    const int nbx = (gc.nx + CB - 1) / CB, nby = (gc.ny + CB - 1) / CB, nbz = (gc.nz + CB - 1) / CB;
    const dim3 gr(nbx * nby * nbz), bl(SMALL_THREADS);
    if (omega != (T)1) hipLaunchKernelGGL((k_small_prolong_post<T, true, CB>), gr, bl, 0, s, gf, gc, c, omega, u, e, rhs, out, nbx, nby);
    else hipLaunchKernelGGL((k_small_prolong_post<T, false, CB>), gr, bl, 0, s, gf, gc, c, omega, u, e, rhs, out, nbx, nby);
}

template bool small_fused_ok<double>(const Geom &, const Geom &);
template bool small_fused_ok<float>(const Geom &, const Geom &);
template void launch_small_pre_rr<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, double, const double *, double *, double *);
template void launch_small_pre_rr<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, float, const float *, float *, float *);
template void launch_small_prolong_post<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, double, const double *, const double *, const double *, double *);
template void launch_small_prolong_post<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, float, const float *, const float *, const float *, float *);

}  // namespace mg
