// mg_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// geometric-multigrid hot path.  Generic versions: correct for every level size,
// dimension and precision; the finest-grid fast paths live in mg_jacobi_fast.hip.
//
// Arithmetic contract (checked bit for bit against oracle/ by tests/):
//  * compiled with -ffp-contract=off: every product and sum rounds separately, in
//    the reference's order  (k-1),(j-1),(i-1),[c],(i+1),(j+1),(k+1)
//    (reference src/domain.cpp:36-38, include/solvers.hpp:36-46,70-80,265-273);
//  * true IEEE division by the diagonal, like `/ m_A.coeffRef(i,i)`;
//  * norms: per-thread double accumulation -> wave64 shuffle tree -> per-block
//    partial -> second kernel that adds the partials in a fixed order, so a norm
//    is reproducible run to run (no float atomics), but its summation order
//    differs from the reference's serial loop (tolerance 1e-12 relative in tests).
//
// Nothing here is a dense contraction: MFMA is deliberately unused; every kernel
// is priced against the HBM roofline.
#include "mg_kernels.h"

#include <algorithm>

namespace mg {

using std::max;

namespace {

constexpr int BX = 64;  // one wave64 spans 64 consecutive x
constexpr int BY = 4;   // 4 waves per workgroup, stacked in y

__device__ __forceinline__ long long lidx(const Geom &g, int z, int y, int x)
{
    return (long long)z * g.plane + (long long)y * g.pitch + x;
}

__device__ __forceinline__ bool on_boundary(const Geom &g, int z, int y, int x)
{
    // reference src/domain.cpp:20-23, extended to the slab-decomposed z axis
    bool b = (x == 0) | (y == 0) | (x == g.nx - 1) | (y == g.ny - 1);
    if (g.dim == 3) {
        int gz = g.gz0 + z;
        b |= (gz == 0) | (gz == g.gnz - 1);
    }
    return b;
}

template <typename T, int DIM>
__device__ __forceinline__ T offdiag_sum(const T *u, long long i, int pitch,
                                         long long plane, const Coef<T> &c)
{
    T sum = 0;
    if (DIM == 3) sum += c.cz * u[i - plane];
    sum += c.cy * u[i - pitch];
    sum += c.cx * u[i - 1];
    sum += c.cx * u[i + 1];
    sum += c.cy * u[i + pitch];
    if (DIM == 3) sum += c.cz * u[i + plane];
    return sum;
}

template <typename T, int DIM>
__device__ __forceinline__ T full_sum(const T *u, long long i, int pitch,
                                      long long plane, const Coef<T> &c)
{
    // Residual: diagonal included, in row order (solvers.hpp:269-271)
    T sum = 0;
    if (DIM == 3) sum += c.cz * u[i - plane];
    sum += c.cy * u[i - pitch];
    sum += c.cx * u[i - 1];
    sum += c.cd * u[i];
    sum += c.cx * u[i + 1];
    sum += c.cy * u[i + pitch];
    if (DIM == 3) sum += c.cz * u[i + plane];
    return sum;
}

template <typename T, int DIM, bool DAMPED>
__device__ __forceinline__ T point_update(const Geom &g, const Coef<T> &c, T omega,
                                          const T *u, const T *rhs, int z, int y, int x)
{
    long long i = lidx(g, z, y, x);
    T b = rhs[i];
    if (on_boundary(g, z, y, x)) return b;  // (b - 0) / 1
    T sum = offdiag_sum<T, DIM>(u, i, g.pitch, g.plane, c);
    T jac = div_cd<T>(b - sum, c);
    if (DAMPED) {
        T uc = u[i];
        return uc + omega * (jac - uc);
    }
    return jac;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sum over the workgroup; every thread gets the same value. sh: >= 18 doubles.
__device__ __forceinline__ double block_sum_bcast(double v, double *sh)
{
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nthreads = blockDim.x * blockDim.y * blockDim.z;
    const int nw = (nthreads + 63) >> 6;
    v = wave_sum(v);
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double s = 0;
        for (int w = 0; w < nw; w++) s += sh[w];
        sh[17] = s;
    }
    __syncthreads();
    double r = sh[17];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------- Jacobi (generic)
template <typename T, int DIM, bool DAMPED>
__global__ __launch_bounds__(BX *BY) void k_jacobi(Geom g, Coef<T> c, T omega,
                                                   const T *__restrict__ u,
                                                   const T *__restrict__ rhs, T *__restrict__ out)
{
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    out[lidx(g, z, y, x)] = point_update<T, DIM, DAMPED>(g, c, omega, u, rhs, z, y, x);
}

// ---------------------------------------------------------------- red-black GS
template <typename T, int DIM>
__global__ __launch_bounds__(BX *BY) void k_rbgs(Geom g, Coef<T> c, int colour, T *u,
                                                 const T *__restrict__ rhs)
{
    // each lane owns one point of the requested colour: x = 2*lane_x + offset
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (y >= g.ny) return;
    int par = (y + g.gz0 + z + colour) & 1;  // x parity that has (x+y+gz)&1 == colour
    int x = 2 * (blockIdx.x * BX + threadIdx.x) + par;
    if (x >= g.nx) return;
    u[lidx(g, z, y, x)] = point_update<T, DIM, false>(g, c, (T)1, u, rhs, z, y, x);
}

// ---------------------------------------------------------------- residual + norm
template <typename T, int DIM, bool SAVE, bool NORM>
__global__ __launch_bounds__(BX *BY) void k_residual(Geom g, Coef<T> c,
                                                     const T *__restrict__ u,
                                                     const T *__restrict__ rhs, T *__restrict__ r,
                                                     double *__restrict__ partials)
{
    __shared__ double sh[18];
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    double sq = 0.;
    if (x < g.nx && y < g.ny) {
        long long i = lidx(g, z, y, x);
        T sum;
        if (on_boundary(g, z, y, x)) sum = (T)1 * u[i];
        else sum = full_sum<T, DIM>(u, i, g.pitch, g.plane, c);
        T res = rhs[i] - sum;
        if (SAVE) r[i] = res;
        sq = (double)res * (double)res;
    }
    if (NORM) {
        double tot = block_sum_bcast(sq, sh);
        if (threadIdx.x == 0 && threadIdx.y == 0)
            partials[blockIdx.x + gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z)] = tot;
    }
}

template <typename T>
__global__ __launch_bounds__(BX *BY) void k_sumsq(Geom g, const T *__restrict__ v,
                                                  double *__restrict__ partials)
{
    __shared__ double sh[18];
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    double sq = 0.;
    if (x < g.nx && y < g.ny) {
        double t = (double)v[lidx(g, z, y, x)];
        sq = t * t;
    }
    double tot = block_sum_bcast(sq, sh);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        partials[blockIdx.x + gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z)] = tot;
}

// fixed-order final reduction: thread t adds partials t, t+1024, ... then the tree
__global__ __launch_bounds__(1024) void k_reduce_final(const double *__restrict__ partials,
                                                       long long n, double *__restrict__ out)
{
    __shared__ double sh[18];
    double s = 0.;
    for (long long i = threadIdx.x; i < n; i += 1024) s += partials[i];
    double tot = block_sum_bcast(s, sh);
    if (threadIdx.x == 0) *out = tot;
}

// ---------------------------------------------------------------- transfers
// semi != 0: semi-coarsening, planes map one to one (z is not coarsened)
template <typename T>
__global__ __launch_bounds__(BX *BY) void k_inject(Geom gf, Geom gc, int semi, const T *__restrict__ fine,
                                                   T *__restrict__ coarse)
{
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (x >= gc.nx || y >= gc.ny) return;
    int fz = (gc.dim == 3) ? (semi ? z : 2 * (gc.gz0 + z) - gf.gz0) : 0;
    coarse[lidx(gc, z, y, x)] = fine[lidx(gf, fz, 2 * y, 2 * x)];
}

// WZ: weights along z too (3-D standard coarsening); otherwise 9-point weights per plane
template <typename T, int DIM, bool WZ>
__global__ __launch_bounds__(BX *BY) void k_restrict_fw(Geom gf, Geom gc,
                                                        const T *__restrict__ fine,
                                                        T *__restrict__ coarse)
{
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (x >= gc.nx || y >= gc.ny) return;
    int fz = (DIM == 3) ? (WZ ? 2 * (gc.gz0 + z) - gf.gz0 : z) : 0;
    long long fi = lidx(gf, fz, 2 * y, 2 * x);
    T out;
    if (on_boundary(gc, z, y, x)) {
        out = fine[fi];
    } else {
        const T q = (T)0.25, hlf = (T)0.5;
        T zacc[3];
#pragma unroll
        for (int dz = 0; dz < (WZ ? 3 : 1); dz++) {
            long long oz = WZ ? (long long)(dz - 1) * gf.plane : 0;
            T yacc[3];
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const T *p = fine + fi + oz + (long long)(dy - 1) * gf.pitch;
                yacc[dy] = q * p[-1] + hlf * p[0] + q * p[1];
            }
            zacc[dz] = q * yacc[0] + hlf * yacc[1] + q * yacc[2];
        }
        out = WZ ? q * zacc[0] + hlf * zacc[1] + q * zacc[2] : zacc[0];
    }
    coarse[lidx(gc, z, y, x)] = out;
}

// Interpolated value at fine node (zf,yf,xf), built in the reference's phase order
// (src/multigrid.cpp:3-27; slow axis first, fast axis last) so that every fine node
// gets bit for bit what the in-place sequential phases produce.
// DIM == 23: 3-D semi-coarsening (planes map one to one, gzf is then the LOCAL plane index)
template <typename T, int DIM>
__device__ __forceinline__ T interp_z(const Geom &gc, const T *__restrict__ c, int gzf, int yc,
                                      int xc)
{
    if (DIM == 2) return c[lidx(gc, 0, yc, xc)];
    if (DIM == 23) return c[lidx(gc, gzf, yc, xc)];
    if ((gzf & 1) == 0) return c[lidx(gc, (gzf >> 1) - gc.gz0, yc, xc)];
    int k0 = ((gzf - 1) >> 1) - gc.gz0;
    return (T)0.5 * (c[lidx(gc, k0, yc, xc)] + c[lidx(gc, k0 + 1, yc, xc)]);
}
template <typename T, int DIM>
__device__ __forceinline__ T interp_y(const Geom &gc, const T *__restrict__ c, int gzf, int yf,
                                      int xc)
{
    if ((yf & 1) == 0) return interp_z<T, DIM>(gc, c, gzf, yf >> 1, xc);
    return (T)0.5 * (interp_z<T, DIM>(gc, c, gzf, (yf - 1) >> 1, xc) +
                     interp_z<T, DIM>(gc, c, gzf, (yf + 1) >> 1, xc));
}
template <typename T, int DIM, bool ADD>
__global__ __launch_bounds__(BX *BY) void k_prolong(Geom gc, Geom gf,
                                                    const T *__restrict__ coarse,
                                                    T *__restrict__ fine)
{
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (x >= gf.nx || y >= gf.ny) return;
    int gzf = (DIM == 23) ? z : gf.gz0 + z;
    T v;
    if ((x & 1) == 0) v = interp_y<T, DIM>(gc, coarse, gzf, y, x >> 1);
    else v = (T)0.5 * (interp_y<T, DIM>(gc, coarse, gzf, y, (x - 1) >> 1) +
                       interp_y<T, DIM>(gc, coarse, gzf, y, (x + 1) >> 1));
    long long i = lidx(gf, z, y, x);
    if (ADD) fine[i] += v; else fine[i] = v;
}

template <typename T>
__global__ __launch_bounds__(BX *BY) void k_correct(Geom g, T *__restrict__ u, T *__restrict__ e)
{
    int x = blockIdx.x * BX + threadIdx.x;
    int y = blockIdx.y * BY + threadIdx.y;
    int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    long long i = lidx(g, z, y, x);
    u[i] += e[i];
    e[i] = 0;
}

// ---------------------------------------------------------------- single-workgroup sweeps
// Device-side sweeps executed by ONE workgroup of 1024 threads (coarsest grid,
// and the bit-faithful lexicographic GS on any level). Each ends with a barrier.
constexpr int SWG = 1024;

template <typename T, int DIM, bool DAMPED>
__device__ void wg_jacobi(const Geom &g, const Coef<T> &c, T omega, const T *u, const T *rhs,
                          T *out)
{
    const int npl = g.nx * g.ny;
    const long long total = (long long)npl * g.nz;
    for (long long q = threadIdx.x; q < total; q += SWG) {
        int z = (int)(q / npl);
        int rem = (int)(q - (long long)z * npl);
        int y = rem / g.nx, x = rem - y * g.nx;
        out[lidx(g, z, y, x)] = point_update<T, DIM, DAMPED>(g, c, omega, u, rhs, z, y, x);
    }
    __syncthreads();
}

template <typename T, int DIM>
__device__ void wg_rbgs(const Geom &g, const Coef<T> &c, T *u, const T *rhs)
{
    const int npl = g.nx * g.ny;
    const long long total = (long long)npl * g.nz;
    for (int colour = 0; colour < 2; colour++) {
        for (long long q = threadIdx.x; q < total; q += SWG) {
            int z = (int)(q / npl);
            int rem = (int)(q - (long long)z * npl);
            int y = rem / g.nx, x = rem - y * g.nx;
            if (((x + y + g.gz0 + z) & 1) != colour) continue;
            u[lidx(g, z, y, x)] = point_update<T, DIM, false>(g, c, (T)1, u, rhs, z, y, x);
        }
        __syncthreads();
    }
}

// Lexicographic Gauss-Seidel (solvers.hpp:33-48) as anti-diagonal wavefronts:
// inside a plane, point (y,x) needs the NEW (y-1,x),(y,x-1) and the OLD
// (y+1,x),(y,x+1); all points with x+y == d are independent once diagonal d-1 is
// done. Planes go in order (new z-1, old z+1). Same inputs per point as the
// serial loop => bit-identical result.
template <typename T, int DIM>
__device__ void wg_gs_lex(const Geom &g, const Coef<T> &c, T *u, const T *rhs)
{
    for (int z = 0; z < g.nz; z++) {
        for (int d = 0; d <= g.nx + g.ny - 2; d++) {
            int ylo = max(0, d - (g.nx - 1));
            int yhi = min(g.ny - 1, d);
            for (int y = ylo + (int)threadIdx.x; y <= yhi; y += SWG) {
                int x = d - y;
                u[lidx(g, z, y, x)] = point_update<T, DIM, false>(g, c, (T)1, u, rhs, z, y, x);
            }
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------- lexicographic GS, 2-D, row threads
// The two Gauss-Seidel pre-sweeps of every outer iteration (main.cpp:85,95) on the FINE grid.
// Same anti-diagonal wavefront as wg_gs_lex, organised so that the serial chain of nx+ny-1 steps
// never waits on memory: thread y owns row y and at step d updates x = d - y;
//   new u(y, x-1)  is its own previous result (register),
//   new u(y-1, x)  is the previous lane's previous result (DPP wave_shr:1; across waves through
//                  a two-slot LDS mailbox),
//   old u(y+1, x), old u(y, x+1) and rhs(y, x) are not on the chain and are fetched PF steps
//                  ahead (their stores happen >= PF+1 barriers after the loads have completed).
// One barrier per step with only ceil(ny/64) waves resident. Inputs and expression order per
// point are those of point_update => bit-identical to the serial loop (solvers.hpp:33-48).
template <typename T>
__device__ __forceinline__ T gs_prev_lane(T v, T edge);
template <>
__device__ __forceinline__ float gs_prev_lane<float>(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
template <>
__device__ __forceinline__ double gs_prev_lane<double>(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <typename T, int PF>
__global__ __launch_bounds__(SWG) void k_gs_lex2d_rows(Geom g, Coef<T> c, int sweeps, T *u, const T *rhs)
{
    __shared__ T mail[2][SWG / 64];
    const int y = threadIdx.x, lane = y & 63, wv = y >> 6;
    const int nx = g.nx, ny = g.ny;
    const bool rowin = y < ny;
    const bool rowb = (y == 0) || (y == ny - 1);
    const long long ro = (long long)min(y, ny - 1) * g.pitch;
    const long long rdn = (long long)min(y + 1, ny - 1) * g.pitch;  // the row below: still old when read
    const int nd = nx + ny - 1;                                      // anti-diagonals
    for (int s = 0; s < sweeps; s++) {
        T pb[PF], pdn[PF], prt[PF];
        // unconditional (clamped) loads: with loads under a branch the compiler cannot count how
        // many younger ones are in flight and falls back to s_waitcnt vmcnt(0) every PF steps
        auto fetch = [&](int d, T &b, T &dn, T &rt) {
            const int x = min(max(d - y, 0), nx - 1);
            b = rhs[ro + x];
            dn = u[rdn + x];
            rt = u[ro + min(x + 1, nx - 1)];
        };
#pragma unroll
        for (int j = 0; j < PF; j++) fetch(j, pb[j], pdn[j], prt[j]);
        T mine = 0;  // new u(y, x-1)
        T pub = 0;   // what this thread produced at the previous step: new u(y, x) of step d-1
        for (int d0 = 0; d0 < nd; d0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const int d = d0 + j, x = d - y;  // steps past nd-1 find every thread out of range
                T from_wave = 0;
                if (lane == 0 && wv > 0) from_wave = mail[(d + 1) & 1][wv - 1];
                const T upnew = gs_prev_lane<T>(pub, from_wave);
                T val = pub;
                if (rowin && x >= 0 && x < nx) {
                    if (rowb || x == 0 || x == nx - 1) {
                        val = pb[j];  // Dirichlet row of the matrix: (b - 0) / 1
                    } else {
                        T sum = 0;
                        sum += c.cy * upnew;
                        sum += c.cx * mine;
                        sum += c.cx * prt[j];
                        sum += c.cy * pdn[j];
                        val = div_cd<T>(pb[j] - sum, c);
                    }
                    u[ro + x] = val;
                    mine = val;
                }
                pub = val;
                if (lane == 63) mail[d & 1][wv] = val;
                fetch(d + PF, pb[j], pdn[j], prt[j]);
                // LDS-only release/acquire around the barrier: a full __syncthreads() would also
                // drain vmcnt, i.e. wait for the prefetches just issued, every step
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
        }
        __syncthreads();  // this sweep's stores are the next sweep's old values
    }
}

// Two lexicographic sweeps in ONE wavefront pass (the reference's `u * GS * GS`, main.cpp:85,95): the
// second sweep follows the first two anti-diagonals behind, so at step d thread y updates point
// x1 = d - y of sweep 1 and point x2 = d - 2 - y of sweep 2. Everything sweep 2 needs is one step old
// and lives in registers: new2(y-1, x2) and new1(y+1, x2) are the neighbouring lanes' results of the
// previous step (DPP, mailboxes across waves), new2(y, x2-1) and new1(y, x2+1) the thread's own.
// Sweep 1's values never go to memory: half the steps and half the stores of two separate sweeps.
template <typename T>
__device__ __forceinline__ T gs_next_lane(T v, T edge);
template <>
__device__ __forceinline__ float gs_next_lane<float>(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
template <>
__device__ __forceinline__ double gs_next_lane<double>(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <typename T, int PF>
__global__ __launch_bounds__(SWG) void k_gs_lex2d_rows_pair(Geom g, Coef<T> c, T *u, const T *rhs)
{
    __shared__ T mail1[2][SWG / 64], mail2[2][SWG / 64], mailn[2][SWG / 64];
    const int y = threadIdx.x, lane = y & 63, wv = y >> 6, nwv = (int)(blockDim.x >> 6);
    const int nx = g.nx, ny = g.ny;
    const bool rowin = y < ny;
    const bool rowb = (y == 0) || (y == ny - 1);
    const long long ro = (long long)min(y, ny - 1) * g.pitch;
    const long long rdn = (long long)min(y + 1, ny - 1) * g.pitch;
    const int nd = nx + ny - 1 + 2;  // sweep 2 finishes two steps after sweep 1
    T pb[PF], pdn[PF], prt[PF];
    auto fetch = [&](int d, T &b, T &dn, T &rt) {  // sweep 1's operands that are not on the chain (old values)
        const int x = min(max(d - y, 0), nx - 1);
        b = rhs[ro + x];
        dn = u[rdn + x];
        rt = u[ro + min(x + 1, nx - 1)];
    };
#pragma unroll
    for (int j = 0; j < PF; j++) fetch(j, pb[j], pdn[j], prt[j]);
    T mine1 = 0, pub1 = 0;       // sweep 1: own previous result (= new1(y, x1-1)); published value
    T mine2 = 0, pub2 = 0;       // sweep 2 likewise
    T bh1 = 0, bh2 = 0;          // rhs(y, x1) of the last two steps: sweep 2 needs rhs(y, x2) = the one of step d-2
    for (int d0 = 0; d0 < nd; d0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int d = d0 + j, x1 = d - y, x2 = d - 2 - y;
            // neighbours' results of the previous step (read before anything of this step is published)
            T w1 = 0, w2 = 0, wn = 0;
            if (lane == 0 && wv > 0) { w1 = mail1[(d + 1) & 1][wv - 1]; w2 = mail2[(d + 1) & 1][wv - 1]; }
            if (lane == 63 && wv < nwv - 1) wn = mailn[(d + 1) & 1][wv + 1];
            const T up1 = gs_prev_lane<T>(pub1, w1);   // new1(y-1, x1)
            const T up2 = gs_prev_lane<T>(pub2, w2);   // new2(y-1, x2)
            const T dn2 = gs_next_lane<T>(pub1, wn);   // new1(y+1, x2): thread y+1's sweep-1 result of step d-1
            const T rt2 = pub1;                        // new1(y, x2+1): own sweep-1 result of step d-1
            const T b1 = pb[j];
            T val1 = pub1, val2 = pub2;
            if (rowin && x1 >= 0 && x1 < nx) {
                if (rowb || x1 == 0 || x1 == nx - 1) {
                    val1 = b1;
                } else {
                    T sum = 0;
                    sum += c.cy * up1;
                    sum += c.cx * mine1;
                    sum += c.cx * prt[j];
                    sum += c.cy * pdn[j];
                    val1 = div_cd<T>(b1 - sum, c);
                }
                mine1 = val1;
            }
            if (rowin && x2 >= 0 && x2 < nx) {
                if (rowb || x2 == 0 || x2 == nx - 1) {
                    val2 = bh2;
                } else {
                    T sum = 0;
                    sum += c.cy * up2;
                    sum += c.cx * mine2;
                    sum += c.cx * rt2;
                    sum += c.cy * dn2;
                    val2 = div_cd<T>(bh2 - sum, c);
                }
                u[ro + x2] = val2;
                mine2 = val2;
            }
            pub1 = val1; pub2 = val2;
            bh2 = bh1; bh1 = b1;
            if (lane == 63) { mail1[d & 1][wv] = val1; mail2[d & 1][wv] = val2; }
            if (lane == 0) mailn[d & 1][wv] = val1;
            fetch(d + PF, pb[j], pdn[j], prt[j]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
}

template <typename T, int DIM>
__device__ double wg_residual_sumsq(const Geom &g, const Coef<T> &c, const T *u, const T *rhs,
                                    double *sh)
{
    const int npl = g.nx * g.ny;
    const long long total = (long long)npl * g.nz;
    double sq = 0.;
    for (long long q = threadIdx.x; q < total; q += SWG) {
        int z = (int)(q / npl);
        int rem = (int)(q - (long long)z * npl);
        int y = rem / g.nx, x = rem - y * g.nx;
        long long i = lidx(g, z, y, x);
        T sum;
        if (on_boundary(g, z, y, x)) sum = (T)1 * u[i];
        else sum = full_sum<T, DIM>(u, i, g.pitch, g.plane, c);
        T res = rhs[i] - sum;
        sq += (double)res * (double)res;
    }
    return block_sum_bcast(sq, sh);
}

template <typename T>
__device__ double wg_sumsq(const Geom &g, const T *v, double *sh)
{
    const int npl = g.nx * g.ny;
    const long long total = (long long)npl * g.nz;
    double sq = 0.;
    for (long long q = threadIdx.x; q < total; q += SWG) {
        int z = (int)(q / npl);
        int rem = (int)(q - (long long)z * npl);
        int y = rem / g.nx, x = rem - y * g.nx;
        double t = (double)v[lidx(g, z, y, x)];
        sq += t * t;
    }
    return block_sum_bcast(sq, sh);
}

template <typename T, int DIM>
__global__ __launch_bounds__(SWG) void k_gs_lex(Geom g, Coef<T> c, int sweeps, T *u, const T *rhs)
{
    for (int s = 0; s < sweeps; s++) wg_gs_lex<T, DIM>(g, c, u, rhs);
}

// Solver::Solve (solvers.hpp:324-342) with (maxit, tol, step = 1) -- the whole
// iterate-to-tolerance loop in one launch: no host round trip per iteration.
// Exit condition reached by every wave: the loop variable `counter` is bounded
// by maxit and all threads see the same broadcast norm.
template <typename T, int DIM>
__global__ __launch_bounds__(SWG) void k_coarse_solve(Geom g, Coef<T> c, T omega, int smoother,
                                                      T *x, T *tmp, const T *rhs, int maxit,
                                                      double tol, int fixed, CoarseOut *out)
{
    __shared__ double sh[18];
    T *cur = x, *oth = tmp;
    const bool damped = (omega != (T)1);
    auto sweep = [&]() {
        if (smoother == 1) {
            if (damped) wg_jacobi<T, DIM, true>(g, c, omega, cur, rhs, oth);
            else wg_jacobi<T, DIM, false>(g, c, omega, cur, rhs, oth);
            T *t = cur; cur = oth; oth = t;
        } else if (smoother == 2) {
            wg_rbgs<T, DIM>(g, c, cur, rhs);
        } else {
            wg_gs_lex<T, DIM>(g, c, cur, rhs);
        }
    };
    double nb = wg_sumsq<T>(g, rhs, sh);  // refresh_normalization_constant, :244-254
    int iters = 0, flag = 0;
    double nr;
    if (fixed) {
        for (int s = 0; s < maxit; s++) sweep();
        iters = maxit;
        nr = wg_residual_sumsq<T, DIM>(g, c, cur, rhs, sh);
    } else {
        int counter = maxit;
        nr = wg_residual_sumsq<T, DIM>(g, c, cur, rhs, sh);
        while (sqrt(nr / nb) > tol) {  // NaN (zero rhs) compares false, like the reference
            if (counter > 0) {
                sweep();
                counter -= 1;
                iters++;
                nr = wg_residual_sumsq<T, DIM>(g, c, cur, rhs, sh);
            } else {
                flag = 1;
                break;
            }
        }
    }
    if (cur != x) {  // odd number of Jacobi sweeps: move the result back into x
        const int npl = g.nx * g.ny;
        const long long total = (long long)npl * g.nz;
        for (long long q = threadIdx.x; q < total; q += SWG) {
            int z = (int)(q / npl);
            int rem = (int)(q - (long long)z * npl);
            int y = rem / g.nx, xx = rem - y * g.nx;
            long long i = lidx(g, z, y, xx);
            x[i] = cur[i];
        }
    }
    if (threadIdx.x == 0) {
        out->iters = iters;
        out->flag = flag;
        out->relres = sqrt(nr / nb);
        out->sumsq_rhs = nb;
        out->sumsq_r = nr;
    }
}

// ---------------------------------------------------------------- LDS-resident coarse solver
// Same loop as k_coarse_solve with the three coarse arrays (x, Jacobi temp, rhs) held in LDS
// for the whole solve: 17^3 or 65^2 doubles x 3 = 118 / 101 KB of the CU's 160 KB. Each
// thread owns <= PT fixed points whose dense indices and boundary flags are computed once,
// so an iteration is LDS loads + the stencil + one wave-shuffle reduction and two barriers:
// no global memory, no integer division, no launch per iteration.
template <typename T, int DIM, int PT>
__global__ __launch_bounds__(SWG) void k_coarse_solve_lds(Geom g, Coef<T> c, T omega, int smoother, T *x,
                                                          const T *rhs, int maxit, double tol, int fixed,
                                                          CoarseOut *out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double part[2][SWG / 64];
    const int npl = g.nx * g.ny;
    const int total = npl * g.nz;
    T *sx = reinterpret_cast<T *>(smem_raw);
    T *st = sx + total;
    T *sr = st + total;
    Geom gl = g;  // dense LDS geometry
    gl.pitch = g.nx;
    gl.plane = npl;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int pidx[PT];
    bool pbnd[PT];
    int np = 0;
#pragma unroll
    for (int k = 0; k < PT; k++) {
        int q = tid + k * SWG;
        pidx[k] = 0; pbnd[k] = true;
        if (q < total) {
            int z = q / npl, rem = q - z * npl, y = rem / g.nx, xx = rem - y * g.nx;
            long long gi = lidx(g, z, y, xx);
            pidx[k] = q;
            pbnd[k] = on_boundary(g, z, y, xx);
            sx[q] = x[gi];
            sr[q] = rhs[gi];
            np = k + 1;
        }
    }
    int parity = 0;
    auto block_sum = [&](double v) -> double {
        v = wave_sum(v);
        if (lane == 0) part[parity][wv] = v;
        __syncthreads();
        double sum = 0;
#pragma unroll
        for (int w = 0; w < SWG / 64; w++) sum += part[parity][w];
        parity ^= 1;
        return sum;
    };
    const bool damped = (omega != (T)1);
    auto residual_sumsq = [&]() -> double {
        double sq = 0.;
#pragma unroll
        for (int k = 0; k < PT; k++) {
            if (k < np) {
                const int i = pidx[k];
                T sum;
                if (pbnd[k]) sum = (T)1 * sx[i];
                else sum = full_sum<T, DIM>(sx, i, gl.pitch, gl.plane, c);
                T res = sr[i] - sum;
                sq += (double)res * (double)res;
            }
        }
        return block_sum(sq);
    };
    auto sweep = [&]() {
        if (smoother == 1) {
#pragma unroll
            for (int k = 0; k < PT; k++) {
                if (k < np) {
                    const int i = pidx[k];
                    T b = sr[i], r = b;
                    if (!pbnd[k]) {
                        T sum = offdiag_sum<T, DIM>(sx, i, gl.pitch, gl.plane, c);
                        T jac = div_cd<T>(b - sum, c);
                        r = damped ? sx[i] + omega * (jac - sx[i]) : jac;
                    }
                    st[i] = r;
                }
            }
            __syncthreads();
            T *t = sx; sx = st; st = t;
        } else if (smoother == 2) {
            wg_rbgs<T, DIM>(gl, c, sx, sr);
        } else {
            wg_gs_lex<T, DIM>(gl, c, sx, sr);
        }
    };
    __syncthreads();
    double sqb = 0.;
#pragma unroll
    for (int k = 0; k < PT; k++)
        if (k < np) { double t = (double)sr[pidx[k]]; sqb += t * t; }
    const double nb = block_sum(sqb);
    int iters = 0, flag = 0;
    double nr;
    if (fixed) {
        for (int s = 0; s < maxit; s++) sweep();
        iters = maxit;
        nr = residual_sumsq();
    } else {
        int counter = maxit;
        nr = residual_sumsq();
        while (sqrt(nr / nb) > tol) {
            if (counter > 0) {
                sweep();
                counter -= 1;
                iters++;
                nr = residual_sumsq();
            } else {
                flag = 1;
                break;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PT; k++) {
        int q = tid + k * SWG;
        if (q < total) {
            int z = q / npl, rem = q - z * npl, y = rem / g.nx, xx = rem - y * g.nx;
            x[lidx(g, z, y, xx)] = sx[q];
        }
    }
    if (tid == 0) {
        out->iters = iters;
        out->flag = flag;
        out->relres = sqrt(nr / nb);
        out->sumsq_rhs = nb;
        out->sumsq_r = nr;
    }
}


// ---------------------------------------------------------------- LDS coarse solver, Jacobi
// The reference's dominant loop (Solver::Solve with a Jacobi smoother on the coarsest level,
// solvers.hpp:324-342: ~1900 iterations per cycle at 65^2) restructured around what an
// iteration really needs:
//  * a thread owns a run of SEG consecutive interior x-points of one row: its own values and right-hand
//    sides stay in registers, x-neighbours inside the run are registers too, so an iteration
//    reads 2 (4 in 3-D) row neighbours per point + the 2 run ends from LDS instead of 11 (15);
//  * residual(x_k) and the sweep x_k -> x_{k+1} read the same neighbours: one pass computes the
//    residual's sum of squares AND the tentative next iterate, which is stored to the other LDS
//    buffer before the reduction barrier -- ONE barrier per iteration; if the norm test says
//    stop, the tentative iterate is simply dropped;
//  * the norm test sqrt(nr/nb) > tol is decided by nr vs tol^2 nb whenever they differ by more
//    than 1e-9 relative (the exact expression cannot disagree there) and by the exact expression
//    otherwise, so the stopping iteration is the reference's;
//  * the wave reduction runs on DPP (no LDS traffic).
// Per-point arithmetic is that of wg_jacobi / the residual: x is bit-identical to the generic
// loop after the same number of iterations (tests/test_gpu_parity.py::test_coarse_solver).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, returned to every lane (fixed order: deterministic)
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    v += dpp_shifted<0x111, 0xf>(v);  // row_shr:1
    v += dpp_shifted<0x112, 0xf>(v);  // row_shr:2
    v += dpp_shifted<0x114, 0xf>(v);  // row_shr:4
    v += dpp_shifted<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row's sum
    v += dpp_shifted<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_shifted<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// (runs of 7 / 8 points need more than the 128 VGPRs a 1024-thread workgroup leaves: those variants are limited to 512
// threads -- 504 at 65^2, 450 at 17^3 -- and spill nothing)
template <int SEG> struct CoarseRowsThreads { static constexpr int value = SEG >= 7 ? 512 : SWG; };

template <typename T, int DIM, int SEG>
__global__ __launch_bounds__(CoarseRowsThreads<SEG>::value) void k_coarse_jacobi_rows(Geom g, Coef<T> c, T omega, T *x, const T *rhs,
                                                            int maxit, double tol, int fixed, CoarseOut *out, int skip, int zero_x)
{
    // Threads own INTERIOR points only, in full runs of SEG:
    // the iteration body has no predicates and no boundary selects, so the SEG points of a thread
    // are one basic block whose LDS loads and arithmetic interleave. The Dirichlet nodes change
    // once (x <- b in the first sweep) and are handled by a strided loop in the first two trips,
    // which also leaves b in both LDS buffers; from then on their residual b - 1*b is a constant.
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double part[2][SWG / 64];
    const int nx = g.nx, ny = g.ny, npl = nx * ny, total = npl * g.nz;
    T *cur = reinterpret_cast<T *>(smem_raw);
    T *nxt = cur + total;
    T *chk = cur + 2 * total;  // (skip > 1 only) the iterate the current window of unchecked sweeps started from
    const int W = nx - 2, nseg = (W + SEG - 1) / SEG;
    const int irows = (ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    const int nthr = (int)blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
    const bool active = tid < nseg * irows;
    // consecutive lanes take consecutive ROWS of one run column: their LDS addresses are nx (odd)
    // elements apart -> conflict-free for any run length. When SEG does not divide the interior
    // width the last run is shifted left to stay full and overlaps its neighbour by ONE point
    // (the launcher admits nothing else): both owners compute and store the same bits, only the
    // shifted run leaves the shared point out of the norm.
    const int seg = active ? tid / irows : 0, row = active ? tid - seg * irows : 0;
    const int z = (DIM == 3) ? 1 + row / (ny - 2) : 0;
    const int y = 1 + ((DIM == 3) ? row % (ny - 2) : row);
    const int xs = min(seg * SEG, W - SEG);
    const bool dup0 = (xs != seg * SEG);  // own point 0 is also the previous run's last point
    const int x0 = 1 + xs;
    const int i0 = (z * ny + y) * nx + x0;  // dense LDS index of the first own point
    const bool damped = (omega != (T)1);
    auto dense_to_global = [&](int q) -> long long {
        const int zz = q / npl, rem = q - zz * npl, yy = rem / nx, xx = rem - yy * nx;
        return lidx(g, zz, yy, xx);
    };
    auto is_bnd = [&](int q) -> bool {
        const int zz = q / npl, rem = q - zz * npl, yy = rem / nx, xx = rem - yy * nx;
        return xx == 0 || xx == nx - 1 || yy == 0 || yy == ny - 1 || (DIM == 3 && (zz == 0 || zz == g.nz - 1));
    };
    // stage x into LDS; own values and right-hand sides into registers
    double sqb = 0.;
    for (int q = tid; q < total; q += nthr) {
        const long long gi = dense_to_global(q);
        cur[q] = zero_x ? (T)0 : x[gi];   // zero_x: the initial guess is zero (the caller skipped the memset of x)
        const double t = (double)rhs[gi];
        sqb += t * t;
    }
    T xv[SEG], bv[SEG];
#pragma unroll
    for (int k = 0; k < SEG; k++) {
        xv[k] = 0; bv[k] = 0;
        if (active) {
            const long long gi = lidx(g, z, y, x0 + k);
            xv[k] = zero_x ? (T)0 : x[gi]; bv[k] = rhs[gi];
        }
    }
    int parity = 0;
    auto block_sum = [&](double v) -> double {  // one barrier; also publishes the LDS stores made before it
        v = wave_sum_dpp(v);
        if (lane == 0) part[parity][wv] = v;
        __syncthreads();
        double sum = 0;
        for (int w = 0; w < nw; w++) sum += part[parity][w];
        parity ^= 1;
        return sum;
    };
    const double nb = block_sum(sqb);  // refresh_normalization_constant, solvers.hpp:244-254
    // sqrt(nr / nb) > tol, decided without the division and the root when it is not close
    const double t2 = tol * tol * nb;
    const bool pretest = (tol > 0) && (t2 > 1e-290) && (t2 < 1e290);
    const double t2_hi = t2 * (1. + 1e-9), t2_lo = t2 * (1. - 1e-9);
    auto above_tol = [&](double nr) -> bool {
        if (pretest) {
            if (nr > t2_hi) return true;
            if (nr < t2_lo) return false;
        }
        return sqrt(nr / nb) > tol;  // NaN (zero rhs) compares false, like the reference
    };
    // Checking the norm only every `skip` sweeps. The Jacobi iteration matrix of this operator (constant diagonal, identity
    // Dirichlet rows) is symmetric with spectrum inside (-1, 1), so the residual's 2-norm never grows from one sweep to the
    // next: if Norm() > tol holds at sweep k + skip it held at every sweep in between, and the reference's loop (one test
    // per sweep, solvers.hpp:324-342) would not have stopped there either. So: after an accepted, checked sweep save the
    // iterate, run skip-1 sweeps WITHOUT residual, norm and reduction (about half the work of a checked one), check again;
    // the first time the test says stop inside a window, go back to the saved iterate and walk the window one checked sweep
    // at a time -- the stopping sweep, the flag and the iterate are exactly the reference's. Fixed-sweep mode needs no test
    // at all until the last sweep.
    auto fast_sweep = [&]() {   // x <- J x : no residual, no norm; one barrier
        if (active) {
            const T el = cur[i0 - 1], er = cur[i0 + SEG];
            T num[SEG], quo[SEG];
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                const T left = (k == 0) ? el : xv[k > 0 ? k - 1 : 0];
                const T right = (k == SEG - 1) ? er : xv[k < SEG - 1 ? k + 1 : 0];
                T os = 0;
                if (DIM == 3) os += c.cz * cur[i0 + k - npl];
                os += c.cy * cur[i0 + k - nx];
                os += c.cx * left;
                os += c.cx * right;
                os += c.cy * cur[i0 + k + nx];
                if (DIM == 3) os += c.cz * cur[i0 + k + npl];
                num[k] = bv[k] - os;
            }
            div_cd_n<T, SEG>(num, quo, c);
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                xv[k] = damped ? xv[k] + omega * (quo[k] - xv[k]) : quo[k];
                nxt[i0 + k] = xv[k];
            }
        }
        __syncthreads();
        T *t_ = cur; cur = nxt; nxt = t_;
    };
    int iters = 0, flag = 0, counter = maxit;
    double nr, bnd_sq = 0.;  // bnd_sq: this thread's share of the Dirichlet nodes' r^2 (constant from trip 2 on)
    int iters_chk = 0;
    bool in_window = false, stepping = false;   // in_window: the last skip-1 sweeps were not checked; stepping: re-walking a window
    for (;;) {
        double sq = 0.;
        if (iters < 2) {  // uniform. Dirichlet nodes: r = b - 1*x, x <- b
            bnd_sq = 0.;
            for (int q = tid; q < total; q += nthr) {
                if (is_bnd(q)) {
                    const T bq = rhs[dense_to_global(q)];
                    const T res = bq - (T)1 * cur[q];
                    bnd_sq += (double)res * (double)res;
                    nxt[q] = bq;
                }
            }
        }
        sq = bnd_sq;
        T nv[SEG];
        if (active) {
            const T el = cur[i0 - 1], er = cur[i0 + SEG];
            T ym[SEG], yp[SEG], zm[SEG], zp[SEG], num[SEG], quo[SEG];
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                ym[k] = cur[i0 + k - nx]; yp[k] = cur[i0 + k + nx];
                zm[k] = 0; zp[k] = 0;
                if (DIM == 3) { zm[k] = cur[i0 + k - npl]; zp[k] = cur[i0 + k + npl]; }
            }
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                const T left = (k == 0) ? el : xv[k > 0 ? k - 1 : 0];
                const T right = (k == SEG - 1) ? er : xv[k < SEG - 1 ? k + 1 : 0];
                T fs = 0;  // residual row, diagonal included (solvers.hpp:269-271)
                if (DIM == 3) fs += c.cz * zm[k];
                fs += c.cy * ym[k];
                fs += c.cx * left;
                fs += c.cd * xv[k];
                fs += c.cx * right;
                fs += c.cy * yp[k];
                if (DIM == 3) fs += c.cz * zp[k];
                const T res = bv[k] - fs;
                const double r2 = (double)res * (double)res;
                sq += (k == 0 && dup0) ? 0. : r2;
                T os = 0;  // Jacobi row, off-diagonals only (solvers.hpp:72-79)
                if (DIM == 3) os += c.cz * zm[k];
                os += c.cy * ym[k];
                os += c.cx * left;
                os += c.cx * right;
                os += c.cy * yp[k];
                if (DIM == 3) os += c.cz * zp[k];
                num[k] = bv[k] - os;
            }
            div_cd_n<T, SEG>(num, quo, c);
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                nv[k] = damped ? xv[k] + omega * (quo[k] - xv[k]) : quo[k];
                nxt[i0 + k] = nv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < SEG; k++) nv[k] = 0;
        }
        nr = block_sum(sq);
        bool go;
        if (fixed) go = iters < maxit;
        else if (above_tol(nr)) { go = counter > 0; if (!go) flag = 1; }
        else go = false;
        if (!go) {  // uniform: every thread sees the same nr
            if (!in_window) break;
            // the stop lies somewhere in the unchecked window: back to its first iterate, then one checked sweep at a time
            for (int q = tid; q < total; q += nthr) cur[q] = chk[q];
            if (active) {
#pragma unroll
                for (int k = 0; k < SEG; k++) xv[k] = chk[i0 + k];
            }
            counter += iters - iters_chk; iters = iters_chk; flag = 0;
            in_window = false; stepping = true;
            __syncthreads();
            continue;
        }
        counter -= 1;
        iters++;
#pragma unroll
        for (int k = 0; k < SEG; k++) xv[k] = nv[k];
        T *t = cur; cur = nxt; nxt = t;
        in_window = false;
        if (fixed) {   // no test until the end: after the two trips that settle the Dirichlet nodes in both buffers, every remaining sweep unchecked
            if (iters >= 2) while (iters < maxit) { fast_sweep(); iters++; counter--; }
        } else if (skip > 1 && !stepping && iters >= 2 && counter > skip) {
            for (int q = tid; q < total; q += nthr) chk[q] = cur[q];
            iters_chk = iters;
            __syncthreads();
            for (int s_ = 0; s_ < skip - 1; s_++) { fast_sweep(); iters++; counter--; }
            in_window = true;
        }
    }
    // cur holds the final iterate, Dirichlet nodes included
    for (int q = tid; q < total; q += nthr) x[dense_to_global(q)] = cur[q];
    if (tid == 0) {
        out->iters = iters;
        out->flag = flag;
        out->relres = sqrt(nr / nb);
        out->sumsq_rhs = nb;
        out->sumsq_r = nr;
    }
}

// ---------------------------------------------------------------- LDS coarse solver, red-black GS
// Solver::Solve (solvers.hpp:324-342) with the red-black smoother (BASELINE config 3's coarse solve: ~60 sweeps of 17^3 per
// cycle) on the row-segment layout of k_coarse_jacobi_rows: a thread owns a run of SEG interior points of one row, own values
// and right-hand sides in registers, two LDS copies of the iterate. A sweep is two colour phases -- every thread evaluates all
// its points and keeps those of the phase's colour ((x + y + z) & 1): the others' values are dropped -- and the residual of
// the previous iterate rides on the red phase (same neighbours): two LDS passes and two barriers per sweep (see the loop),
// where the generic loop (k_coarse_solve_lds) re-derives every point's indices in each of its three passes (7.6 us per
// sweep; this one ~2). Same point_update / residual expressions => the iterate is bit-identical after the same number of
// sweeps (tests/test_gpu_parity.py::test_coarse_solver). zero_x: the guess is zero and x has not been cleared.
template <typename T, int DIM, int SEG>
__global__ __launch_bounds__(CoarseRowsThreads<SEG>::value) void k_coarse_rb_rows(Geom g, Coef<T> c, T *x, const T *rhs, int maxit,
                                                                                  double tol, int fixed, CoarseOut *out, int zero_x)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double part[2][SWG / 64];
    const int nx = g.nx, ny = g.ny, npl = nx * ny, total = npl * g.nz;
    T *cur = reinterpret_cast<T *>(smem_raw);
    T *nxt = cur + total;
    const int W = nx - 2, nseg = (W + SEG - 1) / SEG;
    const int irows = (ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    const int nthr = (int)blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
    const bool active = tid < nseg * irows;
    const int seg = active ? tid / irows : 0, row = active ? tid - seg * irows : 0;
    const int z = (DIM == 3) ? 1 + row / (ny - 2) : 0;
    const int y = 1 + ((DIM == 3) ? row % (ny - 2) : row);
    const int xs = min(seg * SEG, W - SEG);
    const bool dup0 = (xs != seg * SEG);  // own point 0 is also the previous run's last point (both owners store the same bits)
    const int x0 = 1 + xs;
    const int i0 = (z * ny + y) * nx + x0;
    const int p0 = (x0 + y + z) & 1;      // colour of own point 0
    auto dense_to_global = [&](int q) -> long long {
        const int zz = q / npl, rem = q - zz * npl, yy = rem / nx, xx = rem - yy * nx;
        return lidx(g, zz, yy, xx);
    };
    // -1: interior node, else the colour of the Dirichlet node
    auto bnd_colour = [&](int q) -> int {
        const int zz = q / npl, rem = q - zz * npl, yy = rem / nx, xx = rem - yy * nx;
        const bool b = xx == 0 || xx == nx - 1 || yy == 0 || yy == ny - 1 || (DIM == 3 && (zz == 0 || zz == g.nz - 1));
        return b ? ((xx + yy + zz) & 1) : -1;
    };
    double sqb = 0.;
    for (int q = tid; q < total; q += nthr) {
        const long long gi = dense_to_global(q);
        const T xq = zero_x ? (T)0 : x[gi];
        cur[q] = xq; nxt[q] = xq;
        const double t = (double)rhs[gi];
        sqb += t * t;
    }
    T xv[SEG], bv[SEG];
#pragma unroll
    for (int k = 0; k < SEG; k++) {
        xv[k] = 0; bv[k] = 0;
        if (active) {
            const long long gi = lidx(g, z, y, x0 + k);
            xv[k] = zero_x ? (T)0 : x[gi]; bv[k] = rhs[gi];
        }
    }
    int parity = 0;
    auto block_sum = [&](double v) -> double {  // one barrier; also publishes the LDS stores made before it
        v = wave_sum_dpp(v);
        if (lane == 0) part[parity][wv] = v;
        __syncthreads();
        double sum = 0;
        for (int w = 0; w < nw; w++) sum += part[parity][w];
        parity ^= 1;
        return sum;
    };
    const double nb = block_sum(sqb);
    const double t2 = tol * tol * nb;
    const bool pretest = (tol > 0) && (t2 > 1e-290) && (t2 < 1e290);
    const double t2_hi = t2 * (1. + 1e-9), t2_lo = t2 * (1. - 1e-9);
    auto above_tol = [&](double nr) -> bool {
        if (pretest) {
            if (nr > t2_hi) return true;
            if (nr < t2_lo) return false;
        }
        return sqrt(nr / nb) > tol;  // NaN (zero rhs) compares false, like the reference
    };
    // One trip = the residual norm of the current iterate u_k AND sweep k+1, two LDS passes and two barriers:
    //   pass A (reads cur = u_k): r = b - A u_k on every own point (sum of squares -> the norm test) and, from the same
    //          neighbours, the red points of sweep k+1, stored to nxt; block_sum = barrier. If the test says stop, u_k is
    //          still whole in cur and the tentative reds are dropped (the reference tests before it sweeps, solvers.hpp:329);
    //   pass B (reads nxt): the black points from the new reds, stored to nxt; barrier; swap.
    // A black point's neighbours are all red and vice versa, so nxt is complete after pass B; values a pass computes on
    // points of the other colour (from mixed or stale neighbours) are dropped. The Dirichlet nodes turn into b during the
    // first sweep (both buffers get them in the first two trips); their residual is b - 1 * x before it and exactly 0 after.
    int iters = 0, flag = 0, counter = maxit;
    double nr;
    for (;;) {
        double sq = 0.;
        if (iters < 2) {
            for (int q = tid; q < total; q += nthr) {
                if (bnd_colour(q) >= 0) {
                    const T bq = rhs[dense_to_global(q)];
                    const T res = bq - (T)1 * cur[q];
                    sq += (double)res * (double)res;
                    nxt[q] = bq;
                }
            }
        }
        T nv[SEG];
#pragma unroll
        for (int k = 0; k < SEG; k++) nv[k] = xv[k];
        if (active) {
            const T el = cur[i0 - 1], er = cur[i0 + SEG];
            T ym[SEG], yp[SEG], zm[SEG], zp[SEG], num[SEG], quo[SEG];
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                ym[k] = cur[i0 + k - nx]; yp[k] = cur[i0 + k + nx];
                zm[k] = 0; zp[k] = 0;
                if (DIM == 3) { zm[k] = cur[i0 + k - npl]; zp[k] = cur[i0 + k + npl]; }
            }
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                const T left = (k == 0) ? el : xv[k > 0 ? k - 1 : 0];
                const T right = (k == SEG - 1) ? er : xv[k < SEG - 1 ? k + 1 : 0];
                T fs = 0;  // residual row, diagonal included (solvers.hpp:269-271)
                if (DIM == 3) fs += c.cz * zm[k];
                fs += c.cy * ym[k];
                fs += c.cx * left;
                fs += c.cd * xv[k];
                fs += c.cx * right;
                fs += c.cy * yp[k];
                if (DIM == 3) fs += c.cz * zp[k];
                const T res = bv[k] - fs;
                const double r2 = (double)res * (double)res;
                sq += (k == 0 && dup0) ? 0. : r2;
                T os = 0;  // Gauss-Seidel row, off-diagonals only (solvers.hpp:36-46)
                if (DIM == 3) os += c.cz * zm[k];
                os += c.cy * ym[k];
                os += c.cx * left;
                os += c.cx * right;
                os += c.cy * yp[k];
                if (DIM == 3) os += c.cz * zp[k];
                num[k] = bv[k] - os;
            }
            div_cd_n<T, SEG>(num, quo, c);
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                if (((p0 + k) & 1) == 0) { nv[k] = quo[k]; nxt[i0 + k] = quo[k]; }
            }
        }
        nr = block_sum(sq);
        bool go;
        if (fixed) go = iters < maxit;
        else if (above_tol(nr)) { go = counter > 0; if (!go) flag = 1; }
        else go = false;
        if (!go) break;   // uniform: every thread sees the same nr
#pragma unroll
        for (int k = 0; k < SEG; k++) xv[k] = nv[k];
        if (active) {   // black points from the new reds
            const T el = nxt[i0 - 1], er = nxt[i0 + SEG];
            T num[SEG], quo[SEG];
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                const T left = (k == 0) ? el : xv[k > 0 ? k - 1 : 0];
                const T right = (k == SEG - 1) ? er : xv[k < SEG - 1 ? k + 1 : 0];
                T os = 0;
                if (DIM == 3) os += c.cz * nxt[i0 + k - npl];
                os += c.cy * nxt[i0 + k - nx];
                os += c.cx * left;
                os += c.cx * right;
                os += c.cy * nxt[i0 + k + nx];
                if (DIM == 3) os += c.cz * nxt[i0 + k + npl];
                num[k] = bv[k] - os;
            }
            div_cd_n<T, SEG>(num, quo, c);
#pragma unroll
            for (int k = 0; k < SEG; k++) {
                if (((p0 + k) & 1) == 1) { xv[k] = quo[k]; nxt[i0 + k] = quo[k]; }
            }
        }
        __syncthreads();
        T *t_ = cur; cur = nxt; nxt = t_;
        counter -= 1;
        iters++;
    }
    __syncthreads();
    for (int q = tid; q < total; q += nthr) x[dense_to_global(q)] = cur[q];
    if (tid == 0) {
        out->iters = iters;
        out->flag = flag;
        out->relres = sqrt(nr / nb);
        out->sumsq_rhs = nb;
        out->sumsq_r = nr;
    }
}

// ---------------------------------------------------------------- LDS coarse solver, lexicographic GS (2-D)
// Solver::Solve with the Gauss-Seidel smoother (the reference's default, -smt 0) on the coarsest
// grid: iterate and right-hand side in LDS; the sweep is the anti-diagonal wavefront with one
// thread per ROW (thread y updates x = d - y at step d: the new left neighbour is its own previous
// result, the new upper neighbour was stored by thread y-1 one step earlier), so a step is four
// LDS reads, the update and one barrier among 4 waves instead of 16; the residual norm after the
// sweep is spread over all 256 threads. Same per-point inputs as the serial loop => same bits.
constexpr int CGS_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(CGS_THREADS) void k_coarse_gs_rows2d(Geom g, Coef<T> c, T *x, const T *rhs, int maxit,
                                                                  double tol, int fixed, CoarseOut *out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double part[2][CGS_THREADS / 64];
    const int nx = g.nx, ny = g.ny, total = nx * ny;
    T *sx = reinterpret_cast<T *>(smem_raw);
    T *sb = sx + total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double sqb = 0.;
    for (int q = tid; q < total; q += CGS_THREADS) {
        const int yy = q / nx, xx = q - yy * nx;
        const long long gi = lidx(g, 0, yy, xx);
        sx[q] = x[gi];
        const T b = rhs[gi];
        sb[q] = b;
        sqb += (double)b * (double)b;
    }
    int parity = 0;
    auto block_sum = [&](double v) -> double {
        v = wave_sum_dpp(v);
        if (lane == 0) part[parity][wv] = v;
        __syncthreads();
        double sum = 0;
#pragma unroll
        for (int w = 0; w < CGS_THREADS / 64; w++) sum += part[parity][w];
        parity ^= 1;
        return sum;
    };
    const double nb = block_sum(sqb);
    auto residual_sumsq = [&]() -> double {
        double sq = 0.;
        for (int q = tid; q < total; q += CGS_THREADS) {
            const int yy = q / nx, xx = q - yy * nx;
            T sum;
            if (xx == 0 || xx == nx - 1 || yy == 0 || yy == ny - 1) sum = (T)1 * sx[q];
            else sum = full_sum<T, 2>(sx, q, nx, 0, c);
            const T res = sb[q] - sum;
            sq += (double)res * (double)res;
        }
        return block_sum(sq);
    };
    const int y = tid;
    const bool rowin = y < ny, rowb = (y == 0) || (y == ny - 1);
    auto sweep = [&]() {
        T mine = 0;  // new x(y, xx-1)
        for (int d = 0; d <= nx + ny - 2; d++) {
            const int xx = d - y;
            if (rowin && xx >= 0 && xx < nx) {
                const int i = y * nx + xx;
                T val = sb[i];  // Dirichlet row of the matrix: (b - 0) / 1
                if (!(rowb || xx == 0 || xx == nx - 1)) {
                    T sum = 0;
                    sum += c.cy * sx[i - nx];
                    sum += c.cx * mine;
                    sum += c.cx * sx[i + 1];
                    sum += c.cy * sx[i + nx];
                    val = div_cd<T>(sb[i] - sum, c);
                }
                sx[i] = val;
                mine = val;
            }
            __syncthreads();
        }
    };
    int iters = 0, flag = 0;
    double nr;
    if (fixed) {
        for (int s = 0; s < maxit; s++) sweep();
        iters = maxit;
        nr = residual_sumsq();
    } else {
        int counter = maxit;
        nr = residual_sumsq();
        while (sqrt(nr / nb) > tol) {  // NaN (zero rhs) compares false, like the reference
            if (counter > 0) {
                sweep();
                counter -= 1;
                iters++;
                nr = residual_sumsq();
            } else {
                flag = 1;
                break;
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < total; q += CGS_THREADS) {
        const int yy = q / nx, xx = q - yy * nx;
        x[lidx(g, 0, yy, xx)] = sx[q];
    }
    if (tid == 0) {
        out->iters = iters;
        out->flag = flag;
        out->relres = sqrt(nr / nb);
        out->sumsq_rhs = nb;
        out->sumsq_r = nr;
    }
}

// ---------------------------------------------------------------- zebra line Gauss-Seidel along y
// One colour pass (EXTENSION). A thread owns one grid line
// (x, z) of the active colour and solves it with the Thomas algorithm: forward elimination up
// the line (dp into `dp`, the level's scratch array), back substitution down the line. Lanes are
// consecutive in x, so every step of the march is a coalesced access; lines run along y, which a
// z-slab decomposition never cuts. The elimination factors cp(j), den(j) are the same for every
// line and are tabulated once per level on the host. Same operations in the same order as the
// CPU restatement => same bits.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void k_zebra_y(Geom g, Coef<T> c, int colour, T *__restrict__ u,
                                                 const T *__restrict__ rhs, T *__restrict__ dp,
                                                 const T *__restrict__ cp, const T *__restrict__ den)
{
    constexpr int U = 8;  // rows per batch: their loads are independent of the recurrence and go out together
    const int z = blockIdx.y * 4 + threadIdx.y;
    if (z >= g.nz) return;
    const int gz = (DIM == 3) ? g.gz0 + z : 0;
    // only the lines of the active colour get a lane: x = 2 t + parity (a 256 x 1 workgroup spanning one
    // whole row per plane was no faster: 3.45 vs 3.28 ms per sweep at 513^3)
    const int x = 2 * (blockIdx.x * 64 + threadIdx.x) + ((colour + gz) & 1);
    if (x >= g.nx) return;
    const long long base = (long long)z * g.plane + x, sj = g.pitch;
    // dp is scratch: stored packed (x / 2), so that the active colour's lanes write and read whole lines
    // (in the interleaved u array every other element belongs to the other colour)
    const long long dbase = (long long)z * g.plane + (x >> 1);
    const int ny = g.ny;
    if (x == 0 || x == g.nx - 1 || (DIM == 3 && (gz == 0 || gz == g.gnz - 1))) {
        for (int j = 0; j < ny; j++) u[base + j * sj] = rhs[base + j * sj];  // identity rows: u = b / 1
        return;
    }
    T dprev = rhs[base];
    dp[dbase] = dprev;
    for (int j0 = 1; j0 < ny - 1; j0 += U) {
        T R[U], dn[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int j = min(j0 + k, ny - 2);  // clamped rows are computed and dropped
            const long long idx = base + j * sj;
            T S = 0;
            if (DIM == 3) S += c.cz * u[idx - g.plane];
            S += c.cx * u[idx - 1];
            S += c.cx * u[idx + 1];
            if (DIM == 3) S += c.cz * u[idx + g.plane];
            R[k] = rhs[idx] - S;
            dn[k] = den[j];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (j0 + k < ny - 1) {
                dprev = (R[k] - c.cy * dprev) / dn[k];
                dp[dbase + (j0 + k) * sj] = dprev;
            }
        }
    }
    T unext = rhs[base + (ny - 1) * sj];
    u[base + (ny - 1) * sj] = unext;
    for (int j0 = ny - 2; j0 >= 1; j0 -= U) {
        T D[U], C[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int j = max(j0 - k, 1);
            D[k] = dp[dbase + j * sj];
            C[k] = cp[j];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (j0 - k >= 1) {
                unext = D[k] - C[k] * unext;
                u[base + (j0 - k) * sj] = unext;
            }
        }
    }
    u[base] = dp[dbase];
}

// ---------------------------------------------------------------- zebra line Gauss-Seidel along x
// One colour pass, lines along the FAST axis (EXTENSION), coloured by the parity of y (+ z). A thread
// marching along its own line would use 8 bytes of every 128-byte line it touches, so ONE WAVE takes 64 lines of the
// active colour and walks them in chunks of XC = 8 columns (one 64-byte segment per line in fp64):
//   (A) the 64 lanes together evaluate R = b - S on the 64 x XC tile -- 8 consecutive lanes per line segment, so
//       every access is a whole aligned segment -- into LDS; the operands of chunk c+1 are loaded into registers
//       while chunk c's recurrence runs (they are consumed before the next prefetch overwrites them);
//   (B) lane l runs the elimination recurrence of line l over the chunk's columns out of LDS (rows padded to an odd
//       stride: conflict-free), carrying dp(x-1) in a register from chunk to chunk;
//   (C) the tile goes to the dp scratch array, coalesced again.
// The back substitution walks the chunks in reverse the same way. Per line the operations and their order are those of
// the CPU restatement's sequential Thomas solve => same bits; lines never cross a z-slab. (History: 256-line workgroups
// with the loads inside the tile loop 1.06 ms per colour pass at 513^3, batched loads and LDS-staged factors 1.0 ms --
// both latency-bound by their load / recurrence / store phases at two waves per SIMD; this one-wave pipeline: see DESIGN.md.)
constexpr int ZXT = 64;

template <typename T, int DIM>
__global__ __launch_bounds__(ZXT) void k_zebra_x(Geom g, Coef<T> c, int colour, T *__restrict__ u, const T *__restrict__ rhs,
                                                 T *__restrict__ dp, const T *__restrict__ cp, const T *__restrict__ den,
                                                 int lpp, int nlines)
{
    constexpr int XC = 8, LP = XC + 1;                    // columns per chunk (64-byte segments in fp64, 32-byte in fp32: 16 columns
                                                          // would need 197 VGPRs there); tile points per lane = XC
    constexpr int LPL = ZXT / XC;                         // lines covered by one wave-wide access
    __shared__ T tile[ZXT][LP];
    __shared__ long long lbase[ZXT];
    __shared__ int lflag[ZXT];  // 0: no such line, 1: interior line, 2: line inside the boundary (identity rows)
    __shared__ T sfac[XC];      // den(x) / cp(x) of the chunk
    const int t = threadIdx.x;
    {
        const int L = blockIdx.x * ZXT + t;  // line slot: z = L / lpp, y = 2 (L % lpp) + parity
        const int z = L / lpp, k = L - z * lpp;
        const int gz = (DIM == 3) ? g.gz0 + z : 0;
        const int y = 2 * k + ((colour + gz) & 1);
        const bool active = L < nlines && y < g.ny;
        const bool bnd = (y == 0) || (y == g.ny - 1) || (DIM == 3 && (gz == 0 || gz == g.gnz - 1));
        lbase[t] = active ? (long long)z * g.plane + (long long)y * g.pitch : 0;
        lflag[t] = active ? (bnd ? 2 : 1) : 0;
    }
    __syncthreads();
    // LDS-only release / acquire around the barrier: a full __syncthreads() also waits for every outstanding global load,
    // i.e. for the prefetch of the next chunk that was just issued
    auto lds_barrier = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    const int nx = g.nx, nch = (nx + XC - 1) / XC;
    const int myflag = lflag[t];
    // this lane's tile points: column col of lines l0 + LPL * q, q = 0 .. XC-1
    const int col = t % XC, l0 = t / XC;
    long long pb[XC];   // element offset of (line, column 0) for the lane's XC lines
    int pf[XC];
#pragma unroll
    for (int q = 0; q < XC; q++) { pb[q] = lbase[l0 + LPL * q]; pf[q] = lflag[l0 + LPL * q]; }

    T b[XC], um[XC], up[XC], zm[XC], zp[XC], fac = 0;
    auto load_chunk = [&](int ch) {   // operands of chunk ch: only loads
        const int x = ch * XC + col;
        const int xc_ = min(x, nx - 1);
        const bool xin = x < nx, xinner = x > 0 && x < nx - 1;
#pragma unroll
        for (int q = 0; q < XC; q++) {
            const long long idx = pb[q] + xc_;
            const bool inner = pf[q] == 1 && xinner;
            b[q] = (pf[q] && xin) ? rhs[idx] : (T)0;
            um[q] = inner ? u[idx - g.pitch] : (T)0;
            up[q] = inner ? u[idx + g.pitch] : (T)0;
            zm[q] = (DIM == 3 && inner) ? u[idx - g.plane] : (T)0;
            zp[q] = (DIM == 3 && inner) ? u[idx + g.plane] : (T)0;
        }
        fac = den[min(ch * XC + col, nx - 1)];
    };
    T carry = 0;  // dp(x-1) of the lane's own line
    load_chunk(0);
    for (int ch = 0; ch < nch; ch++) {
        const int x0 = ch * XC, ncol = min(XC, nx - x0);
        {   // (A) from the registers loaded one chunk ago
            const int x = x0 + col;
            const bool xinner = x > 0 && x < nx - 1;
#pragma unroll
            for (int q = 0; q < XC; q++) {
                T R = b[q];
                if (pf[q] == 1 && xinner) {
                    T S = 0;
                    if (DIM == 3) S += c.cz * zm[q];
                    S += c.cy * um[q];
                    S += c.cy * up[q];
                    if (DIM == 3) S += c.cz * zp[q];
                    R = R - S;
                }
                tile[l0 + LPL * q][col] = R;
            }
            if (t < XC) sfac[t] = fac;   // lanes 0 .. XC-1 have col == t
        }
        lds_barrier();
        if (ch + 1 < nch) load_chunk(ch + 1);   // in flight during the recurrence and the store below
        if (myflag == 1) {  // (B) dp(0) = b(0); dp(x) = (R(x) - cx dp(x-1)) / den(x); the last column keeps b(nx-1)
#pragma unroll
            for (int k = 0; k < XC; k++) {
                const int x = x0 + k;
                const T r = tile[t][k];
                T nc = (r - c.cx * carry) / sfac[k];
                if (x == 0 || x == nx - 1) nc = r;
                if (k < ncol) { carry = nc; tile[t][k] = nc; }
            }
        }
        lds_barrier();
        {   // (C)
            const int x = x0 + col;
#pragma unroll
            for (int q = 0; q < XC; q++)
                if (pf[q] && x < nx) dp[pb[q] + x] = tile[l0 + LPL * q][col];
        }
        lds_barrier();
    }
    // back substitution: u(nx-1) = b(nx-1); u(x) = dp(x) - cp(x) u(x+1); u(0) = dp(0)
    __syncthreads();   // the dp stores of the forward pass are read back by other lanes: global memory ordered here
    T d[XC];
    auto load_back = [&](int ch) {
        const int x = ch * XC + col;
#pragma unroll
        for (int q = 0; q < XC; q++) d[q] = (pf[q] && x < nx) ? dp[pb[q] + x] : (T)0;
        fac = cp[min(x, nx - 1)];
    };
    load_back(nch - 1);
    for (int ch = nch - 1; ch >= 0; ch--) {
        const int x0 = ch * XC, ncol = min(XC, nx - x0);
#pragma unroll
        for (int q = 0; q < XC; q++) tile[l0 + LPL * q][col] = d[q];
        if (t < XC) sfac[t] = fac;
        lds_barrier();
        if (ch > 0) load_back(ch - 1);
        if (myflag == 1) {
#pragma unroll
            for (int k = XC - 1; k >= 0; k--) {
                const int x = x0 + k;
                const T dd = tile[t][k];
                T nc = dd - sfac[k] * carry;
                if (x == 0 || x == nx - 1) nc = dd;
                if (k < ncol) { carry = nc; tile[t][k] = nc; }
            }
        }
        lds_barrier();
        {
            const int x = x0 + col;
#pragma unroll
            for (int q = 0; q < XC; q++)
                if (pf[q] && x < nx) u[pb[q] + x] = tile[l0 + LPL * q][col];
        }
        lds_barrier();
    }
}

inline dim3 grid_for(int nx, int ny, int nz)
{
    return dim3((nx + BX - 1) / BX, (ny + BY - 1) / BY, nz);
}

}  // namespace

int reduce_partials_capacity(const Geom &g)
{
    dim3 gr = grid_for(g.nx, g.ny, g.nz);
    int cap = (int)(gr.x * gr.y * gr.z);
    cap = max(cap, fast_partials_capacity<double>(g));
    cap = max(cap, fast_partials_capacity<float>(g));
    return cap;
}

template <typename T>
void launch_jacobi(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u,
                   const T *rhs, T *out, bool zero_u)
{
    if (fast_path_ok<T>(g)) { launch_jacobi_fast<T>(s, g, c, omega, u, rhs, out, zero_u); return; }
    dim3 gr = grid_for(g.nx, g.ny, g.nz), bl(BX, BY, 1);
    const bool damped = (omega != (T)1);
    if (g.dim == 3) {
        if (damped) hipLaunchKernelGGL((k_jacobi<T, 3, true>), gr, bl, 0, s, g, c, omega, u, rhs, out);
        else hipLaunchKernelGGL((k_jacobi<T, 3, false>), gr, bl, 0, s, g, c, omega, u, rhs, out);
    } else {
        if (damped) hipLaunchKernelGGL((k_jacobi<T, 2, true>), gr, bl, 0, s, g, c, omega, u, rhs, out);
        else hipLaunchKernelGGL((k_jacobi<T, 2, false>), gr, bl, 0, s, g, c, omega, u, rhs, out);
    }
}

template <typename T>
void launch_rbgs_colour(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u,
                        const T *rhs)
{
    dim3 gr = grid_for((g.nx + 1) / 2, g.ny, g.nz), bl(BX, BY, 1);
    if (g.dim == 3) hipLaunchKernelGGL((k_rbgs<T, 3>), gr, bl, 0, s, g, c, colour, u, rhs);
    else hipLaunchKernelGGL((k_rbgs<T, 2>), gr, bl, 0, s, g, c, colour, u, rhs);
}

template <typename T>
void launch_gs_lex(hipStream_t s, const Geom &g, const Coef<T> &c, int sweeps, T *u, const T *rhs)
{
    static const bool rows = [] { const char *e = getenv("MG_GS_ROWS"); return !(e && e[0] == '0'); }();
    if (g.dim == 3) hipLaunchKernelGGL((k_gs_lex<T, 3>), dim3(1), dim3(SWG), 0, s, g, c, sweeps, u, rhs);
    else if (rows && g.ny <= SWG && g.nx >= 3 && g.ny >= 3) {  // one thread per row
        static const int pf = [] { const char *e = getenv("MG_GS_PF"); return e ? atoi(e) : 4; }();
        static const bool pairs = [] { const char *e = getenv("MG_GS_PAIR"); return !(e && e[0] == '0'); }();
        const dim3 bl(((g.ny + 63) / 64) * 64);
        int left = sweeps;
        while (pairs && left >= 2) {  // two sweeps per wavefront pass
            hipLaunchKernelGGL((k_gs_lex2d_rows_pair<T, 4>), dim3(1), bl, 0, s, g, c, u, rhs);
            left -= 2;
        }
        if (left > 0) {
            if (pf == 16) hipLaunchKernelGGL((k_gs_lex2d_rows<T, 16>), dim3(1), bl, 0, s, g, c, left, u, rhs);
            else if (pf == 8) hipLaunchKernelGGL((k_gs_lex2d_rows<T, 8>), dim3(1), bl, 0, s, g, c, left, u, rhs);
            else hipLaunchKernelGGL((k_gs_lex2d_rows<T, 4>), dim3(1), bl, 0, s, g, c, left, u, rhs);
        }
    }
    else hipLaunchKernelGGL((k_gs_lex<T, 2>), dim3(1), dim3(SWG), 0, s, g, c, sweeps, u, rhs);
}

template <typename T>
void launch_zebra_y(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u, const T *rhs, T *dp,
                    const T *cp_den)
{
    dim3 bl(64, 4, 1), gr(((g.nx + 1) / 2 + 63) / 64, (g.nz + 3) / 4, 1);  // a lane per line of the active colour
    if (g.dim == 3) hipLaunchKernelGGL((k_zebra_y<T, 3>), gr, bl, 0, s, g, c, colour, u, rhs, dp, cp_den, cp_den + g.ny);
    else hipLaunchKernelGGL((k_zebra_y<T, 2>), gr, bl, 0, s, g, c, colour, u, rhs, dp, cp_den, cp_den + g.ny);
}

template <typename T>
void launch_zebra_x(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, T *u, const T *rhs, T *dp,
                    const T *cp_den)
{
    const int lpp = (g.ny + 1) / 2, nlines = lpp * g.nz;   // line slots of one colour: (z, k) -> y = 2k + parity
    dim3 bl(ZXT), gr((nlines + ZXT - 1) / ZXT);
    if (g.dim == 3) hipLaunchKernelGGL((k_zebra_x<T, 3>), gr, bl, 0, s, g, c, colour, u, rhs, dp, cp_den, cp_den + g.nx, lpp, nlines);
    else hipLaunchKernelGGL((k_zebra_x<T, 2>), gr, bl, 0, s, g, c, colour, u, rhs, dp, cp_den, cp_den + g.nx, lpp, nlines);
}

// cp(j), den(j) of the line solve (cl = the off-diagonal along the line), computed in T with the same two operations
// per row as the CPU restatement; out: 2 * n values
template <typename T>
void zebra_line_factors(T cl, T cd, int n, T *out)
{
    T *cp = out, *den = out + n;
    cp[0] = 0; den[0] = 1;
    for (int j = 1; j < n - 1; j++) { den[j] = cd - cl * cp[j - 1]; cp[j] = cl / den[j]; }
    cp[n - 1] = 0; den[n - 1] = 1;
}

template <typename T>
void launch_residual(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs,
                     T *r, double *d_partials, double *d_sumsq)
{
    // d_sumsq == nullptr: residual vector only (V-cycle restriction input), no norm
    if (fast_path_ok<T>(g)) {
        int np = launch_residual_fast<T>(s, g, c, u, rhs, r, d_partials, d_sumsq != nullptr);
        if (d_sumsq) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, d_partials, (long long)np, d_sumsq);
        return;
    }
    dim3 gr = grid_for(g.nx, g.ny, g.nz), bl(BX, BY, 1);
    long long nb = (long long)gr.x * gr.y * gr.z;
#define MG_RES(DIM, SAVE, NORM) \
    hipLaunchKernelGGL((k_residual<T, DIM, SAVE, NORM>), gr, bl, 0, s, g, c, u, rhs, r, d_partials)
    if (g.dim == 3) {
        if (d_sumsq) { if (r) MG_RES(3, true, true); else MG_RES(3, false, true); }
        else MG_RES(3, true, false);
    } else {
        if (d_sumsq) { if (r) MG_RES(2, true, true); else MG_RES(2, false, true); }
        else MG_RES(2, true, false);
    }
#undef MG_RES
    if (d_sumsq) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, d_partials, nb, d_sumsq);
}

void launch_reduce_final(hipStream_t s, const double *d_partials, long long n, double *d_out)
{
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, d_partials, n, d_out);
}

template <typename T>
void launch_sumsq(hipStream_t s, const Geom &g, const T *v, double *d_partials, double *d_sumsq)
{
    dim3 gr = grid_for(g.nx, g.ny, g.nz), bl(BX, BY, 1);
    long long nb = (long long)gr.x * gr.y * gr.z;
    hipLaunchKernelGGL((k_sumsq<T>), gr, bl, 0, s, g, v, d_partials);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, d_partials, nb, d_sumsq);
}

static inline bool is_semi(const Geom &gf, const Geom &gc) { return gf.dim == 3 && gf.gnz == gc.gnz && gf.gnz > 1; }

template <typename T>
void launch_inject(hipStream_t s, const Geom &gf, const Geom &gc, const T *fine, T *coarse)
{
    dim3 gr = grid_for(gc.nx, gc.ny, gc.nz), bl(BX, BY, 1);
    hipLaunchKernelGGL((k_inject<T>), gr, bl, 0, s, gf, gc, is_semi(gf, gc) ? 1 : 0, fine, coarse);
}

template <typename T>
void launch_restrict_fw(hipStream_t s, const Geom &gf, const Geom &gc, const T *fine, T *coarse)
{
    dim3 gr = grid_for(gc.nx, gc.ny, gc.nz), bl(BX, BY, 1);
    if (gc.dim == 3 && is_semi(gf, gc)) hipLaunchKernelGGL((k_restrict_fw<T, 3, false>), gr, bl, 0, s, gf, gc, fine, coarse);
    else if (gc.dim == 3) hipLaunchKernelGGL((k_restrict_fw<T, 3, true>), gr, bl, 0, s, gf, gc, fine, coarse);
    else hipLaunchKernelGGL((k_restrict_fw<T, 2, false>), gr, bl, 0, s, gf, gc, fine, coarse);
}

template <typename T>
void launch_prolong(hipStream_t s, const Geom &gc, const Geom &gf, const T *coarse, T *fine,
                    bool add)
{
    if (prolong_fast_ok<T>(gc, gf)) { launch_prolong_fast<T>(s, gc, gf, coarse, fine, add); return; }
    dim3 gr = grid_for(gf.nx, gf.ny, gf.nz), bl(BX, BY, 1);
    if (gf.dim == 3 && is_semi(gf, gc)) {
        if (add) hipLaunchKernelGGL((k_prolong<T, 23, true>), gr, bl, 0, s, gc, gf, coarse, fine);
        else hipLaunchKernelGGL((k_prolong<T, 23, false>), gr, bl, 0, s, gc, gf, coarse, fine);
    } else if (gf.dim == 3) {
        if (add) hipLaunchKernelGGL((k_prolong<T, 3, true>), gr, bl, 0, s, gc, gf, coarse, fine);
        else hipLaunchKernelGGL((k_prolong<T, 3, false>), gr, bl, 0, s, gc, gf, coarse, fine);
    } else {
        if (add) hipLaunchKernelGGL((k_prolong<T, 2, true>), gr, bl, 0, s, gc, gf, coarse, fine);
        else hipLaunchKernelGGL((k_prolong<T, 2, false>), gr, bl, 0, s, gc, gf, coarse, fine);
    }
}

template <typename T>
void launch_correct(hipStream_t s, const Geom &g, T *u, T *e)
{
    dim3 gr = grid_for(g.nx, g.ny, g.nz), bl(BX, BY, 1);
    hipLaunchKernelGGL((k_correct<T>), gr, bl, 0, s, g, u, e);
}

template <typename T, int DIM, int PT>
static bool try_launch_coarse_lds(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, int smoother, T *x,
                                  const T *rhs, int maxit, double tol, int fixed, CoarseOut *d_out)
{
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    const size_t bytes = 3 * total * sizeof(T);
    if (total > (size_t)PT * SWG || bytes > (size_t)150 * 1024) return false;
    auto kern = k_coarse_solve_lds<T, DIM, PT>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) return false;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(SWG), bytes, s, g, c, omega, smoother, x, rhs, maxit, tol, fixed, d_out);
    return true;
}


template <typename T, int DIM, int SEG>
static bool try_launch_coarse_jacobi_rows(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, T *x,
                                          const T *rhs, int maxit, double tol, int fixed, CoarseOut *d_out, int zero_x)
{
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    // MG_COARSE_SKIP = sweeps between two norm tests (1 = test after every sweep, the plain loop)
    static const int skip_env = [] { const char *e = getenv("MG_COARSE_SKIP"); return e ? atoi(e) : 8; }();
    int skip = skip_env < 1 ? 1 : skip_env;
    if (3 * total * sizeof(T) > (size_t)150 * 1024) skip = 1;   // no room for the window's first iterate
    // the windowed norm test equals the reference's test per sweep only while the residual norm cannot dip below the tolerance
    // and rise again inside a window: guaranteed for 0 < omega <= 1 (damped Jacobi on this operator is a contraction in the
    // 2-norm), not outside that range -- there every sweep is tested
    if (!(omega > (T)0 && omega <= (T)1)) skip = 1;
    const size_t bytes = (skip > 1 ? 3 : 2) * total * sizeof(T);
    const int W = g.nx - 2, nseg = (W + SEG - 1) / SEG;
    if (W < SEG || nseg * SEG - W > 1) return false;  // full runs, at most one shared point per row
    const int threads = nseg * (g.ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    if (threads < 128 || threads > CoarseRowsThreads<SEG>::value || bytes > (size_t)150 * 1024) return false;
    auto kern = k_coarse_jacobi_rows<T, DIM, SEG>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) return false;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(((threads + 63) / 64) * 64), bytes, s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, skip, zero_x);
    return true;
}

template <typename T, int DIM>
static bool try_launch_coarse_jacobi(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, T *x, const T *rhs,
                                     int maxit, double tol, int fixed, CoarseOut *d_out, int zero_x)
{
    static const bool enabled = [] { const char *e = getenv("MG_COARSE_ROWS"); return !(e && e[0] == '0'); }();
    if (!enabled || g.ny < 3 || (DIM == 3 && g.nz < 3)) return false;
    // run length, in measured order of preference (MI355X, one CU): 65^2 -> 8 (504 threads, two
    // waves per SIMD: 52 ms for BASELINE config 1 against 55 ms with 7 and 64 ms with 9);
    // 17^3 -> 5 (675 threads; 8 is 2 % slower per V-cycle at 513^3)
    static const int pref = [] { const char *e = getenv("MG_COARSE_SEG"); return e ? atoi(e) : 0; }();
    const int W = g.nx - 2, irows = (g.ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    const int order2[5] = {pref, 8, 4, 7, 5}, order3[5] = {pref, 5, 4, 8, 7};
    int best = 0;
    for (int seg : (DIM == 3 ? order3 : order2)) {
        if (seg != 4 && seg != 5 && seg != 7 && seg != 8) continue;
        const int nseg = (W + seg - 1) / seg, threads = nseg * irows;
        if (W < seg || nseg * seg - W > 1 || threads < 128 || threads > (seg >= 7 ? 512 : SWG)) continue;
        best = seg;
        break;
    }
    switch (best) {
    case 4: return try_launch_coarse_jacobi_rows<T, DIM, 4>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, zero_x);
    case 5: return try_launch_coarse_jacobi_rows<T, DIM, 5>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, zero_x);
    case 7: return try_launch_coarse_jacobi_rows<T, DIM, 7>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, zero_x);
    case 8: return try_launch_coarse_jacobi_rows<T, DIM, 8>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, zero_x);
    default: return false;  // other widths keep the generic LDS kernel
    }
}

template <typename T, int DIM, int SEG>
static bool try_launch_coarse_rb_rows(hipStream_t s, const Geom &g, const Coef<T> &c, T *x, const T *rhs, int maxit, double tol,
                                      int fixed, CoarseOut *d_out, int zero_x)
{
    const size_t total = (size_t)g.nx * g.ny * g.nz, bytes = 2 * total * sizeof(T);   // two copies of the iterate
    const int W = g.nx - 2, nseg = (W + SEG - 1) / SEG;
    if (W < SEG || nseg * SEG - W > 1) return false;  // full runs, at most one shared point per row
    const int threads = nseg * (g.ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    if (threads < 128 || threads > CoarseRowsThreads<SEG>::value || bytes > (size_t)150 * 1024) return false;
    auto kern = k_coarse_rb_rows<T, DIM, SEG>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) return false;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(((threads + 63) / 64) * 64), bytes, s, g, c, x, rhs, maxit, tol, fixed, d_out, zero_x);
    return true;
}

// red-black coarse solve on the row-segment layout (MG_COARSE_RB_ROWS=0: the generic LDS kernel)
template <typename T, int DIM>
static bool try_launch_coarse_rb(hipStream_t s, const Geom &g, const Coef<T> &c, T *x, const T *rhs, int maxit, double tol,
                                 int fixed, CoarseOut *d_out, int zero_x)
{
    static const bool enabled = [] { const char *e = getenv("MG_COARSE_RB_ROWS"); return !(e && e[0] == '0'); }();
    if (!enabled || g.ny < 3 || (DIM == 3 && g.nz < 3)) return false;
    const int W = g.nx - 2, irows = (g.ny - 2) * (DIM == 3 ? g.nz - 2 : 1);
    for (int seg : {5, 4}) {   // (runs of 7 / 8 points: the colour selects went through scratch memory; not instantiated)
        const int nseg = (W + seg - 1) / seg, threads = nseg * irows;
        if (W < seg || nseg * seg - W > 1 || threads < 128 || threads > SWG) continue;
        if (seg == 5) return try_launch_coarse_rb_rows<T, DIM, 5>(s, g, c, x, rhs, maxit, tol, fixed, d_out, zero_x);
        return try_launch_coarse_rb_rows<T, DIM, 4>(s, g, c, x, rhs, maxit, tol, fixed, d_out, zero_x);
    }
    return false;
}

template <typename T>
static bool try_launch_coarse_gs_rows2d(hipStream_t s, const Geom &g, const Coef<T> &c, T *x, const T *rhs, int maxit,
                                        double tol, int fixed, CoarseOut *d_out)
{
    static const bool enabled = [] { const char *e = getenv("MG_COARSE_GS_ROWS"); return !(e && e[0] == '0'); }();
    const size_t bytes = 2 * (size_t)g.nx * g.ny * sizeof(T);
    if (!enabled || g.dim != 2 || g.ny > CGS_THREADS || g.nx < 3 || g.ny < 3 || bytes > (size_t)150 * 1024) return false;
    auto kern = k_coarse_gs_rows2d<T>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) return false;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(CGS_THREADS), bytes, s, g, c, x, rhs, maxit, tol, fixed, d_out);
    return true;
}

template <typename T>
void launch_coarse_solve(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, int smoother,
                         T *x, T *tmp, const T *rhs, int maxit, double tol, int fixed,
                         CoarseOut *d_out, bool x_is_zero)
{
    // x_is_zero: the solve starts from the zero guess and the caller has NOT cleared x: the row-segment Jacobi and red-black kernels take the
    // guess as a flag (a memset launch less per cycle); every other kernel gets its x cleared here
    if (x_is_zero && smoother == 1 && g.gz0 == 0 && g.gnz == g.nz) {
        if (g.dim == 3 ? try_launch_coarse_jacobi<T, 3>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, 1)
                       : try_launch_coarse_jacobi<T, 2>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, 1)) return;
    }
    if (x_is_zero && smoother == 2 && g.gz0 == 0 && g.gnz == g.nz) {  // so does the row-segment red-black kernel
        if (g.dim == 3 ? try_launch_coarse_rb<T, 3>(s, g, c, x, rhs, maxit, tol, fixed, d_out, 1)
                       : try_launch_coarse_rb<T, 2>(s, g, c, x, rhs, maxit, tol, fixed, d_out, 1)) return;
    }
    if (x_is_zero) (void)hipMemsetAsync(x, 0, (size_t)g.nz * (size_t)g.plane * sizeof(T), s);
    // LDS-resident when the three arrays fit one CU's LDS, global-memory loop otherwise
    if (smoother == 1 && g.gz0 == 0 && g.gnz == g.nz) {  // Jacobi: the row-segment kernel
        if (g.dim == 3 ? try_launch_coarse_jacobi<T, 3>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, 0)
                       : try_launch_coarse_jacobi<T, 2>(s, g, c, omega, x, rhs, maxit, tol, fixed, d_out, 0)) return;
    }
    if (smoother == 0 && try_launch_coarse_gs_rows2d<T>(s, g, c, x, rhs, maxit, tol, fixed, d_out)) return;
    if (smoother == 2 && g.gz0 == 0 && g.gnz == g.nz) {  // red-black: the row-segment kernel
        if (g.dim == 3 ? try_launch_coarse_rb<T, 3>(s, g, c, x, rhs, maxit, tol, fixed, d_out, 0)
                       : try_launch_coarse_rb<T, 2>(s, g, c, x, rhs, maxit, tol, fixed, d_out, 0)) return;
    }
    if (g.dim == 3) {
        if (try_launch_coarse_lds<T, 3, 5>(s, g, c, omega, smoother, x, rhs, maxit, tol, fixed, d_out)) return;
        hipLaunchKernelGGL((k_coarse_solve<T, 3>), dim3(1), dim3(SWG), 0, s, g, c, omega, smoother, x,
                           tmp, rhs, maxit, tol, fixed, d_out);
    } else {
        if (try_launch_coarse_lds<T, 2, 5>(s, g, c, omega, smoother, x, rhs, maxit, tol, fixed, d_out)) return;
        hipLaunchKernelGGL((k_coarse_solve<T, 2>), dim3(1), dim3(SWG), 0, s, g, c, omega, smoother, x,
                           tmp, rhs, maxit, tol, fixed, d_out);
    }
}

#define MG_INSTANTIATE(T)                                                                          \
    template void launch_jacobi<T>(hipStream_t, const Geom &, const Coef<T> &, T, const T *,       \
                                   const T *, T *, bool);                                              \
    template void launch_rbgs_colour<T>(hipStream_t, const Geom &, const Coef<T> &, int, T *,      \
                                        const T *);                                                \
    template void launch_gs_lex<T>(hipStream_t, const Geom &, const Coef<T> &, int, T *, const T *); \
    template void launch_zebra_y<T>(hipStream_t, const Geom &, const Coef<T> &, int, T *, const T *, T *, const T *); \
    template void launch_zebra_x<T>(hipStream_t, const Geom &, const Coef<T> &, int, T *, const T *, T *, const T *); \
    template void zebra_line_factors<T>(T, T, int, T *); \
    template void launch_residual<T>(hipStream_t, const Geom &, const Coef<T> &, const T *,        \
                                     const T *, T *, double *, double *);                          \
    template void launch_sumsq<T>(hipStream_t, const Geom &, const T *, double *, double *);       \
    template void launch_inject<T>(hipStream_t, const Geom &, const Geom &, const T *, T *);       \
    template void launch_restrict_fw<T>(hipStream_t, const Geom &, const Geom &, const T *, T *);  \
    template void launch_prolong<T>(hipStream_t, const Geom &, const Geom &, const T *, T *, bool); \
    template void launch_correct<T>(hipStream_t, const Geom &, T *, T *);                          \
    template void launch_coarse_solve<T>(hipStream_t, const Geom &, const Coef<T> &, T, int, T *,  \
                                         T *, const T *, int, double, int, CoarseOut *, bool);
MG_INSTANTIATE(double)
MG_INSTANTIATE(float)

}  // namespace mg
