// mg_jacobi_fast.hip -- finest-grid fast path for the 7-point sweeps (Jacobi update and
// residual), gfx950.  Design chosen with tools/kbench.hip on MI355X:
//
//  * one lane owns one aligned 16-byte vector of x (2 doubles / 4 floats); a wave64 covers
//    128 / 256 consecutive x of RY = 2 rows; BW = 4 waves are stacked in y;
//  * the workgroup marches ZC = 3 planes in z with the (z-1, z, z+1) values of its columns
//    in registers: every u value is loaded once per workgroup column, the rest of the
//    7-point neighbourhood comes from registers (z, in-wave y), DPP whole-wave shifts (x)
//    and two y-halo rows that hit L1/L2; short ZC keeps concurrently running workgroups
//    on a few adjacent planes (DRAM-page friendly), the z-halo planes come from L2/MALL;
//  * XCD-aware block order: blockIdx -> (blockIdx % 8) * chunk + blockIdx / 8, so each
//    XCD (own L2) sweeps a contiguous range of rows/planes and y/z-halo re-reads stay in
//    that L2;
//  * non-temporal stores (and non-temporal rhs loads on levels larger than the caches):
//    the output and rhs streams have no reuse inside a sweep;
//  * the odd last column (n = 2^k+1) is written as ONE full 128-byte line (boundary value
//    + the row's zero padding): an 8-byte partial-line store per row costs ~5 % of the
//    sweep on HBM3E;
//  * arithmetic order, -ffp-contract=off and the correctly rounded division (mg_geom.h: div_cd_n)
//    are those of mg_kernels.hip: results are bit-identical to the generic kernels and to the
//    CPU restatement the tests compare with.
// MFMA is not used: there is no contraction here, the kernel is HBM-bound (24 B/point).
#include "mg_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace mg {
namespace {

template <typename T> struct VecOf;
template <> struct VecOf<double> { static constexpr int V = 2; typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float> { static constexpr int V = 4; typedef float type __attribute__((ext_vector_type(4))); };

// lane i <- lane i-1 (lane 0 keeps `edge`) : DPP wave_shr:1
__device__ __forceinline__ float from_prev_lane(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double from_prev_lane(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i+1 (lane 63 keeps `edge`) : DPP wave_shl:1
__device__ __forceinline__ float from_next_lane(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double from_next_lane(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

enum { OP_JACOBI = 0, OP_RESIDUAL = 1, OP_RB = 2 };

constexpr int RY = 2, BW = 4, ZC = 3;  // ZC: planes marched per workgroup on a level big enough to fill the chip

// OP_JACOBI : out = Jacobi update (DAMPED selects omega != 1)
// OP_RESIDUAL: out = rhs - A u (stored if SAVE), sum r^2 -> partials[block] if NORM
// OP_RB     : one colour half-sweep of red-black Gauss-Seidel, out of place: points with
//             (x+y+gz)&1 == colour get the Gauss-Seidel update, the others are copied, so
//             red: u -> tmp, black: tmp -> u leaves the sweep's result in u
// ZEROU     : u is identically zero (first pre-smoothing sweep of a coarse level): nothing
//             is loaded for it, which also saves the memset of the initial guess
template <typename T, int OP, bool DAMPED, bool SAVE, bool NORM, bool NTLOAD, bool ZEROU = false>
__global__ __launch_bounds__(64 * BW) void k_sweep3d(Geom g, Coef<T> c, T omega, int colour,
                                                     const T *__restrict__ u,
                                                     const T *__restrict__ rhs, T *__restrict__ out,
                                                     double *__restrict__ partials, int nbx, int nby,
                                                     int nbz)
{
    constexpr int V = VecOf<T>::V;
    typedef typename VecOf<T>::type vec;
    __shared__ double sh[BW];

    const int nblocks = nbx * nby * nbz;
    const int per = (nblocks + 7) >> 3;
    const int bid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // XCD-aware order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double sq = 0.;
    if (bid < nblocks) {
        const int bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
        const int nvec = g.nx / V;
        const int x0 = V * (bx * 64 + lane);
        const bool xin = x0 < V * nvec;
        const int x0c = min(x0, g.pitch - V);  // clamped for loads: every lane stays active
        const int yb = (by * BW + wv) * RY;
        const int zc = (g.nz + nbz - 1) / nbz;   // planes marched per workgroup (the launcher picks nbz)
        const int z0 = bz * zc;
        const int zend = min(z0 + zc, g.nz);
        // does this wave hold the last full vector of the row? then it also writes column nx-1
        const bool tailwave = (g.nx % V == 1) && (bx * 64 * V <= g.nx - 1 - V) && (g.nx - 1 - V < (bx + 1) * 64 * V);

        long long rowoff[RY];
        bool yin[RY], ybnd[RY];
#pragma unroll
        for (int r = 0; r < RY; r++) {
            int y = yb + r;
            yin[r] = y < g.ny;
            int yc = min(y, g.ny - 1);
            ybnd[r] = (yc == 0) || (yc == g.ny - 1);
            rowoff[r] = (long long)yc * g.pitch + x0c;
        }
        const long long off_lo = (long long)max(yb - 1, 0) * g.pitch + x0c;
        const long long off_hi = (long long)min(yb + RY, g.ny - 1) * g.pitch + x0c;

        vec zm[RY], cc[RY], zp[RY];
        const T *pz = u + (long long)z0 * g.plane;
#pragma unroll
        for (int r = 0; r < RY; r++) {
            zm[r] = ZEROU ? (vec)(0) : *(const vec *)(pz - g.plane + rowoff[r]);
            cc[r] = ZEROU ? (vec)(0) : *(const vec *)(pz + rowoff[r]);
        }
        for (int z = z0; z < zend; z++, pz += g.plane) {
            const long long zo = (long long)z * g.plane;
            vec b[RY];
#pragma unroll
            for (int r = 0; r < RY; r++) {
                zp[r] = ZEROU ? (vec)(0) : *(const vec *)(pz + g.plane + rowoff[r]);
                if (NTLOAD) b[r] = __builtin_nontemporal_load((const vec *)(rhs + zo + rowoff[r]));
                else b[r] = *(const vec *)(rhs + zo + rowoff[r]);
            }
            const vec hlo = ZEROU ? (vec)(0) : *(const vec *)(pz + off_lo);
            const vec hhi = ZEROU ? (vec)(0) : *(const vec *)(pz + off_hi);
            const int gz = g.gz0 + z;
            const bool zb = (gz == 0) || (gz == g.gnz - 1);
#pragma unroll
            for (int r = 0; r < RY; r++) {
                T el = 0, er = 0;
                if (!ZEROU && lane == 0) el = pz[rowoff[r] - 1];
                if (!ZEROU && lane == 63) er = pz[rowoff[r] + V];
                const T xm = from_prev_lane(cc[r][V - 1], el);
                const T xp = from_next_lane(cc[r][0], er);
                const vec ym = (r > 0) ? cc[r > 0 ? r - 1 : 0] : hlo;
                const vec yp = (r < RY - 1) ? cc[r < RY - 1 ? r + 1 : 0] : hhi;
                const bool rb = zb || ybnd[r];
                vec res;
                T num[V], quo[V];
#pragma unroll
                for (int e = 0; e < V; e++) {
                    const T left = (e == 0) ? xm : cc[r][e > 0 ? e - 1 : 0];
                    const T right = (e == V - 1) ? xp : cc[r][e < V - 1 ? e + 1 : 0];
                    const bool bnd = rb || (x0 + e == 0) || (x0 + e == g.nx - 1);
                    T sum = 0;
                    sum += c.cz * zm[r][e];
                    sum += c.cy * ym[e];
                    sum += c.cx * left;
                    if (OP == OP_RESIDUAL) sum += c.cd * cc[r][e];
                    sum += c.cx * right;
                    sum += c.cy * yp[e];
                    sum += c.cz * zp[r][e];
                    if (OP == OP_RESIDUAL) {
                        if (bnd) sum = (T)1 * cc[r][e];
                        res[e] = b[r][e] - sum;
                    }
                    num[e] = b[r][e] - sum;
                }
                if (OP != OP_RESIDUAL) {
                    div_cd_n<T, V>(num, quo, c);
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        const bool bnd = rb || (x0 + e == 0) || (x0 + e == g.nx - 1);
                        if (OP == OP_JACOBI) {
                            T jac = quo[e];
                            if (DAMPED) jac = cc[r][e] + omega * (jac - cc[r][e]);
                            res[e] = bnd ? b[r][e] : jac;
                        } else {
                            const bool mine = ((x0 + e + yb + r + gz) & 1) == colour;
                            res[e] = mine ? (bnd ? b[r][e] : quo[e]) : cc[r][e];
                        }
                    }
                }
                if (xin && yin[r]) {
                    if (OP != OP_RESIDUAL || SAVE) __builtin_nontemporal_store(res, (vec *)(out + zo + rowoff[r]));
                    if (NORM) {
#pragma unroll
                        for (int e = 0; e < V; e++) sq += (double)res[e] * (double)res[e];
                    }
                }
                if (tailwave && lane >= 56 && yin[r]) {
                    // column nx-1 (Dirichlet) as one full 128-byte line: value + zero padding
                    const int j = lane - 56;
                    constexpr int LINE = 128 / (int)sizeof(T);
                    const int xs = g.nx - 1 + V * j;
                    const int line_end = ((g.nx - 1) / LINE + 1) * LINE;
                    const long long ro = zo + (rowoff[r] - x0c);
                    T tb = 0, tres = 0;
                    if (j == 0) {
                        tb = rhs[ro + g.nx - 1];
                        if (OP == OP_JACOBI) tres = tb;
                        else if (OP == OP_RB) tres = (((g.nx - 1 + yb + r + gz) & 1) == colour) ? tb : pz[(rowoff[r] - x0c) + g.nx - 1];
                        else tres = tb - (T)1 * (ZEROU ? (T)0 : pz[(rowoff[r] - x0c) + g.nx - 1]);
                        if (NORM) sq += (double)tres * (double)tres;
                    }
                    if (xs < line_end && (OP != OP_RESIDUAL || SAVE)) {
                        vec tv = (vec)(0);
                        tv[0] = tres;
                        __builtin_nontemporal_store(tv, (vec *)(out + ro + xs));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RY; r++) { zm[r] = cc[r]; cc[r] = zp[r]; }
        }
    }
    if (NORM) {
        sq = wave_sum64(sq);
        if (lane == 0) sh[wv] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0;
            for (int w = 0; w < BW; w++) s += sh[w];
            partials[blockIdx.x] = s;
        }
    }
}


// ------------------------------------------------------------------------------------------
// Fused double Jacobi sweep: out = J(J(u)) in ONE pass over HBM (temporal blocking, V(2,2)'s
// sweep pairs). Workgroup = TPR threads = one full grid row (TPR vectors + the odd tail column)
// x TYO output rows, marching ZC planes. Per plane p: (1) the first sweep v(p) on TYO+2 rows
// (overlapped tiling in y) from u planes p-1, p, p+1 held in registers; (2) the second sweep on
// plane q = p-1 of the TYO output rows: its z-neighbours v(q-1), v(q+1) are the thread's own
// registers, its x/y-neighbours come from the LDS copy of v(q) published one step earlier;
// (3) v(p) goes to the other LDS slot; one barrier per plane. The x-neighbours across wave
// boundaries travel through a small LDS mailbox one plane ahead (uedge / utail). Same per-point
// arithmetic as two k_sweep3d launches => bit-identical; 0.85 ms against 2 x 0.60 ms per pair at
// 513^3 fp64 (3 workgroups per CU; DESIGN.md section 4 has the history).
constexpr int J2_TYO = 2;
// z-chunks per launch. Long marches amortise the two extra planes of first-sweep work per chunk,
// short ones give the smaller levels enough workgroups to fill 256 CUs x 3. Measured per V-cycle:
// 513^3: 24 planes (16: +1.2 %, 32: +2 %, 8: +5 %); 257^3: 8 (16: +6 %, 4: +3 %); 129^3: 4 (8: +3 %);
// the rule below gives 24 / 8 / 3 there.
static int j2_nbz(const Geom &g, int tpr, int tyo = 3)   // tyo: output rows per workgroup of the launch (slab rule only)
{
    static const int zc_env = [] { const char *e = getenv("MG_J2_ZC"); return e ? atoi(e) : 0; }();
    static const bool slab_rule = [] { const char *e = getenv("MG_J2_SLAB_RULE"); return !(e && e[0] == '0'); }();
    int zc = 3;  // the longest march that still leaves enough workgroups (semi-coarsened levels: long in z, few rows)
    if (zc_env > 1) zc = zc_env;
    else if (slab_rule && g.gnz != g.nz && g.nz >= 8) {
        // A piece of a z-slab (the interior planes of a rank's share of a distributed level): a few dozen to a few hundred planes,
        // where the rule below would cut 60 planes into 20 marches of 3 (5 first-sweep planes per 3 outputs). Measured on one
        // rank's schedule (tools/dry_zc.sh, 513^2 planes): the best chunk count is the one whose workgroups fill a whole
        // number of rounds of the chip's resident workgroups from just below -- 60 planes: 4 marches of 15 (142 us per pair
        // against 167), 124 planes: 4 of 31 (234 against 267), 253 planes: 13 of 20 (440 against 451).
        const int nby = (g.ny + tyo - 1) / tyo;
        const int slots = 256 * (tpr >= 512 ? 1 : 12 / (tpr / 64));   // resident workgroups: 12 waves per CU at 167 VGPRs
        int bestk = 1; double best = 1e30;
        for (int k = 1; k <= g.nz / 4; k++) {
            const int z = (g.nz + k - 1) / k;
            if (z > 32 && k < g.nz / 4) continue;
            const double r = (double)nby * k / slots;
            // a round that overflows by ONE workgroup costs a whole march (9 marches of 29 at 253 planes = 1539 workgroups on
            // 1536 slots: 431 us against 395): anything within 3 % of the next round counts as that round
            const double cost = ceil(r + 0.03) * (z + 4);
            if (cost < best - 1e-9 || (cost < best + 1e-9 && z < (g.nz + bestk - 1) / bestk)) { best = cost; bestk = k; }
        }
        return bestk;
    } else {
        const int nby = (g.ny + 2) / 3;
        for (int cand : {24, 16, 12, 8, 6, 4})
            if (nby * ((g.nz + cand - 1) / cand) >= 2800) { zc = cand; break; }
    }
    return (g.nz + zc - 1) / zc;
}

// CORR: every u value read is u + P e_coarse computed on the fly (the V-cycle's prolong-add folded
// into the post-smoothing pair: the corrected fine array is never written). P e is built with the
// prolongation's own expression tree -- z midpoints, then y, then x, each 0.5*(a+b) -- from the
// coarse values of 4 coarse rows per plane (own and right column in one 16-byte load, L1/L2-hot), so
// u + P e has exactly the bits k_prolong3d_fast<ADD> would have stored.
// RB: the same pipeline runs ONE red-black Gauss-Seidel sweep instead of two Jacobi sweeps: phase 1
// is the red half-sweep (red points updated from u, black points copied), phase 2 the black
// half-sweep on the plane behind (black points updated from phase 1's values, red points
// copied): out = RB(u) in one pass over HBM instead of two (k_sweep3d<OP_RB> twice).
// ZEROU: u is identically zero (the two pre-smoothing sweeps of a coarse level): nothing is loaded for it.
// MINW: minimum waves per SIMD the register allocation is held to (3 = 168 VGPRs; 2 = 256: the fp32 folding variant spills otherwise)
// VW: elements per lane (0 = the type's 16-byte vector). The fp32 folding variant runs with TWO floats per lane: with four it
// needs 180 VGPRs (two waves per SIMD), with two the registers of a row halve and it keeps the occupancy of the fp64 kernel.
template <typename T, int TPR, bool DAMPED, bool NTLOAD, bool CORR = false, bool RB = false, bool ZEROU = false, int TYO_ = J2_TYO,
          int MINW = 3, int VW = 0>
__global__ __launch_bounds__(TPR, MINW) void k_jacobi2(Geom g, Coef<T> c, T omega, const T *__restrict__ u_,
                                                 const T *__restrict__ rhs_, T *__restrict__ out_, int nby, int nbz,
                                                 const T *__restrict__ coarse, Geom gc, int dup_planes)
{
    constexpr int V = VW ? VW : VecOf<T>::V, TYO = TYO_, TYV = TYO + 2;
    constexpr int CV = V / 2;  // coarse columns owned by this thread
    static_assert(!CORR || TYO == 2, "the correction assumes two output rows (y0 even)");
    constexpr int LP = TPR * V + 2 * V;  // LDS row: V pad | TPR*V values | tail column | pad
    typedef T vec __attribute__((ext_vector_type(V)));
    __shared__ __align__(16) T lds[2][TYV][LP];
    // u values on the wave edges (first / last element of every wave's row segment), published one
    // plane ahead: the x-neighbour a wave's edge lane needs belongs to the neighbouring wave of the
    // same workgroup. Loading it from global memory instead cost 2-3 serialised memory round trips
    // per plane step (the compiler waited after each predicated scalar load): 1.01 -> 0.85 ms per launch.
    constexpr int NWV = TPR / 64;
    __shared__ T uedge[2][TYV][NWV][2];
    __shared__ T utail[2][TYV];  // u(nx-1): the right neighbour of the row's last vector
    // dup_planes > 0: ONE launch runs the same geometry twice, the second copy dup_planes planes further up (the two
    // boundary pieces of a z-slab, mg_solver.cpp: pair_on_slab2_t): the workgroups of the second half shift their
    // pointers and the global plane index of local plane 0 -- everything below is unchanged (all of it scalar)
    const int nblocks = nby * nbz, ntotal = dup_planes > 0 ? 2 * nblocks : nblocks;
    const int per = (ntotal + 7) >> 3;
    int bid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);         // XCD-aware order
    if (bid >= ntotal) return;                                    // whole workgroup
    const bool second = bid >= nblocks;
    if (second) { bid -= nblocks; g.gz0 += dup_planes; }
    const long long dup_off = second ? (long long)dup_planes * g.plane : 0;
    const T *__restrict__ u = u_ + dup_off;
    const T *__restrict__ rhs = rhs_ + dup_off;
    T *__restrict__ out = out_ + dup_off;
    const int by = bid % nby, bz = bid / nby;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);    // wave index: scalar
    const int wl = max(wv - 1, 0), wr = min(wv + 1, NWV - 1);  // neighbouring waves (clamped: edge waves ignore the value)
    const int x0 = V * t;                       // the gate guarantees nx - 1 == TPR * V
    const bool tail = (x0 + V == g.nx - 1);     // last thread: also owns the Dirichlet column nx-1
    const bool tailwave = (t >> 6) == (TPR >> 6) - 1;
    const int y0 = by * TYO;                    // output rows y0 .. y0+TYO-1; v rows y0-1 .. y0+TYO
    const int ZC = (g.nz + nbz - 1) / nbz;       // planes marched per workgroup (the launcher picks nbz)
    const int z0 = bz * ZC, z1 = min(z0 + ZC, g.nz);
    // row offsets are workgroup-uniform (SGPRs); the lane's x0 is added last so that the loads can
    // use the scalar-base + 32-bit-offset addressing form instead of a 64-bit VGPR pair per address
    long long urow[TYV];
    bool ybnd[TYV];
#pragma unroll
    for (int r = 0; r < TYV; r++) {
        const int y = min(max(y0 - 1 + r, 0), g.ny - 1);
        ybnd[r] = (y == 0) || (y == g.ny - 1);
        urow[r] = (long long)y * g.pitch;
    }
    const long long urow_lo = (long long)min(max(y0 - 2, 0), g.ny - 1) * g.pitch;
    const long long urow_hi = (long long)min(y0 + TYO + 1, g.ny - 1) * g.pitch;
    // zhalo = planes that exist below local plane 0 / above plane nz-1: 1 for a whole level (the ghost
    // planes), 2 when g describes the inner planes 1 .. nz-2 of a z-slab (pair_on_slab_t: the only
    // launches whose g is not the whole grid in z)
    const int zhalo = (g.gnz != g.nz) ? 2 : 1;
    // global z of local plane 0 and global plane count
    const int gzo = g.gz0, gzn = g.gnz;
    auto plane_of = [&](int p) { return (long long)min(max(p, -zhalo), g.nz - 1 + zhalo) * g.plane; };

    // ---- on-the-fly prolongation (CORR) ------------------------------------------------------
    // Per coarse row the thread loads its own CV columns AND the column to their right with one
    // 16-byte load at 8-byte alignment (neighbouring lanes overlap by one column): no exchange between
    // lanes or waves is needed. A correction "row" is NR values: [0..CV) own columns, [CV] right column.
    // (The row's last thread reads up to gc.nx - 1 in fp64, one element of row padding beyond it in fp32.)
    constexpr int NR = CV + 1;
    const T hf = (T)0.5;
    // (NR values rounded up to a power of two, aligned like the lane's first coarse column)
    typedef T cwide __attribute__((ext_vector_type(NR <= 2 ? 2 : 4), aligned(CV * sizeof(T))));
    int ucrow[4];          // offsets of coarse rows yc0-1 .. yc0+2 (clamped into the grid): workgroup-uniform
    const int ic0 = CV * t;   // own first coarse column
    if (CORR) {
        const int yc0 = y0 >> 1;
#pragma unroll
        for (int j = 0; j < 4; j++) ucrow[j] = min(max(yc0 - 1 + j, 0), gc.ny - 1) * gc.pitch;
    }
    auto load_crow = [&](const T *base, int j, T (&d)[NR]) {
        const cwide w = *(const cwide *)((base + ucrow[j]) + ic0);
#pragma unroll
        for (int m = 0; m < NR; m++) d[m] = w[m];
    };
    // Loads and arithmetic are kept apart so that a plane step issues ALL its loads (u, rhs,
    // coarse) before the first wait: raw_a/raw_b only load, zfin only computes.
    // coarse plane under fine plane P (+1: the upper one of an odd plane), by GLOBAL plane index: `coarse` is local plane 0 of
    // the coarse slab (gc.gz0 = its global index; a piece of a fine slab keeps the whole coarse slab's pointer). Planes outside the
    // arrays' ghost planes are clamped (their values are not used: `in` below)
    auto cplane = [&](int P, int up1) {
        const int kc = ((gzo + P + up1) >> 1) - gc.gz0;
        return coarse + (long long)min(max(kc, -zhalo), gc.nz - 1 + zhalo) * gc.plane;
    };
    auto raw_a = [&](int P, T (&R)[4][NR]) {
        const T *c0 = cplane(P, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) load_crow(c0, j, R[j]);
    };
    auto raw_b = [&](int P, T (&R)[4][NR]) {  // the upper coarse plane of an odd fine plane
        if ((gzo + P) & 1) {
            const T *c1 = cplane(P, 1);
#pragma unroll
            for (int j = 0; j < 4; j++) load_crow(c1, j, R[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < NR; m++) R[j][m] = 0;
        }
    };
    // Z[j][.]: coarse correction interpolated in z onto fine plane P (zero outside the grid)
    auto zfin = [&](int P, const T (&Ra)[4][NR], const T (&Rb)[4][NR], T (&Z)[4][NR]) {
        const int gP = gzo + P;
        const bool in = (gP >= 0) && (gP < gzn), odd = (gP & 1) != 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int m = 0; m < NR; m++) {
                const T z = odd ? hf * (Ra[j][m] + Rb[j][m]) : Ra[j][m];
                Z[j][m] = in ? z : (T)0;
            }
        }
    };
    auto zrows = [&](int P, T (&Z)[4][NR]) {
        T Ra[4][NR], Rb[4][NR];
        raw_a(P, Ra); raw_b(P, Rb); zfin(P, Ra, Rb, Z);
    };
    // y-interpolated rows from Z: index 0 = halo row y0-2, 1..4 = v rows y0-1 .. y0+2, 5 = halo row y0+3
    auto yrows = [&](const T (&Z)[4][NR], T (&Y)[6][NR]) {
#pragma unroll
        for (int m = 0; m < NR; m++) {
            Y[0][m] = Z[0][m];
            Y[1][m] = hf * (Z[0][m] + Z[1][m]);
            Y[2][m] = Z[1][m];
            Y[3][m] = hf * (Z[1][m] + Z[2][m]);
            Y[4][m] = Z[2][m];
            Y[5][m] = hf * (Z[2][m] + Z[3][m]);
        }
    };
    // P e on the thread's own vector of a row / on its left and right wave-edge neighbours
    auto pe_vec = [&](const T (&Yr)[NR]) {
        vec w;
#pragma unroll
        for (int mm = 0; mm < CV; mm++) {
            w[2 * mm] = Yr[mm];
            w[2 * mm + 1] = hf * (Yr[mm] + Yr[mm + 1]);
        }
        return w;
    };
    auto pe_right = [&](const T (&Yr)[NR]) { return Yr[CV]; };                     // x0+V (even)
    auto add_vec = [&](vec a, vec w) {
        vec o;
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = a[e] + w[e];
        return o;
    };

    vec um[TYV], uc[TYV], up[TYV];
    vec vm[TYO], vc[TYO], vp[TYO];  // own-column v(q-1), v(q), v(q+1) of the output rows
    vec bq[TYO];                    // rhs of the output rows on plane q
    // CORR: halo rows y0-2 / y0+TYO+1 of the current plane p, fetched (and corrected) one step ahead
    vec hlo = (vec)(0), hhi = (vec)(0);
    T ter[TYV];                     // tail thread, transient: u(nx-1) on its way to utail
    auto publish_edges = [&](int slot, const vec (&w)[TYV], const T (&tl)[TYV]) {
#pragma unroll
        for (int r = 0; r < TYV; r++) {
            if (lane == 0) uedge[slot][r][wv][0] = w[r][0];
            if (lane == 63) uedge[slot][r][wv][1] = w[r][V - 1];
            if (tail) utail[slot][r] = tl[r];
        }
    };
#pragma unroll
    for (int r = 0; r < TYV; r++) {
        um[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (plane_of(z0 - 2) + urow[r])) + x0);
        uc[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (plane_of(z0 - 1) + urow[r])) + x0);
        ter[r] = 0;
        if (tail && !ZEROU) ter[r] = (u + (plane_of(z0 - 1) + urow[r]))[x0 + V];
    }
    if (CORR) {
        hlo = *(const vec *)((u + (plane_of(z0 - 1) + urow_lo)) + x0);
        hhi = *(const vec *)((u + (plane_of(z0 - 1) + urow_hi)) + x0);
    }
    if (CORR) {
        T Z[4][NR], Y[6][NR];
        zrows(z0 - 2, Z); yrows(Z, Y);
#pragma unroll
        for (int r = 0; r < TYV; r++) um[r] = add_vec(um[r], pe_vec(Y[1 + r]));
        zrows(z0 - 1, Z); yrows(Z, Y);
#pragma unroll
        for (int r = 0; r < TYV; r++) {
            uc[r] = add_vec(uc[r], pe_vec(Y[1 + r]));
            ter[r] = ter[r] + pe_right(Y[1 + r]);
        }
        hlo = add_vec(hlo, pe_vec(Y[0])); hhi = add_vec(hhi, pe_vec(Y[5]));
    }
    publish_edges((z0 - 1) & 1, uc, ter);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TYO; r++) { vm[r] = (vec)(0); vc[r] = (vec)(0); bq[r] = (vec)(0); }

    for (int p = z0 - 1; p <= z1; p++) {
        const long long po = plane_of(p);
        const T *pu = u + po;
        // Planes p = -1 and p = nz of the global grid (first / last chunk only) are evaluated like any other, on the
        // ghost planes: their v only feeds the outputs of the Dirichlet planes 0 and nz-1, which are selected from rhs.
        // (A uniform test that skipped them cost ~45 zero-initialising moves per plane step and 5-9 VGPRs: the
        // prolongation-folding variant spilled two of them, 1.00-1.08 -> 0.975 ms per launch without the test.)
        vec b[TYV], v[TYV];
        T vtail[TYV];
#pragma unroll
        for (int r = 0; r < TYV; r++) {
            up[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (plane_of(p + 1) + urow[r])) + x0);
            vtail[r] = 0;
        }
        // ---- every load of this step first ...
        T Ra[4][NR], Rb[4][NR];  // CORR: raw coarse values under plane p+1
        if (CORR) { raw_a(p + 1, Ra); raw_b(p + 1, Rb); }
        // plane p+1's tail value (and, CORR, halo rows) ride along with `up`: consumed in the next step
        vec hlo_n = (vec)(0), hhi_n = (vec)(0);
        if (CORR) {
            hlo_n = *(const vec *)((u + (plane_of(p + 1) + urow_lo)) + x0);
            hhi_n = *(const vec *)((u + (plane_of(p + 1) + urow_hi)) + x0);
        }
        T ter_n[TYV];
#pragma unroll
        for (int r = 0; r < TYV; r++) {
            ter_n[r] = 0;
            if (tail && !ZEROU) ter_n[r] = (u + (plane_of(p + 1) + urow[r]))[x0 + V];
        }
        {
            if (!CORR && !ZEROU) {
                hlo = *(const vec *)((pu + urow_lo) + x0);
                hhi = *(const vec *)((pu + urow_hi) + x0);
            }
#pragma unroll
            for (int r = 0; r < TYV; r++) {
                b[r] = *(const vec *)((rhs + (po + urow[r])) + x0);
                if (tail) vtail[r] = (rhs + (po + urow[r]))[x0 + V];  // first sweep on the Dirichlet column: v = rhs
            }
        }
        // ---- ... then the arithmetic
        if (CORR) {
            T Z[4][NR], Yn[6][NR];  // correction rows of plane p+1
            zfin(p + 1, Ra, Rb, Z); yrows(Z, Yn);
#pragma unroll
            for (int r = 0; r < TYV; r++) {
                up[r] = add_vec(up[r], pe_vec(Yn[1 + r]));
                ter_n[r] = ter_n[r] + pe_right(Yn[1 + r]);
            }
            hlo_n = add_vec(hlo_n, pe_vec(Yn[0])); hhi_n = add_vec(hhi_n, pe_vec(Yn[5]));
        }
        publish_edges((p + 1) & 1, up, ter_n);
        {
            const bool zbp = (gzo + p == 0) || (gzo + p == gzn - 1);
#pragma unroll
            for (int r = 0; r < TYV; r++) {
                // x-neighbours across the wave edges: the neighbouring wave's edge element of plane p
                // (published in the previous step); the row's last thread has the tail column instead
                const T el = uedge[p & 1][r][wl][1];
                const T er = tail ? utail[p & 1][r] : uedge[p & 1][r][wr][0];
                const T xm = from_prev_lane(uc[r][V - 1], el);
                const T xp = from_next_lane(uc[r][0], er);
                const vec ym = (r > 0) ? uc[r > 0 ? r - 1 : 0] : hlo;
                const vec yp = (r < TYV - 1) ? uc[r < TYV - 1 ? r + 1 : 0] : hhi;
                const bool rb = zbp || ybnd[r];
                T num[V], quo[V];
#pragma unroll
                for (int e = 0; e < V; e++) {
                    const T left = (e == 0) ? xm : uc[r][e > 0 ? e - 1 : 0];
                    const T right = (e == V - 1) ? xp : uc[r][e < V - 1 ? e + 1 : 0];
                    T sum = 0;
                    sum += c.cz * um[r][e];
                    sum += c.cy * ym[e];
                    sum += c.cx * left;
                    sum += c.cx * right;
                    sum += c.cy * yp[e];
                    sum += c.cz * up[r][e];
                    num[e] = b[r][e] - sum;
                }
                div_cd_n<T, V>(num, quo, c);
#pragma unroll
                for (int e = 0; e < V; e++) {
                    T jac = quo[e];
                    if (DAMPED) jac = uc[r][e] + omega * (jac - uc[r][e]);
                    v[r][e] = (rb || (x0 + e == 0)) ? b[r][e] : jac;
                    if (RB && (((x0 + e + y0 - 1 + r + gzo + p) & 1) != 0)) v[r][e] = uc[r][e];  // not red: unchanged
                }
                // v(p) goes to its LDS slot at once (the slot held v(p-2), last read before the previous
                // barrier): the registers of v rows 0 and TYV-1 and of vtail are free for the second sweep
                *(vec *)&lds[p & 1][r][V + x0] = v[r];
                if (tail) lds[p & 1][r][V + x0 + V] = vtail[r];
            }
        }
#pragma unroll
        for (int r = 0; r < TYO; r++) vp[r] = v[r + 1];
        // ---- second sweep on plane q = p-1
        const int q = p - 1;
        if (q >= z0 && q < z1) {
            const bool zbq = (gzo + q == 0) || (gzo + q == gzn - 1);
            const int sl = q & 1;
            const long long qo = (long long)q * g.plane;
#pragma unroll
            for (int r = 0; r < TYO; r++) {
                const int y = y0 + r;
                if (y < g.ny) {
                    const int lr = r + 1;
                    const T xm = lds[sl][lr][V + x0 - 1], xp = lds[sl][lr][V + x0 + V];
                    const vec ym = *(const vec *)&lds[sl][lr - 1][V + x0];
                    const vec yp = *(const vec *)&lds[sl][lr + 1][V + x0];
                    const bool rb = zbq || (y == 0) || (y == g.ny - 1);
                    vec res;
                    T num[V], quo[V];
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        const T left = (e == 0) ? xm : vc[r][e > 0 ? e - 1 : 0];
                        const T right = (e == V - 1) ? xp : vc[r][e < V - 1 ? e + 1 : 0];
                        T sum = 0;
                        sum += c.cz * vm[r][e];
                        sum += c.cy * ym[e];
                        sum += c.cx * left;
                        sum += c.cx * right;
                        sum += c.cy * yp[e];
                        sum += c.cz * vp[r][e];
                        num[e] = bq[r][e] - sum;
                    }
                    div_cd_n<T, V>(num, quo, c);
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        T jac = quo[e];
                        if (DAMPED) jac = vc[r][e] + omega * (jac - vc[r][e]);
                        res[e] = (rb || (x0 + e == 0)) ? bq[r][e] : jac;
                        if (RB && (((x0 + e + y + gzo + q) & 1) == 0)) res[e] = vc[r][e];  // not black: unchanged
                    }
                    __builtin_nontemporal_store(res, (vec *)((out + (qo + urow[lr])) + x0));
                    constexpr int LINE = 128 / (int)sizeof(T), TLN = LINE / V;   // lanes that write the tail line
                    if (tailwave && lane >= 64 - TLN) {
                        // column nx-1 (Dirichlet) as one full 128-byte line: value + zero padding
                        const int j = lane - (64 - TLN);
                        const int xs = g.nx - 1 + V * j;
                        const int line_end = ((g.nx - 1) / LINE + 1) * LINE;
                        if (xs < line_end) {
                            const long long rb0 = qo + urow[lr];
                            vec tv = (vec)(0);
                            if (j == 0) tv[0] = rhs[rb0 + g.nx - 1];
                            __builtin_nontemporal_store(tv, (vec *)(out + rb0 + xs));
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TYV; r++) { um[r] = uc[r]; uc[r] = up[r]; }
#pragma unroll
        for (int r = 0; r < TYO; r++) { vm[r] = vc[r]; vc[r] = vp[r]; bq[r] = b[r + 1]; }
        if (CORR) { hlo = hlo_n; hhi = hhi_n; }
    }
}

struct FastGrid { int nbx, nby, nbz, grid; };
template <typename T>
FastGrid fast_grid(const Geom &g, int zc = ZC)
{
    constexpr int V = VecOf<T>::V;
    FastGrid f;
    f.nbx = (g.nx / V + 63) / 64;
    f.nby = (g.ny + RY * BW - 1) / (RY * BW);
    f.nbz = (g.nz + zc - 1) / zc;
    // small levels (<= 65^3 in fp64) are latency-bound: every marched plane is one more dependent memory round trip
    // and the grid does not fill the chip anyway -- one plane per workgroup there (7.8 -> ~5 us per sweep at 65^3)
    if (f.nbx * f.nby * f.nbz < 1024) f.nbz = g.nz;
    int nblocks = f.nbx * f.nby * f.nbz;
    f.grid = ((nblocks + 7) / 8) * 8;
    return f;
}

}  // namespace

template <typename T>
bool fast_path_ok(const Geom &g)
{
    constexpr int V = VecOf<T>::V;
    // 3-D, rows long enough to fill a wave, at most the one boundary column left over
    return g.dim == 3 && g.nx >= 33 && (g.nx % V) <= 1 && g.ny >= 3;
}

template <typename T>
int fast_partials_capacity(const Geom &g)
{
    if (!fast_path_ok<T>(g)) return 0;
    const FastGrid f = fast_grid<T>(g);   // upper bound over both march lengths (sub-slabs of a level may pick the other one)
    return std::max(f.grid, ((f.nbx * f.nby * g.nz + 7) / 8) * 8);
}

// arrays larger than this stream through the caches: use non-temporal rhs loads
static bool stream_level(const Geom &g, size_t esize) { return (size_t)g.nz * g.plane * esize >= (size_t)64 << 20; }

template <typename T>
void launch_jacobi_fast(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u,
                        const T *rhs, T *out, bool zero_u)
{
    FastGrid f = fast_grid<T>(g);
    const bool damped = (omega != (T)1), nt = stream_level(g, sizeof(T));
    dim3 gr(f.grid), bl(64 * BW);
#define MG_J(D, N, Z) hipLaunchKernelGGL((k_sweep3d<T, OP_JACOBI, D, true, false, N, Z>), gr, bl, 0, s, g, c, omega, 0, u, rhs, out, (double *)nullptr, f.nbx, f.nby, f.nbz)
    if (zero_u) {
        if (damped) { if (nt) MG_J(true, true, true); else MG_J(true, false, true); }
        else { if (nt) MG_J(false, true, true); else MG_J(false, false, true); }
    } else {
        if (damped) { if (nt) MG_J(true, true, false); else MG_J(true, false, false); }
        else { if (nt) MG_J(false, true, false); else MG_J(false, false, false); }
    }
#undef MG_J
}

template <typename T>
void launch_rb_fast(hipStream_t s, const Geom &g, const Coef<T> &c, int colour, const T *u, const T *rhs, T *out)
{
    FastGrid f = fast_grid<T>(g);
    const bool nt = stream_level(g, sizeof(T));
    dim3 gr(f.grid), bl(64 * BW);
    if (nt) hipLaunchKernelGGL((k_sweep3d<T, OP_RB, false, true, false, true>), gr, bl, 0, s, g, c, (T)1, colour, u, rhs, out, (double *)nullptr, f.nbx, f.nby, f.nbz);
    else hipLaunchKernelGGL((k_sweep3d<T, OP_RB, false, true, false, false>), gr, bl, 0, s, g, c, (T)1, colour, u, rhs, out, (double *)nullptr, f.nbx, f.nby, f.nbz);
}

// returns the number of per-block partials written (0 when no norm was requested)
template <typename T>
int launch_residual_fast(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs,
                         T *r, double *d_partials, bool want_norm)
{
    // the norm-only residual writes nothing, so longer marches (fewer z-halo planes re-read) only help it: 9 planes
    // 0.458 ms at 513^3 against 0.51 with 3, 0.48 with 6, 0.464 with 12 (MG_RES_ZC)
    static const int res_zc = [] { const char *e = getenv("MG_RES_ZC"); return e ? atoi(e) : 9; }();
    FastGrid f = fast_grid<T>(g, (!r && want_norm && res_zc > 0) ? res_zc : ZC);
    const bool nt = stream_level(g, sizeof(T));
    dim3 gr(f.grid), bl(64 * BW);
#define MG_R(S, NO, N) hipLaunchKernelGGL((k_sweep3d<T, OP_RESIDUAL, false, S, NO, N>), gr, bl, 0, s, g, c, (T)1, 0, u, rhs, r, d_partials, f.nbx, f.nby, f.nbz)
    if (r && want_norm) { if (nt) MG_R(true, true, true); else MG_R(true, true, false); }
    else if (r) { if (nt) MG_R(true, false, true); else MG_R(true, false, false); }
    else { if (nt) MG_R(false, true, true); else MG_R(false, true, false); }
#undef MG_R
    return want_norm ? f.grid : 0;
}

// Row widths of the fused kernels: the workgroup is one whole grid row of TPR vectors + the odd column, TPR a
// multiple of the wave size: 64 / 128 / 256 (n = 2^k + 1 grids) and 192 / 384 / 512 (n = 385, 769, 1025 in fp64:
// the reference's own 385 fixture size, and BASELINE config 4's grid in double precision).
static bool j2_row_ok(int v) { return v == 64 || v == 128 || v == 192 || v == 256 || v == 384 || v == 512; }
// 512-thread rows keep two output rows per workgroup (three would need 82 KB of LDS: one workgroup per CU)
static int j2_tyo_for(int tpr, int wanted) { return tpr > 384 ? 2 : wanted; }

// fused double sweep: whole (non-distributed) 3-D level whose rows are exactly 64 ... 512 vectors
// + the odd column; opt out with MG_FUSED_PAIR=0
template <typename T>
bool jacobi2_ok(const Geom &g)
{
    constexpr int V = VecOf<T>::V;
    static const bool enabled = [] { const char *e = getenv("MG_FUSED_PAIR"); return !(e && e[0] == '0'); }();
    if (!enabled || g.dim != 3 || g.gz0 != 0 || g.gnz != g.nz || g.ny < 3 || g.nz < 3) return false;
    const int v = (g.nx - 1) / V;
    return (g.nx - 1) % V == 0 && j2_row_ok(v);
}

// the same kernel on the inner planes of a z-slab (mg_solver.cpp: pair_on_slab_t): `g` = the slab's geometry
template <typename T>
bool jacobi2_slab_ok(const Geom &g)
{
    constexpr int V = VecOf<T>::V;
    static const bool enabled = [] {
        const char *e = getenv("MG_FUSED_PAIR"), *f = getenv("MG_FUSED_SLAB");
        return !(e && e[0] == '0') && !(f && f[0] == '0');
    }();
    if (!enabled || g.dim != 3 || g.ny < 3 || g.nz < 6) return false;
    const int v = (g.nx - 1) / V;
    return (g.nx - 1) % V == 0 && j2_row_ok(v);
}

template <typename T>
int launch_jacobi2(hipStream_t s, const Geom &g, const Coef<T> &c, T omega, const T *u, const T *rhs, T *out, bool zero_u, int dup,
                   double *d_partials)
{
    constexpr int V = VecOf<T>::V;
    if (pair_wide_ok<T>(g)) return launch_pair_wide<T>(s, g, Geom{}, c, omega, u, (const T *)nullptr, rhs, out, zero_u, false, dup, d_partials);
    const int tpr = (g.nx - 1) / V;
    const int ncopy = dup > 0 ? 2 : 1;   // dup: the same geometry once more, `dup` planes further up, in the same launch
    // three output rows per workgroup where the correction is not folded in: 5 instead of 4 first-sweep rows
    // per 3 instead of 2 outputs, 167 VGPRs (still 3 workgroups/CU): 0.86 -> 0.79 ms per pair at 513^3
    static const int tyo_env = [] { const char *e = getenv("MG_J2_TYO"); return e ? atoi(e) : 3; }();
    const int tyo = j2_tyo_for(tpr, tyo_env);
    const int nby = (g.ny + J2_TYO - 1) / J2_TYO, nbz = j2_nbz(g, tpr, tyo == 3 ? 3 : 2);
    const int nblocks = nby * nbz, grid = ((ncopy * nblocks + 7) / 8) * 8;
    const bool damped = (omega != (T)1), nt = stream_level(g, sizeof(T));
    const int nby3 = (g.ny + 2) / 3, grid3 = ((ncopy * nby3 * nbz + 7) / 8) * 8;
#define MG_J2K(TPR, D, N, Z) \
    do { \
        constexpr int TY3 = (TPR > 384) ? 2 : 3; /* never launched with three rows at 512 threads: no such instantiation */ \
        if (tyo == 3) hipLaunchKernelGGL((k_jacobi2<T, TPR, D, N, false, false, Z, TY3>), dim3(grid3), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby3, nbz, (const T *)nullptr, Geom{}, dup); \
        else hipLaunchKernelGGL((k_jacobi2<T, TPR, D, N, false, false, Z>), dim3(grid), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby, nbz, (const T *)nullptr, Geom{}, dup); \
    } while (0)
#define MG_J2(TPR) \
    do { \
        if (zero_u) { if (damped) MG_J2K(TPR, true, false, true); else MG_J2K(TPR, false, false, true); } \
        else if (damped) { if (nt) MG_J2K(TPR, true, true, false); else MG_J2K(TPR, true, false, false); } \
        else { if (nt) MG_J2K(TPR, false, true, false); else MG_J2K(TPR, false, false, false); } \
    } while (0)
    switch (tpr) {
    case 512: MG_J2(512); break;
    case 384: MG_J2(384); break;
    case 256: MG_J2(256); break;
    case 192: MG_J2(192); break;
    case 128: MG_J2(128); break;
    default: MG_J2(64); break;
    }
#undef MG_J2
#undef MG_J2K
    return 0;
}

// one red-black Gauss-Seidel sweep (both colours) in one pass: out = RB(u)
template <typename T>
bool rb_fused_ok(const Geom &g)
{
    static const bool enabled = [] { const char *e = getenv("MG_FUSED_RB"); return !(e && e[0] == '0'); }();
    return enabled && jacobi2_ok<T>(g);
}

// coarse != nullptr: the sweep reads u + P coarse (prolong-add folded in, like launch_jacobi2_corr)
// zero_u: u is identically zero and is not read (the first pre-smoothing sweep of a coarse level: its memset is skipped)
// d_partials (wide tiles only): the sweep also leaves sum (rhs - A u)^2 of its INPUT as one partial sum per workgroup; returns
// how many were written (0: this launch did not compute them)
template <typename T>
int launch_rb_fused(hipStream_t s, const Geom &g, const Coef<T> &c, const T *u, const T *rhs, T *out,
                    const T *coarse, const Geom &gc, int dup, bool zero_u, double *d_partials)
{
    constexpr int V = VecOf<T>::V;
    if (coarse) { dup = 0; zero_u = false; }   // the folding variant only runs on whole levels
    if (pair_wide_ok<T>(g)) return launch_pair_wide<T>(s, g, gc, c, (T)1, u, coarse, rhs, out, zero_u, true, dup, d_partials);
    const int tpr = (g.nx - 1) / V;
    const int ncopy = (dup > 0 && !coarse) ? 2 : 1;
    static const int tyo_env = [] { const char *e = getenv("MG_J2_TYO"); return e ? atoi(e) : 3; }();
    const int tyo = j2_tyo_for(tpr, tyo_env);
    const int nby = (g.ny + J2_TYO - 1) / J2_TYO, nbz = j2_nbz(g, tpr, (tyo == 3 && !coarse) ? 3 : 2);
    const int nblocks = nby * nbz, grid = ((ncopy * nblocks + 7) / 8) * 8;
    const int nby3 = (g.ny + 2) / 3, grid3 = ((ncopy * nby3 * nbz + 7) / 8) * 8;
#define MG_RB2(TPR) \
    do { \
        if (coarse) hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, true, true>), dim3(grid), dim3(TPR), 0, s, g, c, (T)1, u, rhs, out, nby, nbz, coarse, gc, 0); \
        else if (tyo == 3 && zero_u) hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, false, true, true, (TPR > 384) ? 2 : 3>), dim3(grid3), dim3(TPR), 0, s, g, c, (T)1, u, rhs, out, nby3, nbz, (const T *)nullptr, Geom{}, dup); \
        else if (tyo == 3) hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, false, true, false, (TPR > 384) ? 2 : 3>), dim3(grid3), dim3(TPR), 0, s, g, c, (T)1, u, rhs, out, nby3, nbz, (const T *)nullptr, Geom{}, dup); \
        else if (zero_u) hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, false, true, true>), dim3(grid), dim3(TPR), 0, s, g, c, (T)1, u, rhs, out, nby, nbz, (const T *)nullptr, Geom{}, dup); \
        else hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, false, true>), dim3(grid), dim3(TPR), 0, s, g, c, (T)1, u, rhs, out, nby, nbz, (const T *)nullptr, Geom{}, dup); \
    } while (0)
    switch (tpr) {
    case 512: MG_RB2(512); break;
    case 384: MG_RB2(384); break;
    case 256: MG_RB2(256); break;
    case 192: MG_RB2(192); break;
    case 128: MG_RB2(128); break;
    default: MG_RB2(64); break;
    }
#undef MG_RB2
    return 0;
}

template bool rb_fused_ok<double>(const Geom &);
template bool rb_fused_ok<float>(const Geom &);
template int launch_rb_fused<double>(hipStream_t, const Geom &, const Coef<double> &, const double *, const double *, double *, const double *, const Geom &, int, bool, double *);
template int launch_rb_fused<float>(hipStream_t, const Geom &, const Coef<float> &, const float *, const float *, float *, const float *, const Geom &, int, bool, double *);

// prolong-add + two Jacobi sweeps in one pass: out = J(J(u + P coarse))
template <typename T>
bool jacobi2_corr_ok(const Geom &gf, const Geom &gc)
{
    return jacobi2_ok<T>(gf) && gc.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 &&
           gf.nz == 2 * gc.nz - 1 && gc.gz0 == 0 && gc.gnz == gc.nz;
}

// the same on the pieces of a z-slab (both levels distributed, the coarse slab = the planes that coincide with the fine slab's):
// the coarse correction is addressed by global plane index, its two ghost planes either side must be valid
template <typename T>
bool jacobi2_corr_slab_ok(const Geom &gf, const Geom &gc)
{
    static const bool enabled = [] { const char *e = getenv("MG_FUSED_PROLONG_SLAB"); return !(e && e[0] == '0'); }();
    return enabled && jacobi2_slab_ok<T>(gf) && gc.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 &&
           gf.gnz == 2 * gc.gnz - 1 && gf.gz0 == 2 * gc.gz0 && (gf.nz == 2 * gc.nz || gf.nz == 2 * gc.nz - 1) && gc.nz >= 2;
}

template <typename T>
void launch_jacobi2_corr(hipStream_t s, const Geom &g, const Geom &gc, const Coef<T> &c, T omega, const T *u,
                         const T *coarse, const T *rhs, T *out, int dup)
{
    constexpr int V = VecOf<T>::V;
    if (pair_wide_ok<T>(g)) { launch_pair_wide<T>(s, g, gc, c, omega, u, coarse, rhs, out, false, false, dup); return; }
    const int tpr = (g.nx - 1) / V;
    const int nby = (g.ny + J2_TYO - 1) / J2_TYO, nbz = j2_nbz(g, tpr, 2);
    const int nblocks = nby * nbz, grid = (((dup > 0 ? 2 : 1) * nblocks + 7) / 8) * 8;
    const bool damped = (omega != (T)1);
    static const int minw_env = [] { const char *e = getenv("MG_J2C_MINW"); return e ? atoi(e) : 0; }();
    // fp32 (four floats per lane, three coarse values per row) needs 180 VGPRs: held to 168 it spills 12 of them and runs
    // 5.5 ms per launch at 1025^3; at two waves per SIMD, unspilled, 4.8 ms. fp64 fits 163.
    const int minw = minw_env ? minw_env : (sizeof(T) == 4 ? 2 : 3);
    // fp32: two floats per lane (rows of 2 * tpr lanes) up to 256-lane rows: 92 instead of 180 VGPRs, 0.590 against 0.618 ms per
    // launch at 513^3; at 1025^3 the row would be a 512-thread workgroup (eight waves on one barrier per plane) and the four-float
    // kernel at two waves per SIMD wins, 5.12 against 5.49 ms (MG_J2C_V2=0: four floats per lane everywhere)
    static const bool v2_env = [] { const char *e = getenv("MG_J2C_V2"); return !(e && e[0] == '0'); }();
    if constexpr (sizeof(T) == 4) {
        if (v2_env && 2 * tpr <= 256) {
            const int nby2 = (g.ny + 1) / 2, nbz2 = j2_nbz(g, 2 * tpr, 2);
            const int grid2 = (((dup > 0 ? 2 : 1) * nby2 * nbz2 + 7) / 8) * 8;
#define MG_J2C2(TPR2) \
            do { \
                if (damped) hipLaunchKernelGGL((k_jacobi2<T, TPR2, true, true, true, false, false, 2, 3, 2>), dim3(grid2), dim3(TPR2), 0, s, g, c, omega, u, rhs, out, nby2, nbz2, coarse, gc, dup); \
                else hipLaunchKernelGGL((k_jacobi2<T, TPR2, false, true, true, false, false, 2, 3, 2>), dim3(grid2), dim3(TPR2), 0, s, g, c, omega, u, rhs, out, nby2, nbz2, coarse, gc, dup); \
            } while (0)
            switch (2 * tpr) {
            case 256: MG_J2C2(256); return;
            case 128: MG_J2C2(128); return;
            default: break;
            }
#undef MG_J2C2
        }
    }
#define MG_J2C(TPR) \
    do { \
        if (minw == 2) { \
            if (damped) hipLaunchKernelGGL((k_jacobi2<T, TPR, true, true, true, false, false, 2, 2>), dim3(grid), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby, nbz, coarse, gc, dup); \
            else hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, true, false, false, 2, 2>), dim3(grid), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby, nbz, coarse, gc, dup); \
        } else { \
            if (damped) hipLaunchKernelGGL((k_jacobi2<T, TPR, true, true, true>), dim3(grid), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby, nbz, coarse, gc, dup); \
            else hipLaunchKernelGGL((k_jacobi2<T, TPR, false, true, true>), dim3(grid), dim3(TPR), 0, s, g, c, omega, u, rhs, out, nby, nbz, coarse, gc, dup); \
        } \
    } while (0)
    switch (tpr) {
    case 512: MG_J2C(512); break;
    case 384: MG_J2C(384); break;
    case 256: MG_J2C(256); break;
    case 192: MG_J2C(192); break;
    case 128: MG_J2C(128); break;
    default: MG_J2C(64); break;
    }
#undef MG_J2C
}

template bool jacobi2_corr_ok<double>(const Geom &, const Geom &);
template bool jacobi2_corr_ok<float>(const Geom &, const Geom &);
template void launch_jacobi2_corr<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, double, const double *, const double *, const double *, double *, int);
template void launch_jacobi2_corr<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, float, const float *, const float *, const float *, float *, int);
template bool jacobi2_corr_slab_ok<double>(const Geom &, const Geom &);
template bool jacobi2_corr_slab_ok<float>(const Geom &, const Geom &);
template bool jacobi2_ok<double>(const Geom &);
template bool jacobi2_ok<float>(const Geom &);
template int launch_jacobi2<double>(hipStream_t, const Geom &, const Coef<double> &, double, const double *, const double *, double *, bool, int, double *);
template bool jacobi2_slab_ok<double>(const Geom &);
template bool jacobi2_slab_ok<float>(const Geom &);
template int launch_jacobi2<float>(hipStream_t, const Geom &, const Coef<float> &, float, const float *, const float *, float *, bool, int, double *);
template bool fast_path_ok<double>(const Geom &);
template bool fast_path_ok<float>(const Geom &);
template int fast_partials_capacity<double>(const Geom &);
template int fast_partials_capacity<float>(const Geom &);
template void launch_jacobi_fast<double>(hipStream_t, const Geom &, const Coef<double> &, double, const double *, const double *, double *, bool);
template void launch_jacobi_fast<float>(hipStream_t, const Geom &, const Coef<float> &, float, const float *, const float *, float *, bool);
template void launch_rb_fast<double>(hipStream_t, const Geom &, const Coef<double> &, int, const double *, const double *, double *);
template void launch_rb_fast<float>(hipStream_t, const Geom &, const Coef<float> &, int, const float *, const float *, float *);
template int launch_residual_fast<double>(hipStream_t, const Geom &, const Coef<double> &, const double *, const double *, double *, double *, bool);
template int launch_residual_fast<float>(hipStream_t, const Geom &, const Coef<float> &, const float *, const float *, float *, double *, bool);

}  // namespace mg
