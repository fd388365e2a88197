// mg_pair_wide.hip -- the V-cycle's smoothing pairs on HBM-resident levels, wide-tile form (round 3), gfx950.
//
//     out = J(J(u))            pre-smoothing pair            (reference sweep: include/solvers.hpp:64-83, twice)
//     out = J(J(u + P e))      post-smoothing pair with the prolongation (src/multigrid.cpp:3-27) folded in
//     out = J(J(0))            pre-smoothing pair of a level that starts from the zero guess
//
// Why a second form of k_jacobi2 (mg_jacobi_fast.hip). That kernel keeps a workgroup's whole state in the registers of
// ONE thread column: TYO + 2 rows of u per thread for TYO output rows, so the first sweep is evaluated on (TYO + 2) / TYO
// = 1.67 x (plain) or 2 x (folding variant) the rows it owns and the coarse correction on 3 x; its run time followed
// the count of fp64 instructions per output point (75, 67, 88 per point: 0.86, 0.79, 1.03 ms at 513^3), not the bytes it
// moves -- at three waves per SIMD a wave issues one fp64 instruction per 8 cycles and the vector pipe idles whenever
// fewer than two of the three are runnable.  Here a workgroup is a TILE of G groups x R = 2 rows: every thread owns two
// rows (an even one and the odd one above it), G groups of row-wide wave teams are stacked in y, and neighbouring groups
// exchange their rows through LDS instead of recomputing them:
//   * first sweep on G*R rows for G*R - 2 output rows: 8 / 6 = 1.33 x at 513^3 fp64 (G = 4), 16 / 14 at 257^3 (G = 8);
//   * the correction P e is evaluated once per owned row (the two tile-edge groups add one halo row each);
//   * 10 persistent 16-byte registers per thread instead of 24-32: <= 128 VGPRs, four waves per SIMD; one workgroup of
//     1024 threads per CU;
//   * per plane step: u(p+1) (corrected) -> LDS slot A; first sweep on plane p reads its x/y neighbours from LDS slot
//     B (u(p), published one step earlier) and the z neighbours from registers; v(p) -> LDS; second sweep on plane p-1
//     reads v(p-1)'s x/y neighbours from LDS and v(p-2), v(p) from registers; ONE barrier per plane.
// Per-point arithmetic, its order, -ffp-contract=off and the correctly rounded division are those of k_jacobi2 and of
// two k_sweep3d launches: the output is bit-identical (tests/test_gpu_parity.py compares with the oracle).
// Geometry conventions (z-slab pieces, ghost planes, the second copy `dup_planes` further up, coarse planes addressed by
// global plane index) are exactly k_jacobi2's, so the launchers of mg_jacobi_fast.hip hand over to this kernel unchanged.
// MFMA unused: a 7-point stencil is not a contraction; the kernel is bound by HBM traffic (24-25 B per point).
#include "mg_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace mg {
namespace {

// value of lane l (a compile-time constant) in every lane: v_readlane, no LDS
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_bcast(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <typename T> struct WV;
template <> struct WV<double> { static constexpr int V = 2; };
template <> struct WV<float> { static constexpr int V = 4; };

// RB: the pipeline runs red-black Gauss-Seidel instead of Jacobi (see k_jacobi2): phase 1 = red half-sweep (black
// points copied), phase 2 = black half-sweep on the plane behind -- ONE red-black sweep per pass
// NORM: the launch also returns sum r^2 of r = rhs - A u (the residual of the INPUT, Residual::apply_iteration_to_vec +
// Norm of the reference, include/solvers.hpp:257-307): the first sweep has every operand of r in registers, so the outer
// loop's convergence test (src/main.cpp:86-89) costs a few more instructions per point instead of a pass over HBM.
// One partial sum per workgroup -> partials[blockIdx.x] (fixed order inside the workgroup; k_reduce_final adds them).
template <typename T, int TPR, int G, bool DAMPED, bool CORR, bool ZEROU, bool RB = false, bool NORM = false>
__global__ __launch_bounds__(TPR * G) void k_pairw(Geom g, Coef<T> c, T omega, const T *__restrict__ u_,
                                                   const T *__restrict__ rhs_, T *__restrict__ out_, int nby, int zc,
                                                   const T *__restrict__ coarse, Geom gc, int dup_planes,
                                                   double *__restrict__ partials)
{
    constexpr int V = WV<T>::V, CV = V / 2, NR = CV + 1;
    constexpr int R = 2, NROW = G * R, S = NROW - 2;
    constexpr int LP = TPR * V + 2 * V;  // LDS row: V pad | TPR*V values | tail column | pad
    typedef T vec __attribute__((ext_vector_type(V)));
    static_assert(G >= 2 && (TPR % 64) == 0, "row-wide wave teams, at least two groups");
    // u planes p (read) / p+1 (written): rows 1 .. NROW = the tile, row 0 / NROW+1 = the halo rows below / above it
    __shared__ __align__(16) T su[ZEROU ? 1 : 2][ZEROU ? 1 : NROW + 2][ZEROU ? V : LP];
    __shared__ __align__(16) T sv[2][NROW][LP];  // first-sweep planes p-1 (read) / p (written)
    __shared__ double snorm[NORM ? TPR * G / 64 : 1];

    // Work = (copies x) y-tiles x planes. Two ways of dealing it (the launcher's wide_plan picks; zc > 0 is the default):
    //  * zc > 0: z-CHUNKS of zc planes, item = (copy, chunk, tile) with the tile running fastest; workgroup wi takes items wi,
    //    wi + gridDim.x, ... -- with one workgroup per item (the default grid) the loop below runs once;
    //  * zc == 0: the tile-planes are cut into gridDim.x equal RANGES of consecutive planes of consecutive tiles: a workgroup
    //    marches its range, changing tile (new prologue) at most once or twice; every workgroup gets the same number of plane
    //    steps and a chunk boundary -- two planes of first-sweep work done twice -- only exists where a range starts.
    const int t = threadIdx.x, lane = t & 63;
    const int grp = __builtin_amdgcn_readfirstlane(t / TPR);  // y-group of this wave: scalar
    const int xt = t - grp * TPR;                             // lane position in the row
    const int x0 = V * xt;                                    // the gate guarantees nx - 1 == TPR * V
    const bool tail = (xt == TPR - 1);                        // last thread of the row: also owns the Dirichlet column nx-1
    const bool tailwave = (xt >> 6) == (TPR >> 6) - 1;
    const int i0 = grp * R;                                   // this thread: tile rows i0, i0+1
    const bool lo_grp = (grp == 0), hi_grp = (grp == G - 1);  // the groups that also fetch a halo row of u
    double nsq = 0.;
    const long long per_copy = (long long)nby * g.nz, total = (dup_planes > 0 ? 2 : 1) * per_copy;
    const int nwg = (int)gridDim.x, wper = nwg >> 3;          // the launcher makes the grid a multiple of 8
    const int wi = (blockIdx.x & 7) * wper + (blockIdx.x >> 3);  // XCD-aware order: an XCD takes consecutive ranges
    // (chunks: y-neighbouring tiles sit on neighbouring workgroups and march the same planes at the same time, so the halo rows
    // they share are cache hits; ranges keep that only when a range is a whole fraction of a tile's column)
    const int nbz = zc > 0 ? (g.nz + zc - 1) / zc : 0;
    const long long items = (dup_planes > 0 ? 2 : 1) * (long long)nby * nbz;
    long long w0 = zc > 0 ? wi : total * wi / nwg;
    const long long w1 = zc > 0 ? items : total * (wi + 1) / nwg;
    while (w0 < w1) {
    bool second;
    int by, z0, z1;
    if (zc > 0) {
        const long long per = (long long)nby * nbz;
        second = w0 >= per;
        const long long wr = w0 - (second ? per : 0);
        const int bz = (int)(wr / nby);
        by = (int)(wr - (long long)bz * nby);
        z0 = bz * zc; z1 = min(z0 + zc, g.nz);
        w0 += nwg;
    } else {
        second = w0 >= per_copy;
        const long long wr = w0 - (second ? per_copy : 0);
        by = (int)(wr / g.nz);
        z0 = (int)(wr - (long long)by * g.nz); z1 = (int)min((long long)g.nz, z0 + (w1 - w0));
        w0 += z1 - z0;
    }
    const long long dup_off = second ? (long long)dup_planes * g.plane : 0;
    const T *__restrict__ u = u_ + dup_off;
    const T *__restrict__ rhs = rhs_ + dup_off;
    T *__restrict__ out = out_ + dup_off;
    const int Y0 = by * S;                                    // tile rows Y0 .. Y0+NROW-1

    long long urow[R];
    bool ybnd[R], outrow[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = Y0 + i0 + r, yc = min(y, g.ny - 1);
        ybnd[r] = (yc == 0) || (yc == g.ny - 1);
        urow[r] = (long long)yc * g.pitch;
        const int i = i0 + r;
        // second-sweep (output) rows of this tile: its inner rows, and row 0 of the grid
        outrow[r] = (i >= 1 || by == 0) && (i <= NROW - 2) && (y < g.ny);
    }
    const int hy = lo_grp ? max(Y0 - 1, 0) : min(Y0 + NROW, g.ny - 1);
    const long long hrow = (long long)hy * g.pitch;
    const int hs = lo_grp ? 0 : NROW + 1;  // its LDS row

    const int zhalo = (g.gnz != g.nz) ? 2 : 1;
    const int gzo = g.gz0 + (second ? dup_planes : 0), gzn = g.gnz;   // global z of local plane 0 / global plane count
    auto plane_of = [&](int p) { return (long long)min(max(p, -zhalo), g.nz - 1 + zhalo) * g.plane; };

    // ---- on-the-fly prolongation (CORR): coarse rows A = (Y0+i0)/2 under the even row, B = A+1; the odd row is their
    // mean. Tile-edge groups: the halo row below the tile is odd (mean of A-1 and A), the one above it even (= B).
    const T hf = (T)0.5;
    typedef T cwide __attribute__((ext_vector_type(NR <= 2 ? 2 : 4), aligned(CV * sizeof(T))));
    int crow[3] = {0, 0, 0};  // offsets of coarse rows A-1, A, B (clamped into the grid): workgroup-uniform per wave
    const int ic0 = CV * xt;
    if (CORR) {
        const int A = (Y0 + i0) >> 1;
        crow[0] = min(max(A - 1, 0), gc.ny - 1) * gc.pitch;
        crow[1] = min(A, gc.ny - 1) * gc.pitch;
        crow[2] = min(A + 1, gc.ny - 1) * gc.pitch;
    }
    auto cplane = [&](int P, int up1) {
        const int kc = ((gzo + P + up1) >> 1) - gc.gz0;
        return coarse + (long long)min(max(kc, -zhalo), gc.nz - 1 + zhalo) * gc.plane;
    };
    auto load_crow = [&](const T *base, int j, T (&d)[NR]) {
        const cwide w = *(const cwide *)((base + crow[j]) + ic0);
#pragma unroll
        for (int m = 0; m < NR; m++) d[m] = w[m];
    };
    // raw coarse values under fine plane P: Ra = lower (or only) coarse plane, Rb = the upper one of an odd plane
    auto raw = [&](int P, T (&Ra)[3][NR], T (&Rb)[3][NR]) {
        const T *c0 = cplane(P, 0);
        const bool odd = ((gzo + P) & 1) != 0;
        const T *c1 = cplane(P, 1);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j == 0 && !lo_grp) {
#pragma unroll
                for (int m = 0; m < NR; m++) { Ra[j][m] = 0; Rb[j][m] = 0; }
                continue;
            }
            load_crow(c0, j, Ra[j]);
            if (odd) load_crow(c1, j, Rb[j]);
            else {
#pragma unroll
                for (int m = 0; m < NR; m++) Rb[j][m] = 0;
            }
        }
    };
    // z phase: correction rows interpolated onto fine plane P (zero outside the grid)
    auto zfin = [&](int P, const T (&Ra)[3][NR], const T (&Rb)[3][NR], T (&Z)[3][NR]) {
        const int gP = gzo + P;
        const bool in = (gP >= 0) && (gP < gzn), odd = (gP & 1) != 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int m = 0; m < NR; m++) {
                const T z = odd ? hf * (Ra[j][m] + Rb[j][m]) : Ra[j][m];
                Z[j][m] = in ? z : (T)0;
            }
        }
    };
    // x phase on a y-interpolated row: own vector / the tail column
    auto pe_vec = [&](const T (&Yr)[NR]) {
        vec w;
#pragma unroll
        for (int mm = 0; mm < CV; mm++) {
            w[2 * mm] = Yr[mm];
            w[2 * mm + 1] = hf * (Yr[mm] + Yr[mm + 1]);
        }
        return w;
    };
    auto add_vec = [&](vec a, vec w) {
        vec o;
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = a[e] + w[e];
        return o;
    };
    // u + P e on the thread's two rows (w), their tail-column values (tl) and, tile-edge groups, the halo row (h, htl)
    auto correct = [&](const T (&Z)[3][NR], vec (&w)[R], T (&tl)[R], vec &h, T &htl, bool with_halo) {
        T Yodd[NR];
#pragma unroll
        for (int m = 0; m < NR; m++) Yodd[m] = hf * (Z[1][m] + Z[2][m]);
        w[0] = add_vec(w[0], pe_vec(Z[1]));
        w[1] = add_vec(w[1], pe_vec(Yodd));
        tl[0] = tl[0] + Z[1][CV];
        tl[1] = tl[1] + Yodd[CV];
        if (with_halo) {
            if (lo_grp) {
                T Yh[NR];
#pragma unroll
                for (int m = 0; m < NR; m++) Yh[m] = hf * (Z[0][m] + Z[1][m]);
                h = add_vec(h, pe_vec(Yh));
                htl = htl + Yh[CV];
            } else if (hi_grp) {
                h = add_vec(h, pe_vec(Z[2]));
                htl = htl + Z[2][CV];
            }
        }
    };

    // one point-Jacobi / Gauss-Seidel-colour update of a vector: neighbours in z (zm, zp), y (ym, yp), x (xm, xp)
    auto update = [&](const vec &zm, const vec &cc, const vec &zp, const vec &ym, const vec &yp, T xm, T xp,
                      const vec &bb, bool rb) {
        T num[V], quo[V];
        vec res;
#pragma unroll
        for (int e = 0; e < V; e++) {
            const T left = (e == 0) ? xm : cc[e > 0 ? e - 1 : 0];
            const T right = (e == V - 1) ? xp : cc[e < V - 1 ? e + 1 : 0];
            T sum = 0;
            sum += c.cz * zm[e];
            sum += c.cy * ym[e];
            sum += c.cx * left;
            sum += c.cx * right;
            sum += c.cy * yp[e];
            sum += c.cz * zp[e];
            num[e] = bb[e] - sum;
        }
        // wave-uniform fallback test in fp64 (fewer execution-mask regions: -2 % per launch at 513^3); the per-lane form in
        // fp32, where the uniform form's longer live ranges spill (1025^3 pair 2.6 -> 3.1 ms)
        if constexpr (sizeof(T) == 8) div_cd_n_wave<T, V>(num, quo, c);
        else div_cd_n<T, V>(num, quo, c);
#pragma unroll
        for (int e = 0; e < V; e++) {
            T jac = quo[e];
            if (DAMPED) jac = cc[e] + omega * (jac - cc[e]);
            res[e] = (rb || (x0 + e == 0)) ? bb[e] : jac;
        }
        return res;
    };

    // sum over the vector of (rhs - A u)^2, the reference's residual expression term by term (k_sweep3d<OP_RESIDUAL>): the
    // products and the first three partial sums are the update's own
    auto resid_sq = [&](const vec &zm, const vec &cc, const vec &zp, const vec &ym, const vec &yp, T xm, T xp,
                        const vec &bb, bool rb) {
        double sq = 0.;
#pragma unroll
        for (int e = 0; e < V; e++) {
            const T left = (e == 0) ? xm : cc[e > 0 ? e - 1 : 0];
            const T right = (e == V - 1) ? xp : cc[e < V - 1 ? e + 1 : 0];
            T sum = 0;
            sum += c.cz * zm[e];
            sum += c.cy * ym[e];
            sum += c.cx * left;
            sum += c.cd * cc[e];
            sum += c.cx * right;
            sum += c.cy * yp[e];
            sum += c.cz * zp[e];
            if (rb || (x0 + e == 0)) sum = (T)1 * cc[e];
            const T res = bb[e] - sum;
            sq += (double)res * (double)res;
        }
        return sq;
    };
    vec um[R], uc[R], up[R], vm[R], vc[R], bq[R];
    // ---- prologue: planes z0-2 and z0-1 of u (corrected), plane z0-1 published
    {
        T ter[R];
        vec hh = (vec)(0);
        T hter = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            um[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (plane_of(z0 - 2) + urow[r])) + x0);
            uc[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (plane_of(z0 - 1) + urow[r])) + x0);
            ter[r] = 0;
            if (tail && !ZEROU) ter[r] = (u + (plane_of(z0 - 1) + urow[r]))[x0 + V];
        }
        if (!ZEROU && (lo_grp || hi_grp)) {
            hh = *(const vec *)((u + (plane_of(z0 - 1) + hrow)) + x0);
            if (tail) hter = (u + (plane_of(z0 - 1) + hrow))[x0 + V];
        }
        if (CORR) {
            T Ra[3][NR], Rb[3][NR], Z[3][NR];
            T dummy_t[R] = {0, 0};
            vec dummy_h = (vec)(0);
            T dummy_ht = 0;
            raw(z0 - 2, Ra, Rb); zfin(z0 - 2, Ra, Rb, Z);
            correct(Z, um, dummy_t, dummy_h, dummy_ht, false);
            raw(z0 - 1, Ra, Rb); zfin(z0 - 1, Ra, Rb, Z);
            correct(Z, uc, ter, hh, hter, true);
        }
        if constexpr (!ZEROU) {
            const int sl = (z0 - 1) & 1;
#pragma unroll
            for (int r = 0; r < R; r++) {
                *(vec *)&su[sl][1 + i0 + r][V + x0] = uc[r];
                if (tail) su[sl][1 + i0 + r][V + x0 + V] = ter[r];
                if (xt == 0) { su[0][1 + i0 + r][V - 1] = 0; su[1][1 + i0 + r][V - 1] = 0; }
            }
            if (lo_grp || hi_grp) {
                *(vec *)&su[sl][hs][V + x0] = hh;
                if (tail) su[sl][hs][V + x0 + V] = hter;
                if (xt == 0) { su[0][hs][V - 1] = 0; su[1][hs][V - 1] = 0; }
            }
        }
        if (xt == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) { sv[0][i0 + r][V - 1] = 0; sv[1][i0 + r][V - 1] = 0; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) { vm[r] = (vec)(0); vc[r] = (vec)(0); bq[r] = (vec)(0); }

    // Software prefetch, one plane step ahead: the u rows of plane p+2 and the rhs rows of plane p+1 are requested in the
    // middle of step p (see there) and consumed at the top of step p+1, so a step's HBM latency runs under the arithmetic
    // around it (all 16 waves of the tile meet at one barrier per plane: without it the CU alternates between waiting and
    // computing). Two steps ahead for the rhs rows, which pays in k_rrw, spills 50 registers here.
    vec nu[R], nb[R], nh = (vec)(0);
    T nter[R], nvt[R], nhter = 0;
    auto fetch_u = [&](int pu1) {
        const long long pn = plane_of(pu1);
#pragma unroll
        for (int r = 0; r < R; r++) nu[r] = ZEROU ? (vec)(0) : *(const vec *)((u + (pn + urow[r])) + x0);
        if (!ZEROU && (lo_grp || hi_grp)) nh = *(const vec *)((u + (pn + hrow)) + x0);
#pragma unroll
        for (int r = 0; r < R; r++) nter[r] = 0;
        if (tail) {
#pragma unroll
            for (int r = 0; r < R; r++) if (!ZEROU) nter[r] = (u + (pn + urow[r]))[x0 + V];
            if (!ZEROU && (lo_grp || hi_grp)) nhter = (u + (pn + hrow))[x0 + V];
        }
    };
    auto fetch_b = [&](int pb) {
        const long long po = plane_of(pb);
#pragma unroll
        for (int r = 0; r < R; r++) nb[r] = *(const vec *)((rhs + (po + urow[r])) + x0);
#pragma unroll
        for (int r = 0; r < R; r++) nvt[r] = 0;
        if (tail) {
#pragma unroll
            for (int r = 0; r < R; r++) nvt[r] = (rhs + (po + urow[r]))[x0 + V];
        }
    };
    auto fetch = [&](int pu1, int pb) { fetch_u(pu1); fetch_b(pb); };
    fetch(z0, z0 - 1);

    for (int p = z0 - 1; p <= z1; p++) {
        // ---- this step's operands were requested one step ago (between the two sweeps of the last step)
        vec b[R], hn = nh;
        T vtail[R], ter_n[R], hter_n = nhter;
#pragma unroll
        for (int r = 0; r < R; r++) { up[r] = nu[r]; b[r] = nb[r]; ter_n[r] = nter[r]; vtail[r] = nvt[r]; }
        T Ra[3][NR], Rb[3][NR];
        if (CORR) raw(p + 1, Ra, Rb);
        // ---- u(p+1) + P e -> LDS slot (p+1)&1 (read by the next step's first sweep)
        if (CORR) {
            T Z[3][NR];
            zfin(p + 1, Ra, Rb, Z);
            correct(Z, up, ter_n, hn, hter_n, true);
        }
        if constexpr (!ZEROU) {
            const int sn = (p + 1) & 1;
#pragma unroll
            for (int r = 0; r < R; r++) *(vec *)&su[sn][1 + i0 + r][V + x0] = up[r];
            if (lo_grp || hi_grp) *(vec *)&su[sn][hs][V + x0] = hn;
        }
        if (tail) {   // the Dirichlet column: u(p+1) and v(p) = rhs(p) in one predicated region
#pragma unroll
            for (int r = 0; r < R; r++) {
                if constexpr (!ZEROU) su[(p + 1) & 1][1 + i0 + r][V + x0 + V] = ter_n[r];
                sv[p & 1][i0 + r][V + x0 + V] = vtail[r];
            }
            if constexpr (!ZEROU) {
                if (lo_grp || hi_grp) su[(p + 1) & 1][hs][V + x0 + V] = hter_n;
            }
        }
        // ---- first sweep on plane p, both rows
        vec v[R];
        {
            const int sc = p & 1;
            const bool zbp = (gzo + p == 0) || (gzo + p == gzn - 1);
#pragma unroll
            for (int r = 0; r < R; r++) {
                T xm = 0, xp = 0;
                vec yo = (vec)(0);  // the y-neighbour that is not this thread's other row
                if constexpr (!ZEROU) {
                    xm = su[sc][1 + i0 + r][V + x0 - 1];
                    xp = su[sc][1 + i0 + r][V + x0 + V];
                    yo = *(const vec *)&su[sc][(r == 0) ? i0 : i0 + 3][V + x0];
                }
                const vec ym = (r == 0) ? yo : uc[0];
                const vec yp = (r == 0) ? uc[1] : yo;
                v[r] = update(um[r], uc[r], up[r], ym, yp, xm, xp, b[r], zbp || ybnd[r]);
                if (NORM && outrow[r] && p >= z0 && p < z1) {   // every point of the piece belongs to exactly one tile's output rows
                    nsq += resid_sq(um[r], uc[r], up[r], ym, yp, xm, xp, b[r], zbp || ybnd[r]);
                    if (tail) {   // Dirichlet column nx-1: r = rhs - 1 * u
                        const T rt = vtail[r] - (T)1 * xp;
                        nsq += (double)rt * (double)rt;
                    }
                }
                if (RB) {
#pragma unroll
                    for (int e = 0; e < V; e++)
                        if (((x0 + e + Y0 + i0 + r + gzo + p) & 1) != 0) v[r][e] = uc[r][e];  // not red: unchanged
                }
                *(vec *)&sv[sc][i0 + r][V + x0] = v[r];
            }
        }
        // ---- the next step's operands are requested HERE, between the two sweeps: every variant measured faster with them here than
        // at the top of the step (same-box A/B of tools/pairbench at 513^3 fp64: plain pair 0.670 -> 0.641 ms, folding pair
        // 0.753 -> 0.736, red-black sweep 0.650 -> 0.630, 64-plane slab piece 0.095 -> 0.089) -- the first sweep's arithmetic
        // runs with the registers of the prefetch still free, and the requests overlap the second sweep and the barrier
        // (the right-hand side's rows a little later still, after the second sweep, where no coarse rows are in flight and u is read at
        // all: plain pair and red-black sweep -2 %; the folding variant +5 % and the zero-guess pair +17 % with them there)
        constexpr bool LATE_B = !CORR && !ZEROU;
        fetch_u(p + 2);
        if (!LATE_B) fetch_b(p + 1);
        // ---- second sweep on plane q = p-1, output rows
        const int q = p - 1;
        if (q >= z0 && q < z1) {
            const bool zbq = (gzo + q == 0) || (gzo + q == gzn - 1);
            const int sl = q & 1;
            const long long qo = (long long)q * g.plane;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (outrow[r]) {
                    const int i = i0 + r;
                    const T xm = sv[sl][i][V + x0 - 1], xp = sv[sl][i][V + x0 + V];
                    const vec yo = *(const vec *)&sv[sl][(r == 0) ? max(i - 1, 0) : i + 1][V + x0];
                    const vec ym = (r == 0) ? yo : vc[0];
                    const vec yp = (r == 0) ? vc[1] : yo;
                    vec res = update(vm[r], vc[r], v[r], ym, yp, xm, xp, bq[r], zbq || ybnd[r]);
                    if (RB) {
#pragma unroll
                        for (int e = 0; e < V; e++)
                            if (((x0 + e + Y0 + i + gzo + q) & 1) == 0) res[e] = vc[r][e];  // not black: unchanged
                    }
                    __builtin_nontemporal_store(res, (vec *)((out + (qo + urow[r])) + x0));
                    constexpr int LINE = 128 / (int)sizeof(T), TLN = LINE / V;  // lanes that write the tail line
                    if (tailwave && lane >= 64 - TLN) {
                        // column nx-1 (Dirichlet) as one full 128-byte line: value + zero padding
                        const int j = lane - (64 - TLN);
                        const int xs = g.nx - 1 + V * j;
                        const int line_end = ((g.nx - 1) / LINE + 1) * LINE;
                        if (xs < line_end) {
                            const long long rb0 = qo + urow[r];
                            vec tv = (vec)(0);
                            if (j == 0) tv[0] = sv[sl][i][V + TPR * V];   // rhs(q, row, nx-1): the first sweep's value on the Dirichlet column
                            __builtin_nontemporal_store(tv, (vec *)(out + rb0 + xs));
                        }
                    }
                }
            }
        }
        if (LATE_B) fetch_b(p + 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) { um[r] = uc[r]; uc[r] = up[r]; vm[r] = vc[r]; vc[r] = v[r]; bq[r] = b[r]; }
    }
    }   // next chunk of this workgroup's range
    if (NORM) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nsq += __shfl_down(nsq, off, 64);
        if (lane == 0) snorm[t >> 6] = nsq;
        __syncthreads();
        if (t == 0) {
            double tot = 0.;
            for (int w = 0; w < TPR * G / 64; w++) tot += snorm[w];
            partials[blockIdx.x] = tot;
        }
    }
}

// How a launch is dealt to the CUs (one workgroup each: 148 KB of LDS): `grid` workgroups (a multiple of 8 for the XCD-aware
// order) and either RANGES (zc = 0: tile-planes cut into `grid` equal runs -- every CU the same number of plane steps, two extra
// first-sweep planes only where a run starts) or z-CHUNKS of zc planes dealt round-robin (y-neighbouring tiles march the same
// planes at the same time and share their halo rows in L2, but the last round of chunks is partly empty). Ranges keep the
// lockstep only when a run is a whole fraction of a tile's column (513^3 on 256 CUs: 86 tiles x 513 planes / 256 = 172.3 planes
// = a third of a column; 1025^3 fp32: 685 planes = two thirds of one: 3.46 ms as ranges against 3.07 as chunks).
struct WidePlan { int grid, zc; };
static WidePlan wide_plan(const Geom &g, int nby, int ncopy)
{
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        const char *e = getenv("MG_PW_GRID");
        return std::max(8, ((e ? atoi(e) : n) / 8) * 8);
    }();
    static const int mode = [] { const char *e = getenv("MG_PW_MODE"); return e ? atoi(e) : -1; }();   // 0 ranges, 1 chunks, -1 auto
    static const int zc_env = [] { const char *e = getenv("MG_PW_ZC"); return e ? atoi(e) : 0; }();
    const long long total = (long long)ncopy * nby * g.nz;
    const int grid = (int)std::max<long long>(8, (std::min<long long>(ncu, total / 3) / 8) * 8);
    const double run = (double)total / grid, k = std::max(1.0, std::floor(g.nz / run + 0.5));
    const bool aligned = std::fabs(k * run - g.nz) <= std::max(2.0, 0.012 * g.nz);
    // measured on one box, same process (tools/ab_modes.sh, 513^3 fp64, ms per launch ranges / chunks): plain pair 0.661 / 0.654,
    // folding pair 0.723 / 0.734, whole cycle 2.43 / 2.39; a slab piece alone on the chip (64 planes: 86 tiles x 2 chunks fill two
    // thirds of it, x 3 one workgroup more than it): 0.111 ms as chunks, 0.089 as ranges. Chunks are nevertheless the default
    // everywhere: a grid of exactly one workgroup per CU, each holding its CU for the whole launch, takes TWICE as long as soon
    // as anything else holds a CU -- RCCL's send / recv kernels, the boundary launch on the communication stream -- while
    // one-chunk workgroups are dispatched to whatever CUs are free. Ranges: MG_PW_MODE=0 (measurements on an otherwise idle GPU).
    (void)aligned;
    if (mode == 0) return {grid, 0};
    // chunks: the count whose last round is fullest; a chunk of zc planes costs zc + 2 plane steps + the prologue
    int best_zc = std::max(1, g.nz);
    double best = 1e30;
    for (int kk = 1; kk <= std::max(1, g.nz / 4); kk++) {
        const int zc = (g.nz + kk - 1) / kk, nbz = (g.nz + zc - 1) / zc;
        const double rounds = std::ceil((double)ncopy * nby * nbz / grid);
        const double cost = std::max(rounds, 1.0) * (zc + 3.5);
        if (cost < best - 1e-9) { best = cost; best_zc = zc; }
    }
    if (zc_env > 0) best_zc = zc_env;
    // one workgroup per chunk (the kernel's loop then runs once): the hardware dispatcher deals them
    const long long items = (long long)ncopy * nby * ((g.nz + best_zc - 1) / best_zc);
    return {(int)(((items + 7) / 8) * 8), best_zc};
}

int g_wide_mode = -1;

}  // namespace

void set_pair_wide(int mode) { g_wide_mode = mode; }

// wide tiles need whole rows of 128 or 256 lanes (n = 257, 513 in fp64; 513, 1025 in fp32) and enough rows and planes
// to fill 256 CUs with 1024-thread workgroups; everything else stays with k_jacobi2
template <typename T>
bool pair_wide_ok(const Geom &g)
{
    constexpr int V = WV<T>::V;
    static const bool enabled = [] { const char *e = getenv("MG_PAIR_WIDE"); return !(e && e[0] == '0'); }();
    if (g_wide_mode == 0 || (g_wide_mode < 0 && !enabled) || g.dim != 3 || (g.nx - 1) % V != 0) return false;
    const int tpr = (g.nx - 1) / V;
    if (tpr != 128 && tpr != 256) return false;
    static const int min_rows = [] { const char *e = getenv("MG_PW_MIN_NY"); return e ? atoi(e) : 200; }();
    return g.ny >= min_rows && g.nz >= 8;   // (thin pieces -- the boundary planes of a slab -- stay with k_jacobi2: 17 against 28 us)
}

template <typename T>
int launch_pair_wide(hipStream_t s, const Geom &g, const Geom &gc, const Coef<T> &c, T omega, const T *u, const T *coarse,
                     const T *rhs, T *out, bool zero_u, bool rb, int dup, double *d_partials)
{
    constexpr int V = WV<T>::V;
    const int tpr = (g.nx - 1) / V;
    const int G = 1024 / tpr, S = 2 * G - 2;
    const int nby = (g.ny - 1 + S - 1) / S;
    const int ncopy = dup > 0 ? 2 : 1;
    const WidePlan plan = wide_plan(g, nby, ncopy);
    const int grid = plan.grid, zc = plan.zc;
    const bool damped = (omega != (T)1) && !rb;
#define MG_PW(TPR, GG, D, C, Z, RBB) \
    hipLaunchKernelGGL((k_pairw<T, TPR, GG, D, C, Z, RBB>), dim3(grid), dim3(TPR * GG), 0, s, g, c, omega, u, rhs, out, nby, zc, coarse, gc, dup, (double *)nullptr)
#define MG_PWN(TPR, GG, D) \
    hipLaunchKernelGGL((k_pairw<T, TPR, GG, D, false, false, false, true>), dim3(grid), dim3(TPR * GG), 0, s, g, c, omega, u, rhs, out, nby, zc, coarse, gc, dup, d_partials)
#define MG_PWNRB(TPR, GG) \
    hipLaunchKernelGGL((k_pairw<T, TPR, GG, false, false, false, true, true>), dim3(grid), dim3(TPR * GG), 0, s, g, c, omega, u, rhs, out, nby, zc, coarse, gc, dup, d_partials)
    const bool norm = d_partials && !coarse && !zero_u;
#define MG_PW_SHAPE(TPR, GG) \
    do { \
        if (norm) { if (rb) MG_PWNRB(TPR, GG); else if (damped) MG_PWN(TPR, GG, true); else MG_PWN(TPR, GG, false); } \
        else if (rb) { if (coarse) MG_PW(TPR, GG, false, true, false, true); else if (zero_u) MG_PW(TPR, GG, false, false, true, true); else MG_PW(TPR, GG, false, false, false, true); } \
        else if (coarse) { if (damped) MG_PW(TPR, GG, true, true, false, false); else MG_PW(TPR, GG, false, true, false, false); } \
        else if (zero_u) { if (damped) MG_PW(TPR, GG, true, false, true, false); else MG_PW(TPR, GG, false, false, true, false); } \
        else { if (damped) MG_PW(TPR, GG, true, false, false, false); else MG_PW(TPR, GG, false, false, false, false); } \
    } while (0)
    (void)G;
    if (tpr == 256) MG_PW_SHAPE(256, 4);
    else MG_PW_SHAPE(128, 8);
#undef MG_PW_SHAPE
#undef MG_PW
#undef MG_PWN
#undef MG_PWNRB
    return norm ? grid : 0;   // partial sums written (one per workgroup launched)
}

template bool pair_wide_ok<double>(const Geom &);
template bool pair_wide_ok<float>(const Geom &);
template int launch_pair_wide<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, double, const double *, const double *, const double *, double *, bool, bool, int, double *);
template int launch_pair_wide<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, float, const float *, const float *, const float *, float *, bool, bool, int, double *);

}  // namespace mg
