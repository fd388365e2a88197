// mg_rr_wide.hip -- fused residual + full-weighting restriction on HBM-resident levels, wide-tile form (round 3), gfx950.
//
//     rhs_c(K,J,I) = sum_{dz,dy,dx} w r(2K+dz, 2J+dy, 2I+dx),   r = rhs_f - A u  computed on the fly
//
// (Residual::apply_iteration_to_vec, include/solvers.hpp:257-276, followed by the V-cycle extension's full weighting; the
// reference itself restricts by aliasing, include/multigrid.hpp:113,121.) Same arithmetic, weights applied in the oracle's
// order -- x, then y, then z through a three-stage register pipeline -- as k_resid_restrict_fw (mg_transfer_fast.hip), so the
// result is bit-identical to it and to k_restrict_fw(k_residual(u)); coarse boundary nodes inject r(2K,2J,2I).
//
// Why a second form. k_resid_restrict_fw gives one coarse row to a workgroup: three fine residual rows per two new ones
// (1.5 x), five rows of u read per two, and its fabric-side read traffic was 1.3 x the bytes it must move at 513^3. Here the
// workgroup is the TILE of mg_pair_wide.hip: 1024 threads = G groups of row-wide wave teams, every thread owns an even fine
// row and the odd one above it (= one coarse row), groups exchange through LDS:
//   * u(z+1) -> LDS one step ahead (x / y neighbours of the residual stencil), z neighbours in registers, as in k_pairw;
//   * the residual rows of plane z go to LDS; the NEXT step weights them: x-weights of the thread's two rows and of the odd
//     row below (the neighbouring group's: its three x-weights are recomputed, not exchanged), then y, then the z pipeline;
//     one barrier per plane;
//   * residual on 2G rows for 2G - 2 coarse-row pairs (8 / 6); u rows requested one plane ahead, rhs rows two planes ahead
//     (two register sets alternating with the step's parity: the loop body is instantiated once per parity).
// Geometry conventions (slab pieces, ghost planes, the second single coarse plane `dup_kc` further up) are k_resid_restrict_fw's.
#include "mg_kernels.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdlib>

namespace mg {
namespace {

template <typename T> struct RV;
template <> struct RV<double> { static constexpr int V = 2; };
template <> struct RV<float> { static constexpr int V = 4; };

// SEMI: semi-coarsening transition (z kept): every fine plane is a coarse plane, 9-point weights per plane
template <typename T, int TPR, int G, bool SEMI>
__global__ __launch_bounds__(TPR * G) void k_rrw(Geom gf, Geom gc, Coef<T> c, const T *__restrict__ u_,
                                                 const T *__restrict__ rhs_, T *__restrict__ coarse_, int nby,
                                                 int zcc, int dup_kc, int dup_nzf)
{
    constexpr int V = RV<T>::V, CV = V / 2;
    constexpr int R = 2, NROW = G * R, S = NROW - 2;
    constexpr int LP = TPR * V + 2 * V;  // LDS row: V pad | TPR*V values | tail column | pad
    typedef T vec __attribute__((ext_vector_type(V)));
    __shared__ __align__(16) T su[2][NROW + 2][LP];  // u planes z (read) / z+1 (written); rows 0 and NROW+1: halo rows
    __shared__ __align__(16) T sr[2][NROW][LP];      // residual planes z-1 (read) / z (written)

    // work = (copies x) y-tiles x coarse planes, cut into gridDim.x equal ranges of consecutive coarse planes of consecutive
    // tiles (see k_pairw): one workgroup per CU, every CU the same number of plane steps
    const int t = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(t / TPR);
    const int xt = t - grp * TPR;
    const int x0 = V * xt, ic0 = CV * xt;        // first fine x / first coarse column; the gate guarantees nx - 1 == TPR * V
    const bool tail = (xt == TPR - 1);           // also owns the odd last fine column = the last coarse column
    const int i0 = grp * R;
    const bool lo_grp = (grp == 0), hi_grp = (grp == G - 1);
    const T q = (T)0.25, h = (T)0.5;
    const long long per_copy = (long long)nby * gc.nz, total = (dup_kc > 0 ? 2 : 1) * per_copy;
    const int nwg = (int)gridDim.x, wper = nwg >> 3;             // the launcher makes the grid a multiple of 8
    const int wi = (blockIdx.x & 7) * wper + (blockIdx.x >> 3);  // XCD-aware order
    // zcc == 0: ranges. zcc > 0: chunks of zcc coarse planes dealt round-robin with the tile running fastest (k_pairw explains)
    const int nbz = zcc > 0 ? (gc.nz + zcc - 1) / zcc : 0;
    const long long items = (dup_kc > 0 ? 2 : 1) * (long long)nby * nbz;
    long long w0 = zcc > 0 ? wi : total * wi / nwg;
    const long long w1 = zcc > 0 ? items : total * (wi + 1) / nwg;
    const int fgz0_in = gf.gz0, fnz_in = gf.nz, cgz0_in = gc.gz0;
    while (w0 < w1) {
    bool second;
    int by, K0, K1;
    if (zcc > 0) {
        const long long per = (long long)nby * nbz;
        second = w0 >= per;
        const long long wr = w0 - (second ? per : 0);
        const int bz = (int)(wr / nby);
        by = (int)(wr - (long long)bz * nby);
        K0 = bz * zcc; K1 = min(K0 + zcc, gc.nz);
        w0 += nwg;
    } else {
        second = w0 >= per_copy;
        const long long wr = w0 - (second ? per_copy : 0);
        by = (int)(wr / gc.nz);
        K0 = (int)(wr - (long long)by * gc.nz); K1 = (int)min((long long)gc.nz, K0 + (w1 - w0));
        w0 += K1 - K0;
    }
    // the second copy (dup_kc > 0): a single coarse plane dup_kc coarse planes further up, with dup_nzf fine planes
    gf.gz0 = fgz0_in + (second ? 2 * dup_kc : 0); gf.nz = second ? dup_nzf : fnz_in; gc.gz0 = cgz0_in + (second ? dup_kc : 0);
    const long long foff = second ? (long long)2 * dup_kc * gf.plane : 0;
    const T *__restrict__ u = u_ + foff;
    const T *__restrict__ rhs = rhs_ + foff;
    T *__restrict__ coarse = coarse_ + (second ? (long long)dup_kc * gc.plane : 0);
    const int Y0 = by * S;
    const int J = (Y0 + i0) >> 1;                // this thread's coarse row: fine rows 2J-1 (group below), 2J, 2J+1
    const bool emit_row = (grp >= 1 || by == 0) && J < gc.ny;

    long long urow[R];
    bool ybnd[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int yc = min(Y0 + i0 + r, gf.ny - 1);
        ybnd[r] = (yc == 0) || (yc == gf.ny - 1);
        urow[r] = (long long)yc * gf.pitch;
    }
    const int hy = lo_grp ? max(Y0 - 1, 0) : min(Y0 + NROW, gf.ny - 1);
    const long long hrow = (long long)hy * gf.pitch;
    const int hs = lo_grp ? 0 : NROW + 1;

    // fine planes to evaluate (all inside the GLOBAL grid): 2K0-1 .. 2(K1-1)+1, or K0 .. K1-1 when z is kept. On a z-slab the
    // first one can be the lower ghost plane -1, whose residual needs u on plane -2 (two ghost planes of u, one of rhs).
    const int zs = SEMI ? K0 : max(2 * K0 - 1, -gf.gz0), ze = SEMI ? K1 - 1 : min(2 * (K1 - 1) + 1, gf.nz - 1);
    auto uplane = [&](int p) { return (long long)min(p, ze + 1) * gf.plane; };   // planes past ze+1 are never used
    auto bplane = [&](int p) { return (long long)min(p, ze) * gf.plane; };

    vec um[R], uc[R], up[R];
    // The next step's u rows are requested one step ahead (at the top of the step), the right-hand side's rows TWO steps ahead
    // (after the residual, into the register set the step has just consumed: the two sets alternate with the step's parity, so
    // no register is moved while its load is in flight). A build without the arithmetic (loads, LDS publish and barrier only)
    // takes 0.42 ms at 513^3 -- the read ceiling of the chip -- against 0.527 with it: the arithmetic of a step is not hidden
    // behind one step's worth of requests; with the rhs rows two deep 0.44-0.49 (64-plane slab piece 0.072 -> 0.063).
    // (Everything two steps ahead through register moves had measured slower, 0.60 against 0.55 ms.)
    vec nu[R], nb[2][R], nh = (vec)(0);   // rhs: two register sets, requested TWO steps ahead (the sets alternate with the step's parity)
    T nter[R], nbt[2][R], nhter = 0;
    auto fetch_u = [&](int pu1) {
        const long long pn = uplane(pu1);
#pragma unroll
        for (int r = 0; r < R; r++) nu[r] = *(const vec *)((u + (pn + urow[r])) + x0);
        if (lo_grp || hi_grp) nh = *(const vec *)((u + (pn + hrow)) + x0);
#pragma unroll
        for (int r = 0; r < R; r++) nter[r] = 0;
        if (tail) {
#pragma unroll
            for (int r = 0; r < R; r++) nter[r] = (u + (pn + urow[r]))[x0 + V];
            if (lo_grp || hi_grp) nhter = (u + (pn + hrow))[x0 + V];
        }
    };
    auto fetch_b = [&](int pb, auto PAR) {
        constexpr int P = decltype(PAR)::value;
        const long long po = bplane(pb);
#pragma unroll
        for (int r = 0; r < R; r++) nb[P][r] = *(const vec *)((rhs + (po + urow[r])) + x0);
#pragma unroll
        for (int r = 0; r < R; r++) nbt[P][r] = 0;
        if (tail) {
#pragma unroll
            for (int r = 0; r < R; r++) nbt[P][r] = (rhs + (po + urow[r]))[x0 + V];
        }
    };
    // ---- prologue: u planes zs-1 (registers) and zs (registers + LDS)
    {
        T ter[R];
        vec hh = (vec)(0);
        T hter = 0;
        const long long pm = (long long)(zs - 1) * gf.plane, pc = (long long)zs * gf.plane;
#pragma unroll
        for (int r = 0; r < R; r++) {
            um[r] = *(const vec *)((u + (pm + urow[r])) + x0);
            uc[r] = *(const vec *)((u + (pc + urow[r])) + x0);
            ter[r] = 0;
            if (tail) ter[r] = (u + (pc + urow[r]))[x0 + V];
        }
        if (lo_grp || hi_grp) {
            hh = *(const vec *)((u + (pc + hrow)) + x0);
            if (tail) hter = (u + (pc + hrow))[x0 + V];
        }
        const int sl = zs & 1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            *(vec *)&su[sl][1 + i0 + r][V + x0] = uc[r];
            if (tail) su[sl][1 + i0 + r][V + x0 + V] = ter[r];
            if (xt == 0) {
                su[0][1 + i0 + r][V - 1] = 0; su[1][1 + i0 + r][V - 1] = 0;
                sr[0][i0 + r][V - 1] = 0; sr[1][i0 + r][V - 1] = 0;
            }
        }
        if (lo_grp || hi_grp) {
            *(vec *)&su[sl][hs][V + x0] = hh;
            if (tail) su[sl][hs][V + x0 + V] = hter;
            if (xt == 0) { su[0][hs][V - 1] = 0; su[1][hs][V - 1] = 0; }
        }
    }
    fetch_u(zs + 1);
    fetch_b(zs, std::integral_constant<int, 0>{});
    fetch_b(zs + 1, std::integral_constant<int, 1>{});
    __syncthreads();

    T ywm[CV], ywc[CV], ctr[CV], ctr_tail = 0;
#pragma unroll
    for (int m = 0; m < CV; m++) { ywm[m] = 0; ywc[m] = 0; ctr[m] = 0; }

    auto step = [&](int z, auto PAR) {
        constexpr int P = decltype(PAR)::value;
        // ---- this step's u rows were requested one step ago, its rhs rows two steps ago
        vec b[R], hn = nh;
        T ter_n[R], bt[R], hter_n = nhter;
#pragma unroll
        for (int r = 0; r < R; r++) { up[r] = nu[r]; b[r] = nb[P][r]; ter_n[r] = nter[r]; bt[r] = nbt[P][r]; }
        fetch_u(z + 2);
        // ---- u(z+1) -> LDS slot (z+1)&1
        {
            const int sn = (z + 1) & 1;
#pragma unroll
            for (int r = 0; r < R; r++) *(vec *)&su[sn][1 + i0 + r][V + x0] = up[r];
            if (lo_grp || hi_grp) *(vec *)&su[sn][hs][V + x0] = hn;
            if (tail) {
#pragma unroll
                for (int r = 0; r < R; r++) su[sn][1 + i0 + r][V + x0 + V] = ter_n[r];
                if (lo_grp || hi_grp) su[sn][hs][V + x0 + V] = hter_n;
            }
        }
        // ---- residual of plane z, both rows -> LDS
        if (z <= ze) {
            const int sc = z & 1;
            const int gzf = gf.gz0 + z;
            const bool zb = (gzf == 0) || (gzf == gf.gnz - 1);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const T xm = su[sc][1 + i0 + r][V + x0 - 1], xp = su[sc][1 + i0 + r][V + x0 + V];
                const vec yo = *(const vec *)&su[sc][(r == 0) ? i0 : i0 + 3][V + x0];
                const vec ym = (r == 0) ? yo : uc[0];
                const vec yp = (r == 0) ? uc[1] : yo;
                const bool rb = zb || ybnd[r];
                vec res;
#pragma unroll
                for (int e = 0; e < V; e++) {
                    const T left = (e == 0) ? xm : uc[r][e > 0 ? e - 1 : 0];
                    const T right = (e == V - 1) ? xp : uc[r][e < V - 1 ? e + 1 : 0];
                    T sum = 0;
                    sum += c.cz * um[r][e];
                    sum += c.cy * ym[e];
                    sum += c.cx * left;
                    sum += c.cd * uc[r][e];
                    sum += c.cx * right;
                    sum += c.cy * yp[e];
                    sum += c.cz * up[r][e];
                    if (rb || (x0 + e == 0)) sum = (T)1 * uc[r][e];
                    res[e] = b[r][e] - sum;
                }
                *(vec *)&sr[sc][i0 + r][V + x0] = res;
                if (tail) sr[sc][i0 + r][V + x0 + V] = bt[r] - (T)1 * xp;   // odd last fine column (Dirichlet): r = rhs - u
            }
        }
        // (the right-hand side's rows are requested here, after the residual, not with u at the top of the step: whole level 0.540 ->
        // 0.526 ms, 64-plane slab piece 0.078 -> 0.072 in a same-box A/B; after the weights instead 0.536, u after the publish 0.534)
        fetch_b(z + 2, PAR);
        // ---- weights of plane zw = z-1 (its residual rows were published before the last barrier)
        const int zw = z - 1;
        if (zw >= zs && emit_row) {
            const int sl = zw & 1;
            // x-weights of fine rows 2J-1, 2J, 2J+1 = tile rows i0-1, i0, i0+1 (row -1 only for the grid's first row: unused there)
            T xw[3][CV];
            vec rc = (vec)(0);   // residual of the centre row 2J
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int row = max(i0 - 1 + k, 0);
                const vec rr = *(const vec *)&sr[sl][row][V + x0];
                const T rl = sr[sl][row][V + x0 - 1];
                if (k == 1) rc = rr;
#pragma unroll
                for (int m = 0; m < CV; m++) {
                    const T rleft = (m == 0) ? rl : rr[2 * m - 1 > 0 ? 2 * m - 1 : 0];
                    xw[k][m] = q * rleft + h * rr[2 * m] + q * rr[2 * m + 1];
                }
            }
            T yw[CV];
#pragma unroll
            for (int m = 0; m < CV; m++) yw[m] = q * xw[0][m] + h * xw[1][m] + q * xw[2][m];
            const int gzf = gf.gz0 + zw;
            int emitK = -1;
            if (SEMI) emitK = zw;                              // planes map one to one
            else if (zw & 1) emitK = (zw - 1) >> 1;            // zw = 2K+1 closes coarse plane K
            else if (gzf == gf.gnz - 1) emitK = zw >> 1;       // top boundary plane of the grid has no z+1: it injects
            const bool centre = SEMI || !(zw & 1);
            if (centre) {                                      // zw = 2K (or any plane when z is kept): centre plane
#pragma unroll
                for (int m = 0; m < CV; m++) { ywc[m] = yw[m]; ctr[m] = rc[2 * m]; }
                if (tail) ctr_tail = sr[sl][i0][V + x0 + V];
            }
            if (emitK >= K0 && emitK < K1) {
                const bool Kbnd = (gc.gz0 + emitK == 0) || (gc.gz0 + emitK == gc.gnz - 1);
                const bool Jbnd = (J == 0) || (J == gc.ny - 1);
                const long long co = (long long)emitK * gc.plane + (long long)J * gc.pitch;
#pragma unroll
                for (int m = 0; m < CV; m++) {
                    const int I = ic0 + m;
                    const T fw = SEMI ? yw[m] : q * ywm[m] + h * ywc[m] + q * yw[m];
                    coarse[co + I] = (Kbnd || Jbnd || I == 0) ? ctr[m] : fw;
                }
                if (tail) coarse[co + gc.nx - 1] = ctr_tail;   // coarse column nc-1 is a boundary node
            }
            if (!SEMI && (zw & 1)) {
#pragma unroll
                for (int m = 0; m < CV; m++) ywm[m] = yw[m];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) { um[r] = uc[r]; uc[r] = up[r]; }
    };
    for (int z = zs; z <= ze + 1; z += 2) {
        step(z, std::integral_constant<int, 0>{});
        if (z + 1 <= ze + 1) step(z + 1, std::integral_constant<int, 1>{});
    }
    }   // next chunk of this workgroup's range
}

// ranges or chunks (mg_pair_wide.hip: wide_plan), in units of coarse planes (two fine planes each unless z is kept)
struct RRPlan { int grid, zcc; };
static RRPlan rr_wide_plan(const Geom &gc, int nby, int ncopy, bool semi)
{
    const bool piece = gc.gnz != gc.nz;   // a z-slab piece
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        const char *e = getenv("MG_RRW_GRID");
        return std::max(8, ((e ? atoi(e) : n) / 8) * 8);
    }();
    static const int mode = [] { const char *e = getenv("MG_RRW_MODE"); return e ? atoi(e) : -1; }();
    static const int zcc_env = [] { const char *e = getenv("MG_RRW_ZCC"); return e ? atoi(e) : 0; }();
    const long long total = (long long)ncopy * nby * gc.nz;
    const int grid = (int)std::max<long long>(8, (std::min<long long>(ncu, total / 2) / 8) * 8);
    const double run = (double)total / grid, k = std::max(1.0, std::floor(gc.nz / run + 0.5));
    const bool aligned = std::fabs(k * run - gc.nz) <= std::max(1.0, 0.012 * gc.nz);
    (void)aligned; (void)piece;
    if (mode == 0) return {grid, 0};   // ranges only on request (mg_pair_wide.hip: wide_plan says why); whole level 0.543 against 0.525 ms as chunks
    int best_zcc = std::max(1, gc.nz);
    double best = 1e30;
    for (int kk = 1; kk <= gc.nz; kk++) {
        const int zcc = (gc.nz + kk - 1) / kk, nbz = (gc.nz + zcc - 1) / zcc;
        const double rounds = std::ceil((double)ncopy * nby * nbz / grid);
        const double cost = std::max(rounds, 1.0) * ((semi ? 1 : 2) * zcc + 4.5);
        if (cost < best - 1e-9) { best = cost; best_zcc = zcc; }
    }
    if (zcc_env > 0) best_zcc = zcc_env;
    const long long items = (long long)ncopy * nby * ((gc.nz + best_zcc - 1) / best_zcc);   // one workgroup per chunk
    return {(int)(((items + 7) / 8) * 8), best_zcc};
}

int g_rr_wide_mode = -1;

}  // namespace

void set_rr_wide(int mode) { g_rr_wide_mode = mode; }

template <typename T>
bool rr_wide_ok(const Geom &gf, const Geom &gc)
{
    constexpr int V = RV<T>::V;
    static const bool enabled = [] { const char *e = getenv("MG_RR_WIDE"); return !(e && e[0] == '0'); }();
    if (g_rr_wide_mode == 0 || (g_rr_wide_mode < 0 && !enabled) || gf.dim != 3 || (gf.nx - 1) % V != 0 || gf.nx != 2 * gc.nx - 1 || gf.ny != 2 * gc.ny - 1) return false;
    const int tpr = (gf.nx - 1) / V;
    return (tpr == 128 || tpr == 256) && gf.ny >= 200 && gc.nz >= 4;   // (single coarse planes -- a slab's boundary pieces -- stay with k_resid_restrict_fw)
}

template <typename T>
void launch_rr_wide(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, const T *u, const T *rhs, T *coarse,
                    int dup_kc, int dup_nzf)
{
    constexpr int V = RV<T>::V;
    const int tpr = (gf.nx - 1) / V;
    const int G = 1024 / tpr, S = 2 * G - 2;
    const int nby = (gf.ny - 1 + S - 1) / S;
    const bool semi = gf.gnz == gc.gnz && gf.gnz > 1;
    if (gc.nz != 1) dup_kc = 0;
    const int ncopy = dup_kc > 0 ? 2 : 1;
    const RRPlan plan = rr_wide_plan(gc, nby, ncopy, semi);
    const int grid = plan.grid, zcc = plan.zcc;
#define MG_RRW(TPR, GG, SEMI) \
    hipLaunchKernelGGL((k_rrw<T, TPR, GG, SEMI>), dim3(grid), dim3(TPR * GG), 0, s, gf, gc, c, u, rhs, coarse, nby, zcc, dup_kc, dup_nzf)
    if (tpr == 256) { if (semi) MG_RRW(256, 4, true); else MG_RRW(256, 4, false); }
    else { if (semi) MG_RRW(128, 8, true); else MG_RRW(128, 8, false); }
#undef MG_RRW
}

template bool rr_wide_ok<double>(const Geom &, const Geom &);
template bool rr_wide_ok<float>(const Geom &, const Geom &);
template void launch_rr_wide<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, const double *, const double *, double *, int, int);
template void launch_rr_wide<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, const float *, const float *, float *, int, int);

}  // namespace mg
