// tools/kbench.hip -- design-space microbenchmark for the finest-grid Jacobi sweep.
// Standalone (hipcc --offload-arch=gfx950 tools/kbench.hip -o tools/kbench): allocates a
// 3-D level in the library's HBM layout, runs every variant, checks it BIT FOR BIT
// against a plain one-point-per-thread kernel and prints ms and algorithmic GB/s
// (24 B per grid point). The winner is what mg_jacobi_fast.hip ships.
//
// usage: kbench [n=513] [reps=20] [dtype f64]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../multigrid_prj_amd/csrc/mg_geom.h"

using namespace mg;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ reference kernel
template <bool DAMPED>
__global__ __launch_bounds__(256) void k_ref(Geom g, Coef<double> c, double omega,
                                             const double *__restrict__ u,
                                             const double *__restrict__ rhs, double *__restrict__ out)
{
    int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    long long i = (long long)z * g.plane + (long long)y * g.pitch + x;
    double b = rhs[i];
    int gz = g.gz0 + z;
    bool bnd = x == 0 || y == 0 || x == g.nx - 1 || y == g.ny - 1 || gz == 0 || gz == g.gnz - 1;
    double r = b;
    if (!bnd) {
        double sum = 0;
        sum += c.cz * u[i - g.plane];
        sum += c.cy * u[i - g.pitch];
        sum += c.cx * u[i - 1];
        sum += c.cx * u[i + 1];
        sum += c.cy * u[i + g.pitch];
        sum += c.cz * u[i + g.plane];
        double jac = (b - sum) / c.cd;
        r = DAMPED ? u[i] + omega * (jac - u[i]) : jac;
    }
    out[i] = r;
}

// reference colour half-sweep (out of place): points of `colour` get the GS update, others copy
__global__ __launch_bounds__(256) void k_ref_rb(Geom g, Coef<double> c, int colour, const double *__restrict__ u,
                                                const double *__restrict__ rhs, double *__restrict__ out)
{
    int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    long long i = (long long)z * g.plane + (long long)y * g.pitch + x;
    int gz = g.gz0 + z;
    double r = u[i];
    if (((x + y + gz) & 1) == colour) {
        bool bnd = x == 0 || y == 0 || x == g.nx - 1 || y == g.ny - 1 || gz == 0 || gz == g.gnz - 1;
        r = rhs[i];
        if (!bnd) {
            double sum = 0;
            sum += c.cz * u[i - g.plane];
            sum += c.cy * u[i - g.pitch];
            sum += c.cx * u[i - 1];
            sum += c.cx * u[i + 1];
            sum += c.cy * u[i + g.pitch];
            sum += c.cz * u[i + g.plane];
            r = (rhs[i] - sum) / c.cd;
        }
    }
    out[i] = r;
}

// ------------------------------------------------------------------ traffic ceiling
template <bool NT>
__global__ __launch_bounds__(256) void k_copy3(const d2 *__restrict__ u, const d2 *__restrict__ rhs,
                                               d2 *__restrict__ out, long long n2)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
        d2 v = u[i] + rhs[i];
        if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
}

// calibration streams: U = independent 16-byte accesses in flight per stream per thread
template <int U, bool NT, int MODE>  // MODE 0: 2R+1W, 1: 1R+1W, 2: 2R (sum), 3: 1W
__global__ __launch_bounds__(256) void k_stream(const d2 *__restrict__ a, const d2 *__restrict__ b,
                                                d2 *__restrict__ out, long long n2)
{
    const long long stride = (long long)gridDim.x * 256;
    d2 acc = {0, 0};
    for (long long i0 = blockIdx.x * 256ll + threadIdx.x; i0 < n2; i0 += stride * U) {
        d2 va[U], vb[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            long long i = i0 + k * stride;
            if (i < n2) {
                if (MODE != 3) va[k] = NT ? __builtin_nontemporal_load(&a[i]) : a[i];
                if (MODE == 0 || MODE == 2) vb[k] = NT ? __builtin_nontemporal_load(&b[i]) : b[i];
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            long long i = i0 + k * stride;
            if (i < n2) {
                d2 v;
                if (MODE == 0) v = va[k] + vb[k];
                else if (MODE == 1) v = va[k];
                else if (MODE == 2) { acc += va[k] + vb[k]; continue; }
                else v = d2{1.0, 2.0};
                if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
            }
        }
    }
    if (MODE == 2 && acc.x + acc.y == 12345.678) out[0] = acc;
}

// ------------------------------------------------------------------ z-marching variants
__device__ __forceinline__ double dpp_from_prev_lane(double v, double lane0_value)
{   // lane i <- lane i-1 ; lane 0 keeps lane0_value   (DPP wave_shr:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo = __builtin_amdgcn_update_dpp(__double2loint(lane0_value), lo, 0x138, 0xf, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp(__double2hiint(lane0_value), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double dpp_from_next_lane(double v, double lane63_value)
{   // lane i <- lane i+1 ; lane 63 keeps lane63_value  (DPP wave_shl:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo = __builtin_amdgcn_update_dpp(__double2loint(lane63_value), lo, 0x130, 0xf, 0xf, false);
    int rhi = __builtin_amdgcn_update_dpp(__double2hiint(lane63_value), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(rhi, rlo);
}

// One wave = 128 consecutive x (double2 per lane) x RY rows; BW waves stacked in y;
// the workgroup marches ZC planes in z keeping (z-1, z, z+1) of its columns in
// registers, so every u value is fetched from memory once per workgroup column
// (+ y-halo rows from L1/L2 and 2 z-halo planes per ZC).
//  XMODE 0: x+-1 neighbours by (L1-resident) unaligned 8-byte loads
//  XMODE 1: x+-1 neighbours from the adjacent lanes by DPP wave shifts; only the two
//           edge lanes of the wave load
//  NT 0: plain stores; 1: non-temporal stores; 2: + non-temporal rhs loads
//  SWZ: XCD-aware block order (each XCD's L2 serves a contiguous z/y range)
template <int RY, int BW, int ZC, int XMODE, int NT, bool SWZ, bool DAMPED, int TAIL = 1>
__global__ __launch_bounds__(64 * BW) void k_zm(Geom g, Coef<double> c, double omega,
                                                const double *__restrict__ u,
                                                const double *__restrict__ rhs,
                                                double *__restrict__ out, int nbx, int nby, int nbz)
{
    int bid = blockIdx.x;
    const int nblocks = nbx * nby * nbz;
    if (SWZ) {
        int per = (nblocks + 7) >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= nblocks) return;
    }
    const int bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int npairs = g.nx >> 1;
    const int x0 = 2 * (bx * 64 + lane);
    const bool xin = x0 < 2 * npairs;
    const int x0c = min(x0, g.pitch - 2);  // clamped for loads: every lane stays active
    const int yb = (by * BW + wv) * RY;
    const int z0 = bz * ZC;
    const int zend = min(z0 + ZC, g.nz);
    const bool tail = (g.nx & 1) && (x0 + 2 == g.nx - 1);
    const bool tailwave = (g.nx & 1) && (bx * 128 <= g.nx - 3) && (g.nx - 3 < bx * 128 + 128);

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
    const bool xb0 = (x0 == 0), xb1 = (x0 + 1 == g.nx - 1);

    d2 zm[RY], cc[RY], zp[RY];
    const double *pz = u + (long long)z0 * g.plane;
#pragma unroll
    for (int r = 0; r < RY; r++) {
        zm[r] = *(const d2 *)(pz - g.plane + rowoff[r]);
        cc[r] = *(const d2 *)(pz + rowoff[r]);
    }
    for (int z = z0; z < zend; z++, pz += g.plane) {
        const long long zo = (long long)z * g.plane;
        d2 b[RY];
#pragma unroll
        for (int r = 0; r < RY; r++) {
            zp[r] = *(const d2 *)(pz + g.plane + rowoff[r]);
            if (NT == 2) b[r] = __builtin_nontemporal_load((const d2 *)(rhs + zo + rowoff[r]));
            else b[r] = *(const d2 *)(rhs + zo + rowoff[r]);
        }
        d2 hlo = *(const d2 *)(pz + off_lo);
        d2 hhi = *(const d2 *)(pz + off_hi);
        const int gz = g.gz0 + z;
        const bool zb = (gz == 0) || (gz == g.gnz - 1);
#pragma unroll
        for (int r = 0; r < RY; r++) {
            double xm, xp;
            if (XMODE == 0) {
                xm = pz[rowoff[r] - 1];
                xp = pz[rowoff[r] + 2];
            } else {
                double el = 0, er = 0;
                if (lane == 0) el = pz[rowoff[r] - 1];
                if (lane == 63) er = pz[rowoff[r] + 2];
                xm = dpp_from_prev_lane(cc[r].y, el);
                xp = dpp_from_next_lane(cc[r].x, er);
            }
            d2 ym = (r > 0) ? cc[r > 0 ? r - 1 : 0] : hlo;
            d2 yp = (r < RY - 1) ? cc[r < RY - 1 ? r + 1 : 0] : hhi;
            double s0 = 0, s1 = 0;
            s0 += c.cz * zm[r].x; s1 += c.cz * zm[r].y;
            s0 += c.cy * ym.x;    s1 += c.cy * ym.y;
            s0 += c.cx * xm;      s1 += c.cx * cc[r].x;
            s0 += c.cx * cc[r].y; s1 += c.cx * xp;
            s0 += c.cy * yp.x;    s1 += c.cy * yp.y;
            s0 += c.cz * zp[r].x; s1 += c.cz * zp[r].y;
            double j0 = (b[r].x - s0) / c.cd, j1 = (b[r].y - s1) / c.cd;
            if (DAMPED) {
                j0 = cc[r].x + omega * (j0 - cc[r].x);
                j1 = cc[r].y + omega * (j1 - cc[r].y);
            }
            const bool rb = zb || ybnd[r];
            d2 res;
            res.x = (rb || xb0) ? b[r].x : j0;
            res.y = (rb || xb1) ? b[r].y : j1;
            if (xin && yin[r]) {
                d2 *po = (d2 *)(out + zo + rowoff[r]);
                if (NT) __builtin_nontemporal_store(res, po); else *po = res;
                if (TAIL == 1 && tail) out[zo + rowoff[r] + 2] = rhs[zo + rowoff[r] + 2];
            }
            if (TAIL >= 2 && tailwave && lane >= 56 && yin[r]) {
                // last (odd) column as one full 64/128-byte line: boundary value + zero padding
                const int j = lane - 56;
                const int xs = g.nx - 1 + 2 * j;
                const int line_end = (TAIL == 2) ? ((g.nx - 1) / 8 + 1) * 8 : ((g.nx - 1) / 16 + 1) * 16;
                if (xs < line_end) {
                    const long long ro = zo + (rowoff[r] - x0c) ;
                    d2 tv = {0.0, 0.0};
                    if (j == 0) tv.x = rhs[ro + g.nx - 1];
                    d2 *pt = (d2 *)(out + ro + xs);
                    if (NT) __builtin_nontemporal_store(tv, pt); else *pt = tv;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RY; r++) { zm[r] = cc[r]; cc[r] = zp[r]; }
    }
}


// ------------------------------------------------------------------ fused double sweep
// out = J(J(u)): two damped-Jacobi sweeps in ONE pass over HBM (temporal blocking).
// Workgroup = TPR threads = one full grid row (TPR*2 doubles + the odd tail column) x TYO
// output rows, marching ZC planes. Per step p: (1) every thread computes the first sweep
// v(p) on TYO+2 rows (overlapped tiling in y) from u planes p-1,p,p+1 held in registers;
// (2) it computes the second sweep on plane q = p-1 of its TYO rows: z-neighbours v(q-1),
// v(q+1) are its own registers, x/y-neighbours of v(q) come from an LDS plane written in the
// previous step; (3) it publishes v(p) to the other LDS slot; one barrier.
template <int TPR, int TYO, int ZC, bool DAMPED, bool RB = false>
__global__ __launch_bounds__(TPR) void k_j2(Geom g, Coef<double> c, double omega, const double *__restrict__ u,
                                            const double *__restrict__ rhs, double *__restrict__ out, int nby, int nbz)
{
    constexpr int TYV = TYO + 2;
    constexpr int LP = TPR * 2 + 4;  // LDS row pitch (doubles): 2 pad | row | tail | pad
    __shared__ double lds[2][TYV][LP];
    int bid = blockIdx.x;
    const int nblocks = nby * nbz;
    {
        int per = (nblocks + 7) >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= nblocks) return;
    }
    const int by = bid % nby, bz = bid / nby;
    const int t = threadIdx.x, lane = t & 63;
    const int x0 = 2 * t;
    const bool xin = x0 + 1 < g.nx;                 // full pair inside the row
    const int x0c = min(x0, g.pitch - 2);
    const bool tail = (g.nx & 1) && (x0 + 2 == g.nx - 1);
    const int y0 = by * TYO;                        // first output row; v rows y0-1 .. y0+TYO
    const int z0 = bz * ZC, z1 = min(z0 + ZC, g.nz);
    long long ro[TYV];  bool ybnd[TYV];
#pragma unroll
    for (int r = 0; r < TYV; r++) {
        int y = min(max(y0 - 1 + r, 0), g.ny - 1);
        ybnd[r] = (y == 0) || (y == g.ny - 1);
        ro[r] = (long long)y * g.pitch + x0c;
    }
    const long long ro_lo = (long long)min(max(y0 - 2, 0), g.ny - 1) * g.pitch + x0c;
    const long long ro_hi = (long long)min(y0 + TYO + 1, g.ny - 1) * g.pitch + x0c;
    const bool xb0 = (x0 == 0), xb1 = (x0 + 1 == g.nx - 1);
    auto plane_of = [&](int p) { return (long long)min(max(p, -1), g.nz) * g.plane; };  // clamp into the allocation

    d2 um[TYV], uc[TYV], up[TYV];
    d2 vm[TYO], vc[TYO], vp[TYO];   // own-column v(q-1), v(q), v(q+1) of the output rows
    d2 bq[TYO];                     // rhs of the output rows at plane q
#pragma unroll
    for (int r = 0; r < TYV; r++) {
        um[r] = *(const d2 *)(u + plane_of(z0 - 2) + ro[r]);
        uc[r] = *(const d2 *)(u + plane_of(z0 - 1) + ro[r]);
    }
#pragma unroll
    for (int r = 0; r < TYO; r++) { vm[r] = d2{0, 0}; vc[r] = d2{0, 0}; bq[r] = d2{0, 0}; }

    for (int p = z0 - 1; p <= z1; p++) {
        const long long po = plane_of(p);
        const double *pu = u + po;
        // planes outside the domain (p = -1 or nz, only at the first/last chunk) are never
        // evaluated: their v feeds Dirichlet outputs only, and the wave-edge load of row 0 on
        // plane -1 would fall one element before the allocation
        const bool pin = (p >= 0) && (p < g.nz);
        d2 b[TYV], v[TYV];
        double vtail[TYV];
#pragma unroll
        for (int r = 0; r < TYV; r++) {
            up[r] = *(const d2 *)(u + plane_of(p + 1) + ro[r]);
            b[r] = d2{0, 0}; v[r] = d2{0, 0}; vtail[r] = 0;
        }
        if (pin) {
#pragma unroll
            for (int r = 0; r < TYV; r++) b[r] = *(const d2 *)(rhs + po + ro[r]);
            const d2 hlo = *(const d2 *)(pu + ro_lo);
            const d2 hhi = *(const d2 *)(pu + ro_hi);
            const int gzp = g.gz0 + p;
            const bool zbp = (gzp <= 0) || (gzp >= g.gnz - 1);
            // ---- (1) first sweep on plane p, TYV rows
#pragma unroll
            for (int r = 0; r < TYV; r++) {
                double el = 0, er = 0;
                if (lane == 0) el = pu[ro[r] - 1];
                if (lane == 63) er = pu[ro[r] + 2];
                const double xm = dpp_from_prev_lane(uc[r].y, el);
                const double xp = dpp_from_next_lane(uc[r].x, er);
                const d2 ym = (r > 0) ? uc[r > 0 ? r - 1 : 0] : hlo;
                const d2 yp = (r < TYV - 1) ? uc[r < TYV - 1 ? r + 1 : 0] : hhi;
                double s0 = 0, s1 = 0;
                s0 += c.cz * um[r].x; s1 += c.cz * um[r].y;
                s0 += c.cy * ym.x;    s1 += c.cy * ym.y;
                s0 += c.cx * xm;      s1 += c.cx * uc[r].x;
                s0 += c.cx * uc[r].y; s1 += c.cx * xp;
                s0 += c.cy * yp.x;    s1 += c.cy * yp.y;
                s0 += c.cz * up[r].x; s1 += c.cz * up[r].y;
                double j0 = (b[r].x - s0) / c.cd, j1 = (b[r].y - s1) / c.cd;
                if (DAMPED) { j0 = uc[r].x + omega * (j0 - uc[r].x); j1 = uc[r].y + omega * (j1 - uc[r].y); }
                const bool rb = zbp || ybnd[r];
                v[r].x = (rb || xb0) ? b[r].x : j0;
                v[r].y = (rb || xb1) ? b[r].y : j1;
                if (tail) vtail[r] = rhs[po + ro[r] + 2];   // first sweep on the Dirichlet column: v = rhs
                if (RB) {   // red half-sweep: only points with (x+y+z) even change
                    const int yy = min(max(y0 - 1 + r, 0), g.ny - 1);
                    const int par = (x0 + yy + gzp) & 1;   // parity of element .x
                    if (par != 0) v[r].x = uc[r].x;
                    if (par == 0) v[r].y = uc[r].y;
                    if (tail && ((x0 + 2 + yy + gzp) & 1) != 0) vtail[r] = pu[ro[r] + 2];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < TYO; r++) vp[r] = v[r + 1];
        // ---- (2) second sweep on plane q = p-1 of the TYO output rows
        const int q = p - 1;
        if (q >= z0 && q < z1) {
            const int gzq = g.gz0 + q;
            const bool zbq = (gzq == 0) || (gzq == g.gnz - 1);
            const int sl = q & 1;
            const long long qo = (long long)q * g.plane;
#pragma unroll
            for (int r = 0; r < TYO; r++) {
                const int y = y0 + r;
                if (y < g.ny) {
                    const int lr = r + 1;
                    const double xm = lds[sl][lr][2 + x0 - 1], xp = lds[sl][lr][2 + x0 + 2];
                    const d2 ym = *(const d2 *)&lds[sl][lr - 1][2 + x0];
                    const d2 yp = *(const d2 *)&lds[sl][lr + 1][2 + x0];
                    double s0 = 0, s1 = 0;
                    s0 += c.cz * vm[r].x; s1 += c.cz * vm[r].y;
                    s0 += c.cy * ym.x;    s1 += c.cy * ym.y;
                    s0 += c.cx * xm;      s1 += c.cx * vc[r].x;
                    s0 += c.cx * vc[r].y; s1 += c.cx * xp;
                    s0 += c.cy * yp.x;    s1 += c.cy * yp.y;
                    s0 += c.cz * vp[r].x; s1 += c.cz * vp[r].y;
                    double j0 = (bq[r].x - s0) / c.cd, j1 = (bq[r].y - s1) / c.cd;
                    if (DAMPED) { j0 = vc[r].x + omega * (j0 - vc[r].x); j1 = vc[r].y + omega * (j1 - vc[r].y); }
                    const bool rb = zbq || (y == 0) || (y == g.ny - 1);
                    d2 res;
                    res.x = (rb || xb0) ? bq[r].x : j0;
                    res.y = (rb || xb1) ? bq[r].y : j1;
                    double tailres = 0;
                    if (tail) tailres = rhs[qo + ro[lr] + 2];
                    if (RB) {   // black half-sweep: only points with (x+y+z) odd change
                        const int par = (x0 + y + gzq) & 1;
                        if (par != 1) res.x = vc[r].x;
                        if (par == 1) res.y = vc[r].y;
                        if (tail && ((x0 + 2 + y + gzq) & 1) != 1) tailres = lds[sl][lr][2 + x0 + 2];
                    }
                    if (xin) __builtin_nontemporal_store(res, (d2 *)(out + qo + ro[lr]));
                    if (tail) out[qo + ro[lr] + 2] = tailres;
                }
            }
        }
        // ---- (3) publish v(p) for the next step's x/y neighbours
        {
            const int sl = p & 1;
#pragma unroll
            for (int r = 0; r < TYV; r++) {
                if (xin) *(d2 *)&lds[sl][r][2 + x0] = v[r];
                if (tail) lds[sl][r][2 + x0 + 2] = vtail[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TYV; r++) { um[r] = uc[r]; uc[r] = up[r]; }
#pragma unroll
        for (int r = 0; r < TYO; r++) { vm[r] = vc[r]; vc[r] = vp[r]; bq[r] = b[r + 1]; }
    }
}

// ------------------------------------------------------------------ harness
__global__ void k_cmp(Geom g, const double *a, const double *b, unsigned long long *bad)
{
    int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    long long i = (long long)z * g.plane + (long long)y * g.pitch + x;
    if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(bad, 1ull);
}
__global__ void k_fill(double *p, long long n, unsigned long long seed)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        p[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

struct Ctx {
    Geom g;
    Coef<double> c;
    double omega;
    double *u, *rhs, *out, *ref;  // plane-0 pointers
    unsigned long long *bad;
    hipStream_t s;
    hipEvent_t e0, e1;
    int reps;
    double pts;
};

template <typename F>
static void run(Ctx &C, const char *name, F launch, bool check = true)
{
    size_t elems = (size_t)(C.g.nz + 2) * C.g.plane;
    CK(hipMemsetAsync(C.out - C.g.plane, 0xff, elems * 8, C.s));
    for (int i = 0; i < 2; i++) launch();
    CK(hipStreamSynchronize(C.s));
    unsigned long long bad = 0;
    if (check) {
        CK(hipMemsetAsync(C.bad, 0, 8, C.s));
        dim3 gr((C.g.nx + 63) / 64, (C.g.ny + 3) / 4, C.g.nz);
        hipLaunchKernelGGL(k_cmp, gr, dim3(64, 4), 0, C.s, C.g, C.out, C.ref, C.bad);
        CK(hipMemcpyAsync(&bad, C.bad, 8, hipMemcpyDeviceToHost, C.s));
        CK(hipStreamSynchronize(C.s));
    }
    CK(hipEventRecord(C.e0, C.s));
    for (int i = 0; i < C.reps; i++) launch();
    CK(hipEventRecord(C.e1, C.s));
    CK(hipEventSynchronize(C.e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, C.e0, C.e1));
    ms /= C.reps;
    printf("%-44s %8.4f ms  %8.1f GB/s  %5.1f%% of 8TB/s  %s\n", name, ms, C.pts * 24.0 / ms / 1e6,
           C.pts * 24.0 / ms / 1e6 / 80.0, check ? (bad ? "MISMATCH" : "ok") : "-");
    if (bad) printf("   !! %llu mismatching points\n", bad);
    fflush(stdout);
}

template <int RY, int BW, int ZC, int XMODE, int NT, bool SWZ, int TAIL = 1>
static void run_zm(Ctx &C)
{
    const Geom &g = C.g;
    int npairs = g.nx >> 1;
    int nbx = (npairs + 63) / 64, nby = (g.ny + RY * BW - 1) / (RY * BW), nbz = (g.nz + ZC - 1) / ZC;
    int nblocks = nbx * nby * nbz;
    int grid = SWZ ? ((nblocks + 7) / 8) * 8 : nblocks;
    char name[128];
    snprintf(name, sizeof name, "zm RY=%d BW=%d ZC=%-3d X=%s NT=%d SWZ=%d T=%d", RY, BW, ZC, XMODE ? "dpp" : "ld ", NT, (int)SWZ, TAIL);
    run(C, name, [&] {
        hipLaunchKernelGGL((k_zm<RY, BW, ZC, XMODE, NT, SWZ, true, TAIL>), dim3(grid), dim3(64 * BW), 0, C.s, g, C.c,
                           C.omega, C.u, C.rhs, C.out, nbx, nby, nbz);
    });
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 513;
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    Ctx C;
    C.reps = reps;
    Geom &g = C.g;
    g.dim = 3; g.nx = g.ny = g.nz = n; g.pitch = ((n + 15) / 16) * 16;
    g.plane = (long long)g.ny * g.pitch; g.gz0 = 0; g.gnz = n;
    double h = 1.0 / (n - 1), k = h * h;
    C.c = Coef<double>{-1.0 / k, -1.0 / k, -1.0 / k, 6.0 / k};
    C.omega = 6.0 / 7.0;
    C.pts = (double)n * n * n;
    size_t elems = (size_t)(g.nz + 2) * g.plane;
    CK(hipStreamCreate(&C.s));
    CK(hipEventCreate(&C.e0)); CK(hipEventCreate(&C.e1));
    double *bu, *br, *bo, *bf;
    CK(hipMalloc(&bu, elems * 8)); CK(hipMalloc(&br, elems * 8));
    CK(hipMalloc(&bo, elems * 8)); CK(hipMalloc(&bf, elems * 8));
    CK(hipMalloc(&C.bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, C.s, bu, (long long)elems, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, C.s, br, (long long)elems, 2ull);
    C.u = bu + g.plane; C.rhs = br + g.plane; C.out = bo + g.plane; C.ref = bf + g.plane;
    printf("kbench n=%d pitch=%d reps=%d  (GB/s = 24 B/pt algorithmic)\n", n, g.pitch, reps);

    dim3 gr((g.nx + 63) / 64, (g.ny + 3) / 4, g.nz);
    // reference result
    hipLaunchKernelGGL((k_ref<true>), gr, dim3(64, 4), 0, C.s, g, C.c, C.omega, C.u, C.rhs, C.ref);
    CK(hipStreamSynchronize(C.s));
    run(C, "ref: 1 pt/thread 64x4 (library generic)", [&] {
        hipLaunchKernelGGL((k_ref<true>), gr, dim3(64, 4), 0, C.s, g, C.c, C.omega, C.u, C.rhs, C.out);
    });
    long long n2 = (long long)g.nz * g.plane / 2;
    run(C, "copy3 (u+rhs->out, 24 B/pt padded) plain", [&] {
        hipLaunchKernelGGL((k_copy3<false>), dim3(256 * 16), dim3(256), 0, C.s, (const d2 *)C.u, (const d2 *)C.rhs, (d2 *)C.out, n2);
    }, false);
    run(C, "copy3 nontemporal store", [&] {
        hipLaunchKernelGGL((k_copy3<true>), dim3(256 * 16), dim3(256), 0, C.s, (const d2 *)C.u, (const d2 *)C.rhs, (d2 *)C.out, n2);
    }, false);

    if (!(argc > 3 && (std::string(argv[3]) == "zc" || std::string(argv[3]) == "tail" || std::string(argv[3]) == "pair" || std::string(argv[3]) == "j2" || std::string(argv[3]) == "rb2"))) {
        struct { int mode; double bytes; const char *nm; } modes[] = {{0, 24, "2R+1W"}, {1, 16, "1R+1W"}, {2, 16, "2R"}, {3, 8, "1W"}};
        for (auto &m : modes)
            for (int nt = 0; nt < 2; nt++)
                for (int gridmul : {8, 16, 32}) {
                    char name[128];
                    auto go = [&](auto launch, int U) {
                        snprintf(name, sizeof name, "stream %-6s U=%d nt=%d grid=256x%d", m.nm, U, nt, gridmul);
                        size_t elems2 = (size_t)(C.g.nz + 2) * C.g.plane;
                        (void)elems2;
                        for (int i = 0; i < 2; i++) launch();
                        CK(hipEventRecord(C.e0, C.s));
                        for (int i = 0; i < C.reps; i++) launch();
                        CK(hipEventRecord(C.e1, C.s));
                        CK(hipEventSynchronize(C.e1));
                        float ms = 0; CK(hipEventElapsedTime(&ms, C.e0, C.e1)); ms /= C.reps;
                        printf("%-44s %8.4f ms  %8.1f GB/s actual (%.0f B/elem)\n", name, ms, (double)n2 * 2 * m.bytes / ms / 1e6, m.bytes);
                        fflush(stdout);
                    };
#define STREAM(U, NTV, MODEV) go([&] { hipLaunchKernelGGL((k_stream<U, NTV, MODEV>), dim3(256 * gridmul), dim3(256), 0, C.s, (const d2 *)C.u, (const d2 *)C.rhs, (d2 *)C.out, n2); }, U)
#define STREAM_M(U, NTV) do { if (m.mode == 0) STREAM(U, NTV, 0); else if (m.mode == 1) STREAM(U, NTV, 1); else if (m.mode == 2) STREAM(U, NTV, 2); else STREAM(U, NTV, 3); } while (0)
                    if (nt) { STREAM_M(1, true); STREAM_M(4, true); } else { STREAM_M(1, false); STREAM_M(4, false); }
                }
    }
    if (argc > 3 && std::string(argv[3]) == "rb2") {
        double *A = nullptr;
        CK(hipMalloc(&A, elems * 8));
        CK(hipMemset(A, 0, elems * 8));
        A += g.plane;
        hipLaunchKernelGGL(k_ref_rb, gr, dim3(64, 4), 0, C.s, g, C.c, 0, C.u, C.rhs, A);
        hipLaunchKernelGGL(k_ref_rb, gr, dim3(64, 4), 0, C.s, g, C.c, 1, A, C.rhs, C.ref);
        CK(hipStreamSynchronize(C.s));
        auto go = [&](auto kern, const char *nm, int tpr, int tyo, int zc) {
            int nby = (g.ny + tyo - 1) / tyo, nbz = (g.nz + zc - 1) / zc;
            int nblocks = nby * nbz, grid = ((nblocks + 7) / 8) * 8;
            run(C, nm, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(tpr), 0, C.s, g, C.c, C.omega, C.u, C.rhs, C.out, nby, nbz); });
        };
        for (int rep = 0; rep < 2; rep++) {
            run(C, "rb unfused reference pair (1 pt/thread)", [&] {
                hipLaunchKernelGGL(k_ref_rb, gr, dim3(64, 4), 0, C.s, g, C.c, 0, C.u, C.rhs, A);
                hipLaunchKernelGGL(k_ref_rb, gr, dim3(64, 4), 0, C.s, g, C.c, 1, A, C.rhs, C.out);
            });
            go(k_j2<256, 2, 12, false, true>, "rb fused TPR=256 TYO=2 ZC=12", 256, 2, 12);
            go(k_j2<256, 2, 24, false, true>, "rb fused TPR=256 TYO=2 ZC=24", 256, 2, 24);
            go(k_j2<256, 3, 16, false, true>, "rb fused TPR=256 TYO=3 ZC=16", 256, 3, 16);
            go(k_j2<256, 4, 16, false, true>, "rb fused TPR=256 TYO=4 ZC=16", 256, 4, 16);
            go(k_j2<256, 4, 32, false, true>, "rb fused TPR=256 TYO=4 ZC=32", 256, 4, 32);
            go(k_j2<256, 6, 24, false, true>, "rb fused TPR=256 TYO=6 ZC=24", 256, 6, 24);
        }
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "j2") {
        // reference: two plain sweeps u -> A -> ref
        double *A = nullptr;
        CK(hipMalloc(&A, elems * 8));
        CK(hipMemset(A, 0, elems * 8));
        A += g.plane;
        hipLaunchKernelGGL((k_ref<true>), gr, dim3(64, 4), 0, C.s, g, C.c, C.omega, C.u, C.rhs, A);
        hipLaunchKernelGGL((k_ref<true>), gr, dim3(64, 4), 0, C.s, g, C.c, C.omega, A, C.rhs, C.ref);
        CK(hipStreamSynchronize(C.s));
        C.pts *= 2;  // two sweeps per launch: GB/s printed = sweep-equivalents
        auto go = [&](auto kern, const char *nm, int tpr, int tyo, int zc) {
            int nby = (g.ny + tyo - 1) / tyo, nbz = (g.nz + zc - 1) / zc;
            int nblocks = nby * nbz, grid = ((nblocks + 7) / 8) * 8;
            run(C, nm, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(tpr), 0, C.s, g, C.c, C.omega, C.u, C.rhs, C.out, nby, nbz); });
        };
        for (int rep = 0; rep < 2; rep++) {
            go(k_j2<256, 1, 8, true>, "j2 TPR=256 TYO=1 ZC=8", 256, 1, 8);
            go(k_j2<256, 1, 16, true>, "j2 TPR=256 TYO=1 ZC=16", 256, 1, 16);
            go(k_j2<256, 2, 8, true>, "j2 TPR=256 TYO=2 ZC=8", 256, 2, 8);
            go(k_j2<256, 2, 12, true>, "j2 TPR=256 TYO=2 ZC=12", 256, 2, 12);
            go(k_j2<256, 2, 24, true>, "j2 TPR=256 TYO=2 ZC=24", 256, 2, 24);
            go(k_j2<256, 3, 12, true>, "j2 TPR=256 TYO=3 ZC=12", 256, 3, 12);
            go(k_j2<256, 3, 16, true>, "j2 TPR=256 TYO=3 ZC=16", 256, 3, 16);
            go(k_j2<256, 3, 24, true>, "j2 TPR=256 TYO=3 ZC=24", 256, 3, 24);
            go(k_j2<256, 4, 8, true>, "j2 TPR=256 TYO=4 ZC=8", 256, 4, 8);
            go(k_j2<256, 4, 16, true>, "j2 TPR=256 TYO=4 ZC=16", 256, 4, 16);
            go(k_j2<256, 4, 32, true>, "j2 TPR=256 TYO=4 ZC=32", 256, 4, 32);
            go(k_j2<256, 4, 64, true>, "j2 TPR=256 TYO=4 ZC=64", 256, 4, 64);
            go(k_j2<256, 2, 16, true>, "j2 TPR=256 TYO=2 ZC=16", 256, 2, 16);
            go(k_j2<256, 2, 32, true>, "j2 TPR=256 TYO=2 ZC=32", 256, 2, 32);
        }
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "pair") {
        // Two consecutive sweeps u -> A -> B.  "full": two whole-grid launches.  "chunk CH":
        // skewed z-chunk order S1(c), S2(c-1) so S2 reads A and rhs while they are still in the
        // Infinity Cache. Both orders produce identical bits (checked against the full result).
        auto launch_sub = [&](auto kern, const double *src, double *dst, int c0, int c1) {
            Geom gs = g; gs.nz = c1 - c0; gs.gz0 = c0;
            int npairs = gs.nx >> 1;
            int nbx = (npairs + 63) / 64, nby = (gs.ny + 7) / 8, nbz = (gs.nz + 2) / 3;
            int nblocks = nbx * nby * nbz, grid = ((nblocks + 7) / 8) * 8;
            long long off = (long long)c0 * g.plane;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, C.s, gs, C.c, C.omega, src + off, C.rhs + off, dst + off, nbx, nby, nbz);
        };
        double *A = C.out, *B = C.ref;
        double *Bref = nullptr;
        CK(hipMalloc(&Bref, elems * 8));
        Bref += g.plane;
        auto k_nt = k_zm<2, 4, 3, 1, 2, true, true, 3>;
        auto k_plain = k_zm<2, 4, 3, 1, 0, true, true, 3>;
        auto k_st = k_zm<2, 4, 3, 1, 1, true, true, 3>;
        launch_sub(k_nt, C.u, A, 0, g.nz); launch_sub(k_nt, A, Bref, 0, g.nz);
        CK(hipStreamSynchronize(C.s));
        auto timeit = [&](const char *name, auto fn) {
            CK(hipMemsetAsync(B - g.plane, 0xff, elems * 8, C.s));
            fn(); fn();
            CK(hipMemsetAsync(C.bad, 0, 8, C.s));
            hipLaunchKernelGGL(k_cmp, gr, dim3(64, 4), 0, C.s, g, B, Bref, C.bad);
            unsigned long long bad = 0;
            CK(hipMemcpyAsync(&bad, C.bad, 8, hipMemcpyDeviceToHost, C.s));
            CK(hipStreamSynchronize(C.s));
            CK(hipEventRecord(C.e0, C.s));
            for (int i = 0; i < C.reps; i++) fn();
            CK(hipEventRecord(C.e1, C.s));
            CK(hipEventSynchronize(C.e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, C.e0, C.e1)); ms /= C.reps;
            printf("%-44s %8.4f ms per PAIR  %8.1f GB/s/sweep-equivalent (%5.1f%%)  %s\n", name, ms, 2 * C.pts * 24.0 / ms / 1e6,
                   2 * C.pts * 24.0 / ms / 1e6 / 80.0, bad ? "MISMATCH" : "ok");
            fflush(stdout);
        };
        for (int rep = 0; rep < 2; rep++) {
            timeit("pair full (2 launches, nt)", [&] { launch_sub(k_nt, C.u, A, 0, g.nz); launch_sub(k_nt, A, B, 0, g.nz); });
            for (int CH : {6, 9, 12, 15, 18, 24, 30, 36, 48, 60}) {
                char name[96];
                for (int mode = 0; mode < 2; mode++) {
                    snprintf(name, sizeof name, "pair chunked CH=%d S1=%s S2=nt", CH, mode ? "ntstore" : "plain");
                    timeit(name, [&] {
                        int nch = (g.nz + CH - 1) / CH;
                        for (int c = 0; c <= nch; c++) {
                            if (c < nch) { if (mode) launch_sub(k_st, C.u, A, c * CH, std::min(g.nz, (c + 1) * CH)); else launch_sub(k_plain, C.u, A, c * CH, std::min(g.nz, (c + 1) * CH)); }
                            if (c > 0) launch_sub(k_nt, A, B, (c - 1) * CH, std::min(g.nz, c * CH));
                        }
                    });
                }
            }
        }
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "tail") {
        for (int rep = 0; rep < 2; rep++) {
            run_zm<2, 4, 2, 1, 2, true, 1>(C);
            run_zm<2, 4, 2, 1, 2, true, 2>(C);
            run_zm<2, 4, 2, 1, 2, true, 3>(C);
            run_zm<2, 4, 2, 1, 2, true, 0>(C);
            run_zm<2, 4, 2, 1, 2, false, 3>(C);
            run_zm<2, 4, 3, 1, 2, true, 3>(C);
            run_zm<2, 4, 4, 1, 2, true, 2>(C);
            run_zm<2, 4, 4, 1, 2, true, 3>(C);
            run_zm<2, 8, 2, 1, 2, true, 3>(C);
            run_zm<2, 2, 2, 1, 2, true, 3>(C);
            run_zm<1, 4, 2, 1, 2, true, 3>(C);
            run_zm<1, 4, 4, 1, 2, true, 3>(C);
            run_zm<1, 8, 4, 1, 2, true, 3>(C);
            run_zm<3, 4, 2, 1, 2, true, 3>(C);
            run_zm<4, 4, 2, 1, 2, true, 3>(C);
            run_zm<2, 4, 2, 0, 2, true, 3>(C);
            run_zm<2, 4, 2, 1, 1, true, 3>(C);
            run_zm<2, 4, 2, 1, 0, true, 3>(C);
        }
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "zc") {
        for (int rep = 0; rep < 3; rep++) {
            run_zm<2, 4, 1, 1, 1, true>(C);
            run_zm<2, 4, 2, 1, 1, true>(C);
            run_zm<2, 4, 4, 1, 1, true>(C);
            run_zm<2, 4, 6, 1, 1, true>(C);
            run_zm<2, 4, 8, 1, 1, true>(C);
            run_zm<2, 4, 8, 1, 2, true>(C);
            run_zm<2, 4, 8, 1, 1, false>(C);
            run_zm<2, 4, 8, 1, 1, true, 0>(C);
            run_zm<2, 4, 12, 1, 1, true>(C);
            run_zm<2, 4, 16, 1, 1, true>(C);
            run_zm<1, 4, 4, 1, 1, true>(C);
            run_zm<1, 4, 8, 1, 1, true>(C);
            run_zm<1, 4, 8, 1, 2, true>(C);
            run_zm<1, 8, 8, 1, 1, true>(C);
            run_zm<1, 4, 8, 1, 1, true, 0>(C);
            run_zm<2, 8, 8, 1, 1, true>(C);
            run_zm<2, 2, 8, 1, 1, true>(C);
            run_zm<4, 4, 8, 1, 1, true>(C);
            run_zm<4, 4, 4, 1, 1, true>(C);
        }
        return 0;
    }
    // RY, BW, ZC, XMODE, NT, SWZ
    run_zm<2, 4, 24, 1, 1, true>(C);
    run_zm<2, 4, 32, 1, 1, true>(C);
    run_zm<2, 4, 32, 1, 2, true>(C);
    run_zm<2, 4, 32, 1, 1, false>(C);
    run_zm<2, 4, 40, 1, 1, true>(C);
    run_zm<2, 4, 48, 1, 1, true>(C);
    run_zm<2, 4, 16, 1, 1, true>(C);
    run_zm<2, 4, 8, 1, 1, true>(C);
    run_zm<2, 2, 32, 1, 1, true>(C);
    run_zm<2, 8, 32, 1, 1, true>(C);
    run_zm<2, 16, 32, 1, 1, true>(C);
    run_zm<1, 4, 32, 1, 1, true>(C);
    run_zm<1, 4, 16, 1, 1, true>(C);
    run_zm<1, 8, 32, 1, 1, true>(C);
    run_zm<1, 16, 32, 1, 1, true>(C);
    run_zm<3, 4, 32, 1, 1, true>(C);
    run_zm<4, 4, 32, 1, 2, true>(C);
    run_zm<4, 4, 16, 1, 1, true>(C);
    if (argc > 3 && std::string(argv[3]) == "quick") return 0;
    run_zm<1, 4, 16, 0, 0, false>(C);
    run_zm<1, 4, 16, 1, 0, false>(C);
    run_zm<1, 4, 64, 0, 0, false>(C);
    run_zm<1, 4, 64, 0, 0, true>(C);
    run_zm<1, 8, 64, 0, 0, true>(C);
    run_zm<2, 4, 16, 0, 0, false>(C);
    run_zm<2, 4, 16, 0, 0, true>(C);
    run_zm<2, 4, 64, 0, 0, false>(C);
    run_zm<2, 4, 64, 0, 0, true>(C);
    run_zm<2, 4, 64, 1, 0, true>(C);
    run_zm<2, 4, 64, 0, 1, true>(C);
    run_zm<2, 4, 64, 1, 1, true>(C);
    run_zm<2, 4, 64, 1, 2, true>(C);
    run_zm<2, 8, 64, 1, 1, true>(C);
    run_zm<2, 4, 32, 1, 1, true>(C);
    run_zm<2, 4, 128, 1, 1, true>(C);
    run_zm<4, 4, 16, 0, 0, false>(C);
    run_zm<4, 4, 64, 0, 0, false>(C);
    run_zm<4, 4, 64, 0, 0, true>(C);
    run_zm<4, 4, 64, 1, 0, true>(C);
    run_zm<4, 4, 64, 0, 1, true>(C);
    run_zm<4, 4, 64, 1, 1, true>(C);
    run_zm<4, 4, 64, 1, 2, true>(C);
    run_zm<4, 4, 32, 1, 1, true>(C);
    run_zm<4, 4, 128, 1, 1, true>(C);
    run_zm<4, 2, 64, 1, 1, true>(C);
    run_zm<4, 8, 64, 1, 1, true>(C);
    run_zm<8, 4, 64, 1, 1, true>(C);
    run_zm<8, 2, 64, 1, 1, true>(C);
    return 0;
}
