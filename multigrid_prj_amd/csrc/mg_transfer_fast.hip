// mg_transfer_fast.hip -- fast path of the 3-D prolongation (InterpolationClass::interpolate,
// reference src/multigrid.cpp:3-27, extended to 3-D, and its fine += P coarse variant), gfx950.
//
// One lane owns CV coarse columns (1 in fp64, 2 in fp32) = one aligned 16-byte vector of fine
// x; one thread produces the 2 x 2 fine (z, y) rows fed by the coarse cell (zc, yc): 4 coarse
// loads give, in the reference's phase order (slow axis first, x last),
//     Zv(y) = C(zc,y) | 0.5*(C(zc,y) + C(zc+1,y))          z even | odd
//     Y     = Zv(yc)  | 0.5*(Zv(yc) + Zv(yc+1))            y even | odd
//     fine  = Y(ic)   | 0.5*(Y(ic) + Y(ic+1))              x even | odd
// with Y(ic+1) of the next lane fetched by a DPP wave shift (the last lane of the wave loads
// its own four extra coarse values). The fine array is read and written exactly once with
// full-width vectors: 17 B per fine point in fp64 (16 fine + 1 coarse). Values are bit-identical
// to the generic k_prolong and to the oracle (same 0.5*(a+b) tree). The odd last fine column is
// written as one full 128-byte line (see mg_jacobi_fast.hip).
#include "mg_kernels.h"

namespace mg {
namespace {

template <typename T> struct PV;
template <> struct PV<double> { static constexpr int V = 2; typedef double vec __attribute__((ext_vector_type(2))); };
template <> struct PV<float> { static constexpr int V = 4; typedef float vec __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ float next_lane(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double next_lane(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int PBW = 4;  // waves per workgroup, stacked over coarse rows

// c00 = C(zc,yc), c01 = C(zc,yc+1), c10 = C(zc+1,yc), c11 = C(zc+1,yc+1)  ->  Y[2*zr + yr]
template <typename T>
__device__ __forceinline__ void interp_zy(T c00, T c01, T c10, T c11, T Y[4])
{
    const T h = (T)0.5;
    const T z1y0 = h * (c00 + c10);  // odd z at coarse row yc
    const T z1y1 = h * (c01 + c11);  // odd z at coarse row yc+1
    Y[0] = c00;                      // z even, y even
    Y[1] = h * (c00 + c01);          // z even, y odd
    Y[2] = z1y0;                     // z odd,  y even
    Y[3] = h * (z1y0 + z1y1);        // z odd,  y odd
}

template <typename T, bool ADD>
__global__ __launch_bounds__(64 * PBW) void k_prolong3d_fast(Geom gc, Geom gf, const T *__restrict__ coarse,
                                                             T *__restrict__ fine, int nbx, int nby)
{
    constexpr int V = PV<T>::V, CV = V / 2;
    typedef typename PV<T>::vec vec;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int zc = blockIdx.y;                     // local coarse plane; fine planes 2zc, 2zc+1
    const int yc = by * PBW + wv;                  // coarse row; fine rows 2yc, 2yc+1
    if (yc >= gc.ny) return;                       // wave-uniform
    const int ic0 = CV * (bx * 64 + lane);         // first coarse column of this lane
    const int x0 = 2 * ic0;                        // first fine x of this lane
    const int nvec = gf.nx / V;                    // full fine vectors per row
    const bool xin = x0 < V * nvec;
    // coarse column index clamped per column (lanes past the row end stay active for the DPP)
    auto col = [&](int m) { return min(ic0 + m, gc.nx - 1); };
    const int yc1 = min(yc + 1, gc.ny - 1);
    // zc+1 may be the coarse upper ghost plane (slab decomposition) -- it is allocated
    const long long c_z0 = (long long)zc * gc.plane, c_z1 = (long long)(zc + 1) * gc.plane;
    const long long r0 = (long long)yc * gc.pitch, r1 = (long long)yc1 * gc.pitch;

    T Y[CV][4];
#pragma unroll
    for (int m = 0; m < CV; m++)
        interp_zy<T>(coarse[c_z0 + r0 + col(m)], coarse[c_z0 + r1 + col(m)], coarse[c_z1 + r0 + col(m)],
                     coarse[c_z1 + r1 + col(m)], Y[m]);
    // Y of the next coarse column: next lane's first column, or own loads on the wave's edge
    T edge[4] = {0, 0, 0, 0};
    if (lane == 63) {
        const int ie = col(CV);
        interp_zy<T>(coarse[c_z0 + r0 + ie], coarse[c_z0 + r1 + ie], coarse[c_z1 + r0 + ie],
                     coarse[c_z1 + r1 + ie], edge);
    }
    T Yn[4];
#pragma unroll
    for (int q = 0; q < 4; q++) Yn[q] = next_lane(Y[0][q], edge[q]);

    const bool tailwave = (gf.nx % V == 1) && (bx * 64 * V <= gf.nx - 1 - V) && (gf.nx - 1 - V < (bx + 1) * 64 * V);
    // value of the last fine column (x = nx-1, even => coarse column nc-1 = next of the tail lane)
    T tailv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // broadcast the tail lane's Yn to lanes 56..63 of the wave (they write the line)
        const int src = ((gf.nx - 1 - V) / V) & 63;
        tailv[q] = __shfl(Yn[q], src, 64);
    }

#pragma unroll
    for (int zr = 0; zr < 2; zr++) {
        const int zf = 2 * zc + zr;
        if (zf >= gf.nz) break;
#pragma unroll
        for (int yr = 0; yr < 2; yr++) {
            const int yf = 2 * yc + yr;
            if (yf >= gf.ny) break;
            const int q = 2 * zr + yr;
            const long long fo = (long long)zf * gf.plane + (long long)yf * gf.pitch;
            if (xin) {
                vec v;
#pragma unroll
                for (int m = 0; m < CV; m++) {
                    const T right = (m == CV - 1) ? Yn[q] : Y[m < CV - 1 ? m + 1 : 0][q];
                    v[2 * m] = Y[m][q];
                    v[2 * m + 1] = (T)0.5 * (Y[m][q] + right);
                }
                vec *pf = (vec *)(fine + fo + x0);
                if (ADD) {
                    vec old = *pf;
#pragma unroll
                    for (int e = 0; e < V; e++) old[e] += v[e];
                    v = old;
                }
                __builtin_nontemporal_store(v, pf);
            }
            if (tailwave && lane >= 56) {
                const int j = lane - 56;
                constexpr int LINE = 128 / (int)sizeof(T);
                const int xs = gf.nx - 1 + V * j;
                const int line_end = ((gf.nx - 1) / LINE + 1) * LINE;
                if (xs < line_end) {
                    vec tv = (vec)(0);
                    if (j == 0) tv[0] = ADD ? fine[fo + gf.nx - 1] + tailv[q] : tailv[q];
                    __builtin_nontemporal_store(tv, (vec *)(fine + fo + xs));
                }
            }
        }
    }
}

}  // namespace

template <typename T>
bool prolong_fast_ok(const Geom &gc, const Geom &gf)
{
    constexpr int V = PV<T>::V;
    return gf.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 && gc.nx >= 17 && (gf.nx % V) == 1 &&
           gf.gz0 == 2 * gc.gz0 && gf.nz <= 2 * gc.nz && gf.nz >= 2 * gc.nz - 1;
}

template <typename T>
void launch_prolong_fast(hipStream_t s, const Geom &gc, const Geom &gf, const T *coarse, T *fine, bool add)
{
    constexpr int CV = PV<T>::V / 2;
    // lanes cover coarse columns 0 .. nc-2 (the last coarse column only feeds the tail)
    const int ncol = gc.nx - 1;
    const int nbx = (ncol + 64 * CV - 1) / (64 * CV);
    const int nby = (gc.ny + PBW - 1) / PBW;
    dim3 gr(nbx * nby, gc.nz), bl(64 * PBW);
    if (add) hipLaunchKernelGGL((k_prolong3d_fast<T, true>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
    else hipLaunchKernelGGL((k_prolong3d_fast<T, false>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
}

template bool prolong_fast_ok<double>(const Geom &, const Geom &);
template bool prolong_fast_ok<float>(const Geom &, const Geom &);
template void launch_prolong_fast<double>(hipStream_t, const Geom &, const Geom &, const double *, double *, bool);
template void launch_prolong_fast<float>(hipStream_t, const Geom &, const Geom &, const float *, float *, bool);

}  // namespace mg
