// mg_geom.h -- device-side description of one grid level as it lives in HBM.
//
// Layout: every level owns dense arrays of
//   (nz + 2) planes x ny rows x pitch elements,
// pitch = nx rounded up to 128 B so every row starts on a 128-byte line and lane
// l of a wave reads x = 2l,2l+1 (fp64) / 4l..4l+3 (fp32) with one aligned 16-byte
// access.  Plane 0 and plane nz+1 are ghost planes (z-halo of the slab
// decomposition; unused and zero on one GPU); `p` pointers handed to kernels
// point at local plane 0, i.e. one plane into the allocation.  Padding columns
// x >= nx are never written and stay zero.
#ifndef MG_GEOM_H
#define MG_GEOM_H

namespace mg {

struct Geom {
    int dim;          // 2 or 3 (2-D levels have nz == 1 and no z coupling)
    int nx, ny, nz;   // local extents; nx == ny == global n, nz = local planes
    int pitch;        // elements per row
    long long plane;  // elements per plane = ny * pitch
    int gz0;          // global z index of local plane 0
    int gnz;          // global number of planes (n in 3-D, 1 in 2-D)
};

// off-diagonals (negative) per axis and the diagonal of the level's operator,
// include/linear_system.hpp:27-28,37-38 of the reference
// rcd / win drive div_cd() below: rcd = RN(1/cd), win = width of the numerator's exponent window
// in which the three-operation division is exact (0 = always use the hardware division).
template <typename T>
struct Coef {
    T cx, cy, cz, cd;
    T rcd;
    unsigned win;
};

// ---- correctly rounded division by the diagonal in three operations ---------------------------
// The Jacobi / Gauss-Seidel update divides by the constant diagonal cd. An IEEE division costs
// ~20 VALU issue slots in fp64 (v_div_scale, quarter-rate v_rcp, Newton steps, v_div_fmas,
// v_div_fixup); with y = RN(1/cd) computed once on the host,
//     q = RN(a*y);  r = a - q*cd (exact, one FMA);  q' = RN(q + r*y)
// is RN(a/cd) (Markstein's theorem: y correctly rounded, q within one ulp, cd's significand not
// all ones, no intermediate under/overflow). The host checks cd (make_coef); the exponent window
// on `a` keeps q and r far from the subnormal and overflow ranges; numerators outside it (zero,
// tiny, huge, Inf, NaN) take the hardware division. Bit-identical to a / cd either way, which
// the parity tests check on every sweep they compare.
template <typename T> struct DivWindow;
template <> struct DivWindow<double> { static constexpr unsigned lo = 1023 - 900, span = 1800, cd_lo = 1023 - 100, cd_span = 200; };
template <> struct DivWindow<float> { static constexpr unsigned lo = 127 - 60, span = 120, cd_lo = 127 - 30, cd_span = 60; };

inline unsigned biased_exponent(double v) { unsigned long long u; __builtin_memcpy(&u, &v, 8); return (unsigned)(u >> 52) & 0x7ffu; }
inline unsigned biased_exponent(float v) { unsigned u; __builtin_memcpy(&u, &v, 4); return (u >> 23) & 0xffu; }

template <typename T>
inline Coef<T> make_coef(double cx, double cy, double cz, double cd)
{
    Coef<T> c{(T)cx, (T)cy, (T)cz, (T)cd, (T)0, 0u};
    const T one = (T)1;
    c.rcd = one / c.cd;  // correctly rounded by the host FPU
    unsigned long long mant, all;
    if (sizeof(T) == 8) { unsigned long long u; __builtin_memcpy(&u, &c.cd, 8); all = (1ull << 52) - 1; mant = u & all; }
    else { unsigned u; __builtin_memcpy(&u, &c.cd, 4); all = (1u << 23) - 1; mant = u & all; }
    const bool ok = (biased_exponent(c.cd) - DivWindow<T>::cd_lo) < DivWindow<T>::cd_span && mant != all;
    c.win = ok ? DivWindow<T>::span : 0u;
    return c;
}

#ifdef __HIPCC__
__device__ __forceinline__ unsigned dev_biased_exponent(double v) { return ((unsigned)__double2hiint(v) >> 20) & 0x7ffu; }
__device__ __forceinline__ unsigned dev_biased_exponent(float v) { return (__float_as_uint(v) >> 23) & 0xffu; }
__device__ __forceinline__ double dev_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float dev_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T div_cd(T a, const Coef<T> &c)
{
    if ((dev_biased_exponent(a) - DivWindow<T>::lo) < c.win) {
        const T q = a * c.rcd;
        const T r = dev_fma(-q, c.cd, a);
        return dev_fma(r, c.rcd, q);
    }
    return a / c.cd;
}

// N quotients at once, straight-line: the three-operation quotients are computed for every
// element and ONE rarely taken branch redoes the group with the hardware division when any
// numerator was outside the window (a per-element branch would split the unrolled stencil code
// into basic blocks and serialise the loads and the arithmetic of neighbouring points).
template <typename T, int N>
__device__ __forceinline__ void div_cd_n(const T (&a)[N], T (&q)[N], const Coef<T> &c)
{
    bool ok = true;
#pragma unroll
    for (int e = 0; e < N; e++) {
        const T q0 = a[e] * c.rcd;
        const T r = dev_fma(-q0, c.cd, a[e]);
        q[e] = dev_fma(r, c.rcd, q0);
        ok = ok && ((dev_biased_exponent(a[e]) - DivWindow<T>::lo) < c.win);
    }
    if (__builtin_expect(!ok, 0)) {
#pragma unroll
        for (int e = 0; e < N; e++) q[e] = a[e] / c.cd;
    }
}

// The same with a WAVE-uniform test: the three-operation quotients run for every lane with the full execution mask (the
// per-lane test above makes the compiler wrap them in execution-mask regions, 8-10 scalar instructions per group); when ANY
// lane of the wave fell outside the window the whole wave redoes the group with the hardware division -- same bits for the
// lanes that were inside it.
template <typename T, int N>
__device__ __forceinline__ void div_cd_n_wave(const T (&a)[N], T (&q)[N], const Coef<T> &c)
{
    bool ok = true;
#pragma unroll
    for (int e = 0; e < N; e++) {
        const T q0 = a[e] * c.rcd;
        const T r = dev_fma(-q0, c.cd, a[e]);
        q[e] = dev_fma(r, c.rcd, q0);
        ok = ok && ((dev_biased_exponent(a[e]) - DivWindow<T>::lo) < c.win);
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
#pragma unroll
        for (int e = 0; e < N; e++) q[e] = a[e] / c.cd;
    }
}
#endif

struct CoarseOut {
    int iters;
    int flag;
    double relres;
    double sumsq_rhs;
    double sumsq_r;
};

}  // namespace mg
#endif
