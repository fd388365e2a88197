// tools/pairbench.hip -- A/B of the fused smoothing pairs on one level: k_jacobi2 (mg_jacobi_fast.hip) against the
// wide-tile k_pairw (mg_pair_wide.hip), through the product's own launchers. Every variant is run both ways on the same
// random arrays, compared BIT FOR BIT (k_jacobi2 is the one the parity tests pin to the oracle) and timed with HIP events.
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Iinclude tools/pairbench.hip \
//            multigrid_prj_amd/csrc/mg_jacobi_fast.hip multigrid_prj_amd/csrc/mg_pair_wide.hip -o tools/pairbench
// usage: pairbench [n=513] [reps=10] [f64|f32] [whole|slab|all]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../multigrid_prj_amd/csrc/mg_kernels.h"

using namespace mg;

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ void k_diff(const unsigned *a, const unsigned *b, size_t nwords, unsigned long long *bad)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    unsigned long long n = 0;
    for (; i < nwords; i += st) n += a[i] != b[i];
    if (n) atomicAdd(bad, n);
}

static unsigned long long lcg = 88172645463325252ull;
static double rnd()
{
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
}

template <typename T>
struct Level {
    Geom g;
    int gh;
    size_t elems;
    T *base[4];  // u, rhs, outA, outB (allocation starts; local plane 0 is gh planes in)
    T *p(int k) const { return base[k] + (size_t)gh * g.plane; }
};

template <typename T>
static Level<T> make_level(int n, int gh, int narr)
{
    Level<T> L{};
    Geom &g = L.g;
    g.dim = 3; g.nx = g.ny = g.nz = n;
    const int line = 128 / (int)sizeof(T);
    g.pitch = ((n + line - 1) / line) * line;
    g.plane = (long long)g.ny * g.pitch;
    g.gz0 = 0; g.gnz = n;
    L.gh = gh;
    L.elems = (size_t)(n + 2 * gh) * g.plane;
    for (int k = 0; k < narr; k++) CK(hipMalloc(&L.base[k], L.elems * sizeof(T)));
    return L;
}

template <typename T>
static void fill_random(const Level<T> &L, int k, double scale)
{
    const Geom &g = L.g;
    std::vector<T> h(L.elems, (T)0);
    for (int z = -L.gh; z < g.nz + L.gh; z++)
        for (int y = 0; y < g.ny; y++)
            for (int x = 0; x < g.nx; x++) h[(size_t)(z + L.gh) * g.plane + (size_t)y * g.pitch + x] = (T)(scale * rnd());
    CK(hipMemcpy(L.base[k], h.data(), L.elems * sizeof(T), hipMemcpyHostToDevice));
}

template <typename T>
static int run(int n, int reps, const std::string &what)
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int gh = 2;
    Level<T> F = make_level<T>(n, gh, 4), C = make_level<T>((n + 1) / 2, gh, 1);
    fill_random(F, 0, 1.0); fill_random(F, 1, 100.0); fill_random(C, 0, 0.3);
    const double h = 1.0 / (n - 1);
    const Coef<T> c = make_coef<T>(-1.0 / (h * h), -1.0 / (h * h), -1.0 / (h * h), 6.0 / (h * h));
    unsigned long long *d_bad;
    CK(hipMalloc(&d_bad, 8));
    int fails = 0;
    const double pts = (double)n * n * n;

    // fn(wide): enqueue the launch under test writing F.p(wide ? 3 : 2)
    auto ab = [&](const char *name, double npts, double bytes_pt, auto fn) {
        float ms[2] = {0, 0};
        for (int wide = 0; wide < 2; wide++) {
            set_pair_wide(wide); set_rr_wide(wide);
            CK(hipMemsetAsync(F.base[2 + wide], 0x5a, F.elems * sizeof(T), s));
            fn(wide); fn(wide);
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; i++) fn(wide);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[wide], e0, e1));
            ms[wide] /= reps;
        }
        CK(hipMemsetAsync(d_bad, 0, 8, s));
        hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, s, (const unsigned *)F.base[2], (const unsigned *)F.base[3],
                           F.elems * sizeof(T) / 4, d_bad);
        unsigned long long bad = 0;
        CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const double gb = npts * bytes_pt / 1e9;
        printf("%-34s k_jacobi2 %8.4f ms (%5.2f TB/s)   k_pairw %8.4f ms (%5.2f TB/s, frac %.3f)   %s\n", name, ms[0],
               gb / ms[0], ms[1], gb / ms[1], gb / ms[1] / 8.0, bad ? "MISMATCH" : "bit-equal");
        if (bad) { printf("   %llu differing 32-bit words\n", bad); fails++; }
        fflush(stdout);
    };

    const T om = (T)(6.0 / 7.0);
    const double B = sizeof(T);
    // warm-up: the first launches after start-up run at other clocks than the steady state (the first row read 10-15 % slow)
    for (int i = 0; i < 40; i++) launch_jacobi2<T>(s, F.g, c, om, F.p(0), F.p(1), F.p(2), false, 0);
    CK(hipStreamSynchronize(s));
    if (what == "whole" || what == "all") {
        const Geom &g = F.g;
        ab("pair J(J(u)) damped", pts, 3 * B, [&](int w) { launch_jacobi2<T>(s, g, c, om, F.p(0), F.p(1), F.p(2 + w), false, 0); });
        double *d_part = nullptr;
        CK(hipMalloc(&d_part, sizeof(double) * 65536));
        // the variant that also sums (rhs - A u)^2 of its input (only k_pairw has it: the row-column kernel ignores the request)
        ab("pair J(J(u)) damped + norm", pts, 3 * B, [&](int w) { launch_jacobi2<T>(s, g, c, om, F.p(0), F.p(1), F.p(2 + w), false, 0, d_part); });
        ab("pair J(J(u)) omega=1", pts, 3 * B, [&](int w) { launch_jacobi2<T>(s, g, c, (T)1, F.p(0), F.p(1), F.p(2 + w), false, 0); });
        ab("pair J(J(0)) damped", pts, 2 * B, [&](int w) { launch_jacobi2<T>(s, g, c, om, F.p(0), F.p(1), F.p(2 + w), true, 0); });
        ab("pair J(J(u+Pe)) damped", pts, 3.125 * B, [&](int w) { launch_jacobi2_corr<T>(s, g, C.g, c, om, F.p(0), C.p(0), F.p(1), F.p(2 + w), 0); });
        ab("pair J(J(u+Pe)) omega=1", pts, 3.125 * B, [&](int w) { launch_jacobi2_corr<T>(s, g, C.g, c, (T)1, F.p(0), C.p(0), F.p(1), F.p(2 + w), 0); });
        ab("red-black sweep RB(u)", pts, 3 * B, [&](int w) { launch_rb_fused<T>(s, g, c, F.p(0), F.p(1), F.p(2 + w), (const T *)nullptr, g, 0); });
        ab("red-black sweep RB(u+Pe)", pts, 3.125 * B, [&](int w) { launch_rb_fused<T>(s, g, c, F.p(0), F.p(1), F.p(2 + w), C.p(0), C.g, 0); });
    }
    if (what == "slab" || what == "all") {
        // a rank's slab in the middle of the grid: nzs planes from global plane zs (even), two ghost planes either side
        for (int nzs : {64, 32}) {
            const int zs = ((n / 3) / 4) * 4;
            Geom gs = F.g; gs.nz = nzs; gs.gz0 = zs;
            Geom gcs = C.g; gcs.nz = nzs / 2; gcs.gz0 = zs / 2;
            const long long off = (long long)zs * F.g.plane, offc = (long long)(zs / 2) * C.g.plane;
            const double sp = (double)n * n * nzs;
            char nm[96];
            snprintf(nm, sizeof nm, "slab %d planes: whole, J(J(u))", nzs);
            ab(nm, sp, 3 * B, [&](int w) { launch_jacobi2<T>(s, gs, c, om, F.p(0) + off, F.p(1) + off, F.p(2 + w) + off, false, 0); });
            Geom gi = gs; gi.nz = nzs - 4; gi.gz0 = zs + 2;
            snprintf(nm, sizeof nm, "slab %d planes: interior, J(J(u))", nzs);
            ab(nm, (double)n * n * (nzs - 4), 3 * B, [&](int w) { launch_jacobi2<T>(s, gi, c, om, F.p(0) + off + 2 * F.g.plane, F.p(1) + off + 2 * F.g.plane, F.p(2 + w) + off + 2 * F.g.plane, false, 0); });
            Geom glo = gs; glo.nz = 2;
            snprintf(nm, sizeof nm, "slab %d planes: boundary x2, J(J(u))", nzs);
            ab(nm, (double)n * n * 4, 3 * B, [&](int w) { launch_jacobi2<T>(s, glo, c, om, F.p(0) + off, F.p(1) + off, F.p(2 + w) + off, false, nzs - 2); });
            snprintf(nm, sizeof nm, "slab %d planes: whole, J(J(u+Pe))", nzs);
            ab(nm, sp, 3.125 * B, [&](int w) { launch_jacobi2_corr<T>(s, gs, gcs, c, om, F.p(0) + off, C.p(0) + offc, F.p(1) + off, F.p(2 + w) + off, 0); });
            snprintf(nm, sizeof nm, "slab %d planes: interior, J(J(u+Pe))", nzs);
            ab(nm, (double)n * n * (nzs - 4), 3.125 * B, [&](int w) { launch_jacobi2_corr<T>(s, gi, gcs, c, om, F.p(0) + off + 2 * F.g.plane, C.p(0) + offc, F.p(1) + off + 2 * F.g.plane, F.p(2 + w) + off + 2 * F.g.plane, 0); });
            snprintf(nm, sizeof nm, "slab %d planes: boundary x2, J(J(u+Pe))", nzs);
            ab(nm, (double)n * n * 4, 3.125 * B, [&](int w) { launch_jacobi2_corr<T>(s, glo, gcs, c, om, F.p(0) + off, C.p(0) + offc, F.p(1) + off, F.p(2 + w) + off, nzs - 2); });
            snprintf(nm, sizeof nm, "slab %d planes: whole, RB(u)", nzs);
            ab(nm, sp, 3 * B, [&](int w) { launch_rb_fused<T>(s, gs, c, F.p(0) + off, F.p(1) + off, F.p(2 + w) + off, (const T *)nullptr, gs, 0); });
        }
    }
    if (what == "rr" || what == "all" || what == "whole") {
        // residual + full weighting: whole level, and a slab's pieces (interior coarse planes, the two boundary planes in one launch)
        auto cmp_coarse = [&](const char *name, double npts, auto fn) {
            float ms[2] = {0, 0};
            T *outc[2];
            for (int w = 0; w < 2; w++) { CK(hipMalloc(&outc[w], C.elems * sizeof(T))); CK(hipMemset(outc[w], 0x5a, C.elems * sizeof(T))); }
            for (int wide = 0; wide < 2; wide++) {
                set_rr_wide(wide);
                T *pc = outc[wide] + (size_t)C.gh * C.g.plane;
                fn(pc); fn(pc);
                CK(hipStreamSynchronize(s));
                CK(hipGetLastError());
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; i++) fn(pc);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[wide], e0, e1));
                ms[wide] /= reps;
            }
            CK(hipMemsetAsync(d_bad, 0, 8, s));
            hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, s, (const unsigned *)outc[0], (const unsigned *)outc[1], C.elems * sizeof(T) / 4, d_bad);
            unsigned long long bad = 0;
            CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            const double gb = npts * 2.125 * B / 1e9;
            printf("%-34s k_resid_restrict_fw %8.4f ms (%5.2f TB/s)   k_rrw %8.4f ms (%5.2f TB/s, frac %.3f)   %s\n", name, ms[0], gb / ms[0], ms[1],
                   gb / ms[1], gb / ms[1] / 8.0, bad ? "MISMATCH" : "bit-equal");
            if (bad) { printf("   %llu differing 32-bit words\n", bad); fails++; }
            fflush(stdout);
            CK(hipFree(outc[0])); CK(hipFree(outc[1]));
        };
        cmp_coarse("residual + restriction, whole level", pts, [&](T *pc) { launch_resid_restrict_fw<T>(s, F.g, C.g, c, F.p(0), F.p(1), pc, 0, 0); });
        for (int nzs : {64, 32}) {
            const int zs = ((n / 3) / 4) * 4;
            Geom gs = F.g; gs.nz = nzs; gs.gz0 = zs;
            Geom gcs = C.g; gcs.nz = nzs / 2; gcs.gz0 = zs / 2;
            const long long off = (long long)zs * F.g.plane, offc = (long long)(zs / 2) * C.g.plane;
            char nm[96];
            snprintf(nm, sizeof nm, "rr slab %d planes: whole", nzs);
            cmp_coarse(nm, (double)n * n * nzs, [&](T *pc) { launch_resid_restrict_fw<T>(s, gs, gcs, c, F.p(0) + off, F.p(1) + off, pc + offc, 0, 0); });
            // the pieces of resid_restrict_on_slab_t: interior coarse planes 1 .. nzc-2, then planes 0 and nzc-1 in one launch
            const int nzc = nzs / 2;
            Geom gci = gcs; gci.nz = nzc - 2; gci.gz0 = gcs.gz0 + 1;
            Geom gfi = gs; gfi.nz = 2 * (nzc - 2); gfi.gz0 = gs.gz0 + 2;
            snprintf(nm, sizeof nm, "rr slab %d planes: interior", nzs);
            cmp_coarse(nm, (double)n * n * (nzs - 4), [&](T *pc) { launch_resid_restrict_fw<T>(s, gfi, gci, c, F.p(0) + off + 2 * F.g.plane, F.p(1) + off + 2 * F.g.plane, pc + offc + C.g.plane, 0, 0); });
            Geom gc0 = gcs; gc0.nz = 1;
            Geom gf0 = gs; gf0.nz = 2;
            snprintf(nm, sizeof nm, "rr slab %d planes: boundary x2", nzs);
            cmp_coarse(nm, (double)n * n * 4, [&](T *pc) { launch_resid_restrict_fw<T>(s, gf0, gc0, c, F.p(0) + off, F.p(1) + off, pc + offc, nzc - 1, 2); });
        }
    }
    set_pair_wide(-1); set_rr_wide(-1);
    printf("%s\n", fails ? "FAILED" : "all variants bit-equal");
    return fails;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 513;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const std::string ty = argc > 3 ? argv[3] : "f64";
    const std::string what = argc > 4 ? argv[4] : "all";
    printf("pairbench n=%d reps=%d %s %s\n", n, reps, ty.c_str(), what.c_str());
    return ty == "f32" ? run<float>(n, reps, what) : run<double>(n, reps, what);
}
