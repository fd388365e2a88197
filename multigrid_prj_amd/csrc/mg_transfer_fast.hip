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

// SEMI: the transition keeps z (semi-coarsening): fine plane z <-> coarse plane z, no z phase
template <typename T, bool ADD, bool SEMI>
__global__ __launch_bounds__(64 * PBW) void k_prolong3d_fast(Geom gc, Geom gf, const T *__restrict__ coarse,
                                                             T *__restrict__ fine, int nbx, int nby)
{
    constexpr int V = PV<T>::V, CV = V / 2;
    typedef typename PV<T>::vec vec;
    (void)nby;
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
    const long long c_z0 = (long long)zc * gc.plane, c_z1 = SEMI ? c_z0 : (long long)(zc + 1) * gc.plane;
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
    for (int zr = 0; zr < (SEMI ? 1 : 2); zr++) {
        const int zf = SEMI ? zc : 2 * zc + zr;
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


// ------------------------------------------------------------------------------------------
// Fused residual + full-weighting restriction (V-cycle, non-distributed levels):
//     rhs_c(K,J,I) = sum_{dz,dy,dx} w r(2K+dz, 2J+dy, 2I+dx),   r = rhs_f - A u  on the fly,
// so the fine residual array is neither written nor read back (17 B instead of 41 B per fine
// point). Lane <-> coarse column I <-> fine x-vector (2I, 2I+1 [,2I+2, 2I+3]); wave <-> coarse
// row J (fine rows 2J-1, 2J, 2J+1); the workgroup marches the fine planes of zcc coarse planes
// with u(z-1), u(z), u(z+1) of its three residual rows in registers. Weights are applied in
// the oracle's order -- x (q r(x-1) + h r(x) + q r(x+1), r(x-1) from the previous lane by DPP),
// then y, then z through a three-stage register pipeline -- so the result is bit-identical to
// k_restrict_fw(k_residual(u)). Coarse boundary nodes inject r(2K,2J,2I).
__device__ __forceinline__ float prev_lane(float v, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double prev_lane(double v, double edge)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}


// One workgroup = CR consecutive coarse rows over their whole width: NW waves side by side in
// x; each lane evaluates the 2*CR+1 fine residual rows those coarse rows need (neighbouring
// workgroups recompute the shared odd row: 1.5x residual work for CR = 1, 1.25x for CR = 2).
// SEMI: semi-coarsening transition: every fine plane is a coarse plane, 9-point weights per plane
template <typename T, bool NTLOAD, int CR, bool SEMI>
__global__ __launch_bounds__(512) void k_resid_restrict_fw(Geom gf, Geom gc, Coef<T> c, const T *__restrict__ u_,
                                                            const T *__restrict__ rhs_, T *__restrict__ coarse_,
                                                            int nby, int nbz, int zcc, int dup_kc, int dup_nzf)
{
    constexpr int V = PV<T>::V, CV = V / 2, NR = 2 * CR + 1;
    typedef typename PV<T>::vec vec;
    __shared__ T edge[2][8][NR];  // [slot][wave][row]: last residual element of each wave
    // u on the wave edges, published one plane ahead (as in k_jacobi2): the x-neighbour of a wave's
    // first / last lane is the neighbouring wave's last / first element; only the row's very last
    // lane needs a value no wave holds (the column after the last vector), fetched one plane ahead
    __shared__ T ued[2][NR][8][2];
    __shared__ T utl[2][NR];
    // dup_kc > 0: the launch covers a second single coarse plane, dup_kc coarse planes (2 dup_kc fine planes) further up,
    // with dup_nzf fine planes: the second half of the workgroups shift pointers and plane indices (all scalar)
    const int nblocks = nby * nbz, ntotal = dup_kc > 0 ? 2 * nblocks : nblocks;
    const int per = (ntotal + 7) >> 3;
    int bid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware order
    if (bid >= ntotal) return;                                    // whole workgroup
    const bool second = bid >= nblocks;
    if (second) { bid -= nblocks; gf.gz0 += 2 * dup_kc; gf.nz = dup_nzf; gc.gz0 += dup_kc; }
    const long long foff = second ? (long long)2 * dup_kc * gf.plane : 0;
    const T *__restrict__ u = u_ + foff;
    const T *__restrict__ rhs = rhs_ + foff;
    T *__restrict__ coarse = coarse_ + (second ? (long long)dup_kc * gc.plane : 0);
    const int J0 = (bid % nby) * CR, bz = bid / nby;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (int)(blockDim.x >> 6);
    const int wl = max(wv - 1, 0), wr = min(wv + 1, nwv - 1);
    const bool lastlane = (wv == nwv - 1) && (lane == 63);
    const int ic0 = CV * (wv * 64 + lane);   // first coarse column of the lane
    const int x0 = 2 * ic0;                  // first fine x
    const int ecol = (x0 + V <= gf.nx - 1) ? V : 0;  // a lane clamped past the row end re-reads its own element (value unused)
    const int x0c = min(x0, gf.pitch - V);   // clamped for loads; every lane stays active
    const bool cin = ic0 + CV - 1 <= gc.nx - 2;      // owns CV real coarse columns (tail column excluded)
    const bool tail_lane = (x0 + V == gf.nx - 1);    // the lane next to the odd last fine column
    const int K0 = bz * zcc, K1 = min(K0 + zcc, gc.nz);  // zcc coarse planes per workgroup
    // fine rows 2*J0-1 .. 2*J0+2*CR-1 (clamped rows only ever feed injecting boundary nodes)
    long long ro[NR];
    bool ybnd[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int y = min(max(2 * J0 - 1 + r, 0), gf.ny - 1);
        ybnd[r] = (y == 0) || (y == gf.ny - 1);
        ro[r] = (long long)y * gf.pitch + x0c;
    }
    const long long ro_lo = (long long)min(max(2 * J0 - 2, 0), gf.ny - 1) * gf.pitch + x0c;
    const long long ro_hi = (long long)min(2 * J0 + 2 * CR, gf.ny - 1) * gf.pitch + x0c;
    const T q = (T)0.25, h = (T)0.5;

    // fine planes to evaluate (all inside the GLOBAL grid): 2K0-1 .. 2(K1-1)+1, or K0 .. K1-1 when z is kept. On a z-slab
    // (gf.gz0 > 0: the coarse slab starts at the fine slab's first plane) the first one is the lower ghost plane -1, whose
    // residual needs u on plane -2: the caller exchanged two ghost planes of u and one of rhs.
    const int zs = SEMI ? K0 : max(2 * K0 - 1, -gf.gz0), ze = SEMI ? K1 - 1 : min(2 * (K1 - 1) + 1, gf.nz - 1);
    vec um[NR], uc[NR], up[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        um[r] = *(const vec *)(u + (long long)(zs - 1) * gf.plane + ro[r]);  // zs-1 >= -1: ghost plane
        uc[r] = *(const vec *)(u + (long long)zs * gf.plane + ro[r]);
    }
    {
        const int sl = zs & 1;   // (-1 & 1 == 1: the parity of a ghost plane is as good as any other)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (lane == 0) ued[sl][r][wv][0] = uc[r][0];
            if (lane == 63) ued[sl][r][wv][1] = uc[r][V - 1];
            if (lastlane) utl[sl][r] = u[(long long)zs * gf.plane + ro[r] + ecol];
        }
    }
    __syncthreads();
    T ywm[CR][CV], ywc[CR][CV], ctr[CR][CV], ctr_tail[CR];
#pragma unroll
    for (int j = 0; j < CR; j++) {
        ctr_tail[j] = 0;
#pragma unroll
        for (int m = 0; m < CV; m++) { ywm[j][m] = 0; ywc[j][m] = 0; ctr[j][m] = 0; }
    }
    int slot = 0;

    for (int z = zs; z <= ze; z++) {
        const long long zo = (long long)z * gf.plane;
        const T *pz = u + zo;
        vec b[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            up[r] = *(const vec *)(pz + gf.plane + ro[r]);  // z+1 <= nz: ghost plane
            if (NTLOAD) b[r] = __builtin_nontemporal_load((const vec *)(rhs + zo + ro[r]));
            else b[r] = *(const vec *)(rhs + zo + ro[r]);
        }
        const vec hlo = *(const vec *)(pz + ro_lo);
        const vec hhi = *(const vec *)(pz + ro_hi);
        // wave-edge scalars and the Dirichlet tail column: loaded here with everything else (inside
        // the row loop each of them cost its own memory round trip: three per plane and wave)
        T elv[NR], erv[NR], ern[NR], tlb[CR], tlu[CR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            ern[r] = 0;
            if (lastlane) ern[r] = pz[gf.plane + ro[r] + ecol];  // plane z+1, consumed in the next step
            elv[r] = ued[z & 1][r][wl][1];
            erv[r] = (wv == nwv - 1) ? utl[z & 1][r] : ued[z & 1][r][wr][0];
        }
#pragma unroll
        for (int j = 0; j < CR; j++) {
            tlb[j] = 0; tlu[j] = 0;
            if (tail_lane) {
                const long long i = zo + (ro[2 * j + 1] - x0c) + gf.nx - 1;
                tlb[j] = rhs[i]; tlu[j] = u[i];
            }
        }
        const int gzf = gf.gz0 + z;
        const bool zb = (gzf == 0) || (gzf == gf.gnz - 1);
        vec res[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const T el = elv[r], er = erv[r];
            const T xm = prev_lane(uc[r][V - 1], el);
            const T xp = next_lane(uc[r][0], er);
            const vec ym = (r == 0) ? hlo : uc[r > 0 ? r - 1 : 0];
            const vec yp = (r == NR - 1) ? hhi : uc[r < NR - 1 ? r + 1 : 0];
            const bool rb = zb || ybnd[r];
#pragma unroll
            for (int e = 0; e < V; e++) {
                const T left = (e == 0) ? xm : uc[r][e > 0 ? e - 1 : 0];
                const T right = (e == V - 1) ? xp : uc[r][e < V - 1 ? e + 1 : 0];
                const bool bnd = rb || (x0 + e == 0) || (x0 + e == gf.nx - 1);
                T sum = 0;
                sum += c.cz * um[r][e];
                sum += c.cy * ym[e];
                sum += c.cx * left;
                sum += c.cd * uc[r][e];
                sum += c.cx * right;
                sum += c.cy * yp[e];
                sum += c.cz * up[r][e];
                if (bnd) sum = (T)1 * uc[r][e];
                res[r][e] = b[r][e] - sum;
            }
            if (lane == 63) edge[slot][wv][r] = res[r][V - 1];
        }
        {
            const int sl = (z + 1) & 1;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (lane == 0) ued[sl][r][wv][0] = up[r][0];
                if (lane == 63) ued[sl][r][wv][1] = up[r][V - 1];
                if (lastlane) utl[sl][r] = ern[r];
            }
        }
        __syncthreads();
        T xw[NR][CV];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const T from_left_wave = (lane == 0 && wv > 0) ? edge[slot][wv - 1][r] : (T)0;
            const T rprev = prev_lane(res[r][V - 1], from_left_wave);
#pragma unroll
            for (int m = 0; m < CV; m++) {
                const T rleft = (m == 0) ? rprev : res[r][2 * m - 1 > 0 ? 2 * m - 1 : 0];
                xw[r][m] = q * rleft + h * res[r][2 * m] + q * res[r][2 * m + 1];
            }
        }
        slot ^= 1;
        int emitK = -1;
        if (SEMI) emitK = z;                             // planes map one to one
        else if (z & 1) emitK = (z - 1) >> 1;            // z = 2K+1 closes coarse plane K
        else if (gzf == gf.gnz - 1) emitK = z >> 1;      // top boundary plane of the grid has no z+1: it injects
        const bool centre = SEMI || !(z & 1);
#pragma unroll
        for (int j = 0; j < CR; j++) {
            const int J = J0 + j;
            T yw[CV];
#pragma unroll
            for (int m = 0; m < CV; m++) yw[m] = q * xw[2 * j][m] + h * xw[2 * j + 1][m] + q * xw[2 * j + 2][m];
            if (centre) {                                // z = 2K (or any plane when z is kept): centre plane
#pragma unroll
                for (int m = 0; m < CV; m++) { ywc[j][m] = yw[m]; ctr[j][m] = res[2 * j + 1][2 * m]; }
                if (tail_lane) ctr_tail[j] = tlb[j] - (T)1 * tlu[j];  // odd last fine column (Dirichlet): r = rhs - u
            }
            if (emitK >= K0 && emitK < K1 && J < gc.ny) {
                const bool Kbnd = (gc.gz0 + emitK == 0) || (gc.gz0 + emitK == gc.gnz - 1);
                const bool Jbnd = (J == 0) || (J == gc.ny - 1);
                const long long co = (long long)emitK * gc.plane + (long long)J * gc.pitch;
#pragma unroll
                for (int m = 0; m < CV; m++) {
                    const int I = ic0 + m;
                    const bool Ibnd = (I == 0) || (I == gc.nx - 1);
                    const T fw = SEMI ? yw[m] : q * ywm[j][m] + h * ywc[j][m] + q * yw[m];
                    if (cin) coarse[co + I] = (Kbnd || Jbnd || Ibnd) ? ctr[j][m] : fw;
                }
                if (tail_lane) coarse[co + gc.nx - 1] = ctr_tail[j];  // coarse column nc-1 is a boundary node
            }
            if (!SEMI && (z & 1)) {
#pragma unroll
                for (int m = 0; m < CV; m++) ywm[j][m] = yw[m];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { um[r] = uc[r]; uc[r] = up[r]; }
    }
}

}  // namespace

static bool transfer_is_semi(const Geom &gf, const Geom &gc) { return gf.dim == 3 && gf.gnz == gc.gnz && gf.gnz > 1; }

template <typename T>
bool prolong_fast_ok(const Geom &gc, const Geom &gf)
{
    constexpr int V = PV<T>::V;
    if (!(gf.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 && gc.nx >= 17 && (gf.nx % V) == 1)) return false;
    if (transfer_is_semi(gf, gc)) return gf.gz0 == gc.gz0 && gf.nz == gc.nz;
    return gf.gz0 == 2 * gc.gz0 && gf.nz <= 2 * gc.nz && gf.nz >= 2 * gc.nz - 1;
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
    if (transfer_is_semi(gf, gc)) {
        if (add) hipLaunchKernelGGL((k_prolong3d_fast<T, true, true>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
        else hipLaunchKernelGGL((k_prolong3d_fast<T, false, true>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
    } else {
        if (add) hipLaunchKernelGGL((k_prolong3d_fast<T, true, false>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
        else hipLaunchKernelGGL((k_prolong3d_fast<T, false, false>), gr, bl, 0, s, gc, gf, coarse, fine, nbx, nby);
    }
}

template bool prolong_fast_ok<double>(const Geom &, const Geom &);
template bool prolong_fast_ok<float>(const Geom &, const Geom &);
template void launch_prolong_fast<double>(hipStream_t, const Geom &, const Geom &, const double *, double *, bool);
template void launch_prolong_fast<float>(hipStream_t, const Geom &, const Geom &, const float *, float *, bool);

}  // namespace mg

namespace mg {

template <typename T>
bool resid_restrict_fast_ok(const Geom &gf, const Geom &gc)
{
    constexpr int V = PV<T>::V;
    const bool zok = transfer_is_semi(gf, gc) ? gf.nz == gc.nz : gf.nz == 2 * gc.nz - 1;
    return gf.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 && zok &&
           gf.gz0 == 0 && gc.gz0 == 0 && gf.gnz == gf.nz && gc.gnz == gc.nz && gc.nx >= 17 && (gf.nx % V) == 1 &&
           (gc.nx - 1 + 64 * (V / 2) - 1) / (64 * (V / 2)) <= 8;
}

// the same kernel on a z-slab of a distributed level: the coarse slab (the next level's, or the staging slab of the first
// gathered level) holds the coarse planes that coincide with this rank's fine planes
template <typename T>
bool resid_restrict_slab_ok(const Geom &gf, const Geom &gc)
{
    constexpr int V = PV<T>::V;
    const bool semi = transfer_is_semi(gf, gc);
    const bool zok = semi ? (gf.nz == gc.nz && gf.gz0 == gc.gz0)
                          : (gf.gz0 == 2 * gc.gz0 && (gf.nz == 2 * gc.nz || gf.nz == 2 * gc.nz - 1) && gf.gnz == 2 * gc.gnz - 1);
    return gf.dim == 3 && gf.nx == 2 * gc.nx - 1 && gf.ny == 2 * gc.ny - 1 && zok && gc.nz >= 1 && gc.nx >= 17 && (gf.nx % V) == 1 &&
           (gc.nx - 1 + 64 * (V / 2) - 1) / (64 * (V / 2)) <= 8;
}

template <typename T>
void launch_resid_restrict_fw(hipStream_t s, const Geom &gf, const Geom &gc, const Coef<T> &c, const T *u,
                              const T *rhs, T *coarse, int dup_kc, int dup_nzf)
{
    if (rr_wide_ok<T>(gf, gc)) { launch_rr_wide<T>(s, gf, gc, c, u, rhs, coarse, dup_kc, dup_nzf); return; }
    constexpr int CV = PV<T>::V / 2;
    constexpr int CR = 1;                                     // coarse rows per workgroup (2 measured slower again after the mailbox change: 185 VGPRs, 3.28 vs 3.10 ms per cycle)
    const int ncol = gc.nx - 1;                               // coarse columns owned by lanes
    const int nw = (ncol + 64 * CV - 1) / (64 * CV);          // waves side by side in x (<= 8)
    const int nby = (gc.ny + CR - 1) / CR;
    static const int zcc_env = [] { const char *e = getenv("MG_RR_ZCC"); return e ? atoi(e) : 0; }();
    static const int zcc_big = [] { const char *e = getenv("MG_RR_ZCC_BIG"); return e ? atoi(e) : 0; }();
    // Coarse planes marched per workgroup: 4; 7 on a level whose arrays stream from HBM (>= 512 MB: 513^3 fp64 and up). Measured
    // at 513^3 over 4 ... 43 on two boxes (tools/sweep_env.sh MG_RR_ZCC_BIG): 6, 7, 10, 11, 14 take 0.52-0.55 ms where 4, 8, 9, 12,
    // 13, 16 take 0.56-0.61 (11 -> 9 u planes read per 8 owned; the in-between values lose it again to how their chunk stride
    // falls on the memory channels); 7 was the best value that was good on both. Slabs and smaller levels: no difference (tools/dry_sweep.sh).
    int zcc = zcc_env > 0 ? zcc_env : 4;
    if (zcc_env <= 0 && (size_t)gf.nz * gf.plane * sizeof(T) >= ((size_t)512 << 20)) zcc = zcc_big > 0 ? zcc_big : 7;
    // small levels are latency-bound (one dependent memory round trip per marched plane) and their grids do not fill
    // the chip: one coarse plane per workgroup there
    if (zcc_env <= 0 && nby * ((gc.nz + zcc - 1) / zcc) < 1024) zcc = 1;
    const int nbz = (gc.nz + zcc - 1) / zcc;
    if (gc.nz != 1) dup_kc = 0;
    const int nblocks = nby * nbz, grid = (((dup_kc > 0 ? 2 : 1) * nblocks + 7) / 8) * 8;
    // rhs with ordinary loads: the odd fine row between two coarse rows is read by two workgroups, and a non-temporal first read
    // made the second one miss (0.537 -> 0.519 ms at 513^3, three alternating runs each; MG_RR_NT=1 restores the streaming loads)
    static const bool nt_env = [] { const char *e = getenv("MG_RR_NT"); return e && e[0] == '1'; }();
    const bool nt = nt_env && (size_t)gf.nz * gf.plane * sizeof(T) >= ((size_t)64 << 20);
    const dim3 bl(64 * nw);
#define MG_RR(NT, SEMI) \
    hipLaunchKernelGGL((k_resid_restrict_fw<T, NT, CR, SEMI>), dim3(grid), bl, 0, s, gf, gc, c, u, rhs, coarse, nby, nbz, zcc, dup_kc, dup_nzf)
    if (transfer_is_semi(gf, gc)) { if (nt) MG_RR(true, true); else MG_RR(false, true); }
    else { if (nt) MG_RR(true, false); else MG_RR(false, false); }
#undef MG_RR
}

template bool resid_restrict_slab_ok<double>(const Geom &, const Geom &);
template bool resid_restrict_slab_ok<float>(const Geom &, const Geom &);
template bool resid_restrict_fast_ok<double>(const Geom &, const Geom &);
template bool resid_restrict_fast_ok<float>(const Geom &, const Geom &);
template void launch_resid_restrict_fw<double>(hipStream_t, const Geom &, const Geom &, const Coef<double> &, const double *, const double *, double *, int, int);
template void launch_resid_restrict_fw<float>(hipStream_t, const Geom &, const Geom &, const Coef<float> &, const float *, const float *, float *, int, int);

}  // namespace mg
