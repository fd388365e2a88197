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
template <typename T>
struct Coef {
    T cx, cy, cz, cd;
};

struct CoarseOut {
    int iters;
    int flag;
    double relres;
    double sumsq_rhs;
    double sumsq_r;
};

}  // namespace mg
#endif
