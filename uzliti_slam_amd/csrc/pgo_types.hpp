// pgo_types.hpp — device-side view of one pose-graph problem (shared by host code and kernels).
//
// HBM layout (all f64 unless noted):
//   pose / pose_trial  [n][8]   (tx,ty,tz, qw,qx,qy,qz, pad): one 64-byte line per vertex gather
//   edge arrays, SoA   ei/ej [e] i32; zinv [7][e] = Z^-1 as (t, q); info [36][e]; robust [e] u8
//   block-CSR of H over the free vertices ("half-edge slots"): row a holds one slot per incident
//   system edge, sorted by edge index; slot s carries
//       blk [s][36]   H_{a,col[s]} = J_a^T W J_col      (col = -1 when the neighbour is fixed)
//       dcon[s][36]   this edge's share of H_aa = J_a^T W J_a
//       gcon[s][6]    this edge's share of -b_a = J_a^T W e
//   so assembly is a gather over contiguous slots: no atomics, bit-reproducible.
#pragma once
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {

constexpr int kMaxPartials = 1024;

struct PgoDev {
    int32_t n, nb, e, nslots;
    double* pose;
    double* pose_trial;
    const int32_t* v2b;      // [n]  free-block index or -1
    const int32_t* b2v;      // [nb]
    const int32_t* ei;
    const int32_t* ej;
    const double* zinv;      // [7][e]
    const double* info;      // [36][e]
    const uint8_t* robust;   // [e]
    const int32_t* slot_i;   // [e] slot of the edge in row v2b[ei], -1 if ei is fixed
    const int32_t* slot_j;
    const int32_t* row_ptr;  // [nb+1]
    const int32_t* col;      // [nslots]
    double* blk;
    double* dcon;
    double* gcon;
    double* hdiag;           // [nb][36]
    double* minv;            // [nb][36]  (H_aa + lambda I)^-1
    double* b;               // [nb][6]
    double* x;               // PCG vectors [nb][6]
    double* r;
    double* z;
    double* p;
    double* ap;
    double* part_a;          // [kMaxPartials] block partials (p.Ap, chi2, ...)
    double* part_b;          // [kMaxPartials] block partials (r.z, scale, ...)
    double* part_c;          // [kMaxPartials] block partials (max |H_jj|)
    double* scal;            // [8]: 0 rz, 1 rz threshold, 2 rz_prev, 3 lambda, 4 chi2, 5 scale, 6 diagmax
    int32_t* flags;          // [4]: 0 done, 1 iterations, 2 breakdown
};

// scalars copied back to the host after each LM trial / PCG chunk
struct PgoHostScal {
    double scal[8];
    int32_t flags[4];
};

}  // namespace uzl
