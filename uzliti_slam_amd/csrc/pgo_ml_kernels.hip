// pgo_ml_kernels.hip — additive multilevel preconditioner for the PCG of the pose-graph solve (G8).
//
// Why: block-Jacobi PCG on a 1k..20k-vertex pose graph needs ~10^3 iterations per LM step; the slowly
// converging error is "chunks of trajectory moving rigidly".  Aggregating 8 consecutive vertices and giving
// each aggregate the 6 rigid-body modes (world twist about its centroid = the exact gauge null space of a
// pose graph) as coarse space, recursively, removes that error: ~10x fewer iterations, independent of lambda.
// (pgo_types.hpp has the formulas; DESIGN.md has the measurements.)
//
// Per linearisation : geometry (centroids, P), Galerkin products A_{l+1} = P^T A_l P level by level
//                     (transform -> contribution array -> ordered reduce: deterministic, no atomics)
// Per LM trial      : D_l(lambda)^-1 = (G_l + lambda M_l)^-1, dense inverse of the <= 48-dof top level
// Per PCG iteration : ml_spmv (p, A p, restricted A p) -> ml_cg (alpha, coarse chain in LDS, x, r, z)
#include <hip/hip_ext.h>
#include <mutex>
#include <type_traits>

#include "pgo_device.hpp"
#include "uzl_common.hpp"

namespace uzl {

// Diagnostic build only (-DUZL_STAMPS): where a latency-bound kernel spends its time.  Block 0 / thread 0 adds
// the 100 MHz s_memrealtime deltas between labelled points into a global table that a test reads back; the
// table is read by nothing else and no output depends on it.
#ifdef UZL_STAMPS
__device__ unsigned long long g_stamps[64];
#define STAMP_DECL unsigned long long st_prev_ = __builtin_amdgcn_s_memrealtime(); int st_i_ = 0;
#define STAMP(base) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); \
        atomicAdd(&g_stamps[(base) + st_i_], n_ - st_prev_); st_prev_ = n_; } st_i_++; } while (0)
#else
#define STAMP_DECL
#define STAMP(base) do { } while (0)
#endif

struct P3 { double X[9], Y[9], Z[9]; };       // P = [[X, Y], [0, Z]]

__device__ __forceinline__ void mat3(const double* A, const double* B, double* C)          // C = A B
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_acc(const double* A, const double* B, double* C)      // C += A B
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] += A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ __forceinline__ void matT3(const double* A, const double* B, double* C)         // C = A^T B
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
__device__ __forceinline__ void matT3_acc(const double* A, const double* B, double* C)     // C += A^T B
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] += A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}

// prolongation block of entity i at level f (towards level f+1)
__device__ __forceinline__ void make_P(int f, const double* __restrict__ geo, int i, P3& P)
{
    if (f == 0) {
        const double* g = geo + (size_t)i * 12;
        const double dx = g[9], dy = g[10], dz = g[11];
        const double S[9] = {0, dz, -dy, -dz, 0, dx, dy, -dx, 0};       // -[d]x
#pragma unroll
        for (int k = 0; k < 9; k++) { P.X[k] = g[k]; P.Z[k] = 0.5 * g[k]; }
        mat3(P.X, S, P.Y);
    } else {
        const double* g = geo + (size_t)i * 3;
        const double dx = g[0], dy = g[1], dz = g[2];
#pragma unroll
        for (int k = 0; k < 9; k++) { P.X[k] = (k % 4 == 0) ? 1. : 0.; P.Z[k] = P.X[k]; }
        P.Y[0] = 0; P.Y[1] = dz; P.Y[2] = -dy; P.Y[3] = -dz; P.Y[4] = 0; P.Y[5] = dx; P.Y[6] = dy; P.Y[7] = -dx; P.Y[8] = 0;
    }
}

// T = PL^T F PR (all 6x6 row-major, F given as pointer)
__device__ __forceinline__ void galerkin(const P3& L, const double* __restrict__ F, const P3& R, double* __restrict__ T)
{
    double F11[9], F12[9], F21[9], F22[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            F11[r * 3 + c] = F[r * 6 + c]; F12[r * 3 + c] = F[r * 6 + 3 + c];
            F21[r * 3 + c] = F[(3 + r) * 6 + c]; F22[r * 3 + c] = F[(3 + r) * 6 + 3 + c];
        }
    double G11[9], G12[9], G21[9], G22[9];
    mat3(F11, R.X, G11); mat3(F11, R.Y, G12); mat3_acc(F12, R.Z, G12);
    mat3(F21, R.X, G21); mat3(F21, R.Y, G22); mat3_acc(F22, R.Z, G22);
    double T11[9], T12[9], T21[9], T22[9];
    matT3(L.X, G11, T11); matT3(L.X, G12, T12);
    matT3(L.Y, G11, T21); matT3_acc(L.Z, G21, T21);
    matT3(L.Y, G12, T22); matT3_acc(L.Z, G22, T22);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            T[r * 6 + c] = T11[r * 3 + c]; T[r * 6 + 3 + c] = T12[r * 3 + c];
            T[(3 + r) * 6 + c] = T21[r * 3 + c]; T[(3 + r) * 6 + 3 + c] = T22[r * 3 + c];
        }
}

// ---- geometry of level l (>= 1): centroid of every aggregate, then the children's offsets d.  cen = {x, y, z, vertices underneath}.
//      An EMPTY row (b2v < 0: padding of a strong-aggregate numbering, pgo_schur.hpp) has no pose: it carries no weight and gets a zero
//      prolongation block (R^T = 0), so it adds nothing to any coarse operator.
__device__ __forceinline__ void ml_geometry_one(const PgoDev& D, const MlDev& ml, const double* __restrict__ pose, int l, int A, bool centroid_only = false)
{
    const int nc = ml.lv[l - 1].n;
    const int fan = ml.lv[l].fan;
    const int c0 = A * fan, c1 = (c0 + fan < nc) ? c0 + fan : nc;
    double cx = 0, cy = 0, cz = 0, wsum = 0;
    for (int c = c0; c < c1; c++) {
        double px, py, pz, w;
        if (l == 1) {
            const int v = D.b2v[c];
            if (v < 0) continue;
            const Pose P = load_pose(pose, v);
            px = P.t.x; py = P.t.y; pz = P.t.z; w = 1.;
        } else {
            const double* cc = ml.lv[l - 1].cen + (size_t)c * 4;
            px = cc[0]; py = cc[1]; pz = cc[2]; w = cc[3];
        }
        cx += w * px; cy += w * py; cz += w * pz; wsum += w;
    }
    if (wsum > 0.) { cx /= wsum; cy /= wsum; cz /= wsum; }
    double* cen = ml.lv[l].cen + (size_t)A * 4;
    cen[0] = cx; cen[1] = cy; cen[2] = cz; cen[3] = wsum;
    if (centroid_only) return;
    for (int c = c0; c < c1; c++) {
        if (l == 1) {
            double* g = ml.lv[0].geo + (size_t)c * 12;
            const int v = D.b2v[c];
            if (v < 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) g[k] = 0.;
                continue;
            }
            const Pose P = load_pose(pose, v);
            const M33 R = qrot(P.q);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) g[r * 3 + k] = R.m[k * 3 + r];       // R^T
            g[9] = P.t.x - cx; g[10] = P.t.y - cy; g[11] = P.t.z - cz;
        } else {
            const double* cc = ml.lv[l - 1].cen + (size_t)c * 4;
            double* g = ml.lv[l - 1].geo + (size_t)c * 3;
            g[0] = cc[0] - cx; g[1] = cc[1] - cy; g[2] = cc[2] - cz;
        }
    }
}
// l >= 1: that level, one lane per aggregate.  l = 0: ALL levels by one workgroup, level after level (a level needs the centroids of the
// one below; hierarchies of up to kGeoAllMax level-1 aggregates: two launches less per rebuild of a config-2-sized graph)
constexpr int kGeoAllMax = 1024;
__device__ __forceinline__ void ml_geometry_kernel_body(PgoDev D, const MlDev* __restrict__ mlp,
                                                          const double* __restrict__ pose, int l)
{
    const MlDev& ml = *mlp;
    if (l > 0) {
        const int A = blockIdx.x * kBlk + threadIdx.x;
        if (A < ml.lv[l].n) ml_geometry_one(D, ml, pose, l, A);
        return;
    }
    if (blockIdx.x != 0) return;
    for (int q = 1; q <= ml.levels; q++) {
        // centroids: a lane per aggregate; the children's offsets (level 1: R^T and the offset of every vertex): a lane per CHILD - eight
        // times the lanes for the part that was eight sequential pose loads and rotations per lane
        for (int A = threadIdx.x; A < ml.lv[q].n; A += kBlk) ml_geometry_one(D, ml, pose, q, A, true);
        __syncthreads();                                   // (workgroup-scope release / acquire: the centroids just written are read next)
        const int fan = ml.lv[q].fan, nc = ml.lv[q - 1].n;
        for (int c = threadIdx.x; c < nc; c += kBlk) {
            const double* __restrict__ cen = ml.lv[q].cen + (size_t)(c / fan) * 4;
            if (q == 1) {
                double* g = ml.lv[0].geo + (size_t)c * 12;
                const int v = D.b2v[c];
                if (v < 0) {
#pragma unroll
                    for (int k = 0; k < 12; k++) g[k] = 0.;
                } else {
                    const Pose P = load_pose(pose, v);
                    const M33 R = qrot(P.q);
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int k = 0; k < 3; k++) g[r * 3 + k] = R.m[k * 3 + r];       // R^T
                    g[9] = P.t.x - cen[0]; g[10] = P.t.y - cen[1]; g[11] = P.t.z - cen[2];
                }
            } else {
                const double* cc = ml.lv[q - 1].cen + (size_t)c * 4;
                double* g = ml.lv[q - 1].geo + (size_t)c * 3;
                g[0] = cc[0] - cen[0]; g[1] = cc[1] - cen[1]; g[2] = cc[2] - cen[2];
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(kBlk) void ml_geometry_kernel(PgoDev D, const MlDev* __restrict__ mlp, const double* __restrict__ pose, int l)
{
    ml_geometry_kernel_body(D, mlp, pose, l);
}

// ---- the same Galerkin product as ONE kernel per level: a GATHER.  The host cuts the coarse level's output blocks into chunks of
//      consecutive blocks with <= kGalItems contributions (MlLevel::chunk / cslot, build_ml); a workgroup transforms its chunk's
//      contributions into LDS (one lane each, through its two prolongation blocks) and sums them per output block in contribution
//      order - the sums of rounds 1-3's transform + ordered-reduce pair in the same order, the same bits - with no contribution array in
//      memory and one launch per level instead of two.  A block with more contributions than fit (the top of a dense hierarchy) is a
//      chunk of its own, in passes.
__device__ __forceinline__ void ml_galerkin_kernel_body(PgoDev D, const MlDev* __restrict__ mlp, int f)
{
    __shared__ double sc[kGalItems * 36];
    const MlDev& ml = *mlp;
    const MlLevel& L = ml.lv[f];
    const MlLevel& C = ml.lv[f + 1];
    if ((int)blockIdx.x >= C.n_chunks) return;
    const int32_t* __restrict__ ch = C.chunk + 5 * (size_t)blockIdx.x;
    const int kind = ch[0], o0 = ch[1], no = ch[2], q0 = ch[3], nq = ch[4];
    const int tid = threadIdx.x, ko = tid % 36, oo = tid / 36;          // reduce: lane (output oo of a round of 7, element ko)
    const double* __restrict__ geo = L.geo;
    const double* __restrict__ Fblk = (f == 0) ? D.blk : L.blk;
    const int32_t* __restrict__ Fcol = (f == 0) ? D.col : L.col;
    auto contribution = [&](int q, int slot_in_lds) {                    // P_row^T F P_col of fine slot cslot[q]
        const int s = C.cslot[q];
        P3 PL, PR;
        make_P(f, geo, L.srow[s], PL);
        make_P(f, geo, Fcol[s], PR);
        double T[36];
        galerkin(PL, Fblk + (size_t)s * 36, PR, T);
#pragma unroll
        for (int k = 0; k < 36; k++) sc[slot_in_lds * 36 + k] = T[k];
    };
    if (kind == 0) {                                                     // ---- off-diagonal blocks [o0, o0 + no)
        double acc = 0.;
        for (int base = 0; base < nq; base += kGalItems) {
            const int cnt = min(kGalItems, nq - base);
            __syncthreads();
            if (tid < cnt) contribution(q0 + base + tid, tid);
            __syncthreads();
            if (no == 1) { if (tid < 36) for (int i = 0; i < cnt; i++) acc += sc[i * 36 + tid]; }
            else if (tid < 252) {
                for (int o = oo; o < no; o += 7) {
                    const int b = o0 + o;
                    double t = 0.;
                    for (int q = C.off_ptr[b] - q0; q < C.off_ptr[b + 1] - q0; q++) t += sc[q * 36 + ko];
                    C.blk[(size_t)b * 36 + ko] = t;
                }
            }
        }
        if (no == 1 && tid < 36) C.blk[(size_t)o0 * 36 + tid] = acc;
        return;
    }
    // ---- diagonal blocks of aggregates [o0, o0 + no): same-aggregate off-diagonal contributions, then the children's G and M
    const int fan = C.fan, nf = L.n;
    const int c_first = o0 * fan, c_last = min(nf, (o0 + no) * fan), nch = c_last - c_first;
    const bool zero_children = f == 0 && !D.diag_owner;                  // sharded solve: level-1 arrays are summed over ranks afterwards
    auto children = [&](int at) {                                        // items [at, at + nch): P^T G P, [at + nch, at + 2 nch): P^T M P
        if (tid < 2 * nch) {
            const int j = tid % nch, c = c_first + j, which = tid / nch;
            double T[36];
            if (zero_children) {
#pragma unroll
                for (int k = 0; k < 36; k++) T[k] = 0.;
            } else {
                P3 P;
                make_P(f, geo, c, P);
                if (which == 0) galerkin(P, ((f == 0) ? D.hdiag : L.G) + (size_t)c * 36, P, T);
                else if (f == 0) {
                    const double I6[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
                    galerkin(P, I6, P, T);
                } else galerkin(P, L.M + (size_t)c * 36, P, T);
            }
#pragma unroll
            for (int k = 0; k < 36; k++) sc[(at + tid) * 36 + k] = T[k];
        }
    };
    const int qd0 = q0 - C.n_off_contrib;                                // = diag_ptr[o0]
    if (nq + 2 * nch <= kGalItems) {                                     // one pass
        if (tid < nq) contribution(q0 + tid, tid);
        else if (tid - nq < 2 * nch) { /* children below: their lanes are the first 2 nch - keep the mapping simple */ }
        __syncthreads();
        children(nq);
        __syncthreads();
        if (tid < 252) {
            for (int o = oo; o < no; o += 7) {
                const int A = o0 + o;
                double t = 0., m = 0.;
                for (int q = C.diag_ptr[A] - qd0; q < C.diag_ptr[A + 1] - qd0; q++) t += sc[q * 36 + ko];
                const int ca = A * fan, cb = min(nf, ca + fan);
                for (int c = ca; c < cb; c++) { t += sc[(nq + c - c_first) * 36 + ko]; m += sc[(nq + nch + c - c_first) * 36 + ko]; }
                C.G[(size_t)A * 36 + ko] = t;
                C.M[(size_t)A * 36 + ko] = m;
            }
        }
        return;
    }
    // one aggregate with more contributions than fit: in passes, then its children
    double acc = 0.;
    for (int base = 0; base < nq; base += kGalItems) {
        const int cnt = min(kGalItems, nq - base);
        __syncthreads();
        if (tid < cnt) contribution(q0 + base + tid, tid);
        __syncthreads();
        if (tid < 36) for (int i = 0; i < cnt; i++) acc += sc[i * 36 + tid];
    }
    __syncthreads();
    children(0);
    __syncthreads();
    if (tid < 36) {
        double m = 0.;
        for (int j = 0; j < nch; j++) { acc += sc[j * 36 + tid]; m += sc[(nch + j) * 36 + tid]; }
        C.G[(size_t)o0 * 36 + tid] = acc;
        C.M[(size_t)o0 * 36 + tid] = m;
    }
}
__global__ __launch_bounds__(kBlk) void ml_galerkin_kernel(PgoDev D, const MlDev* __restrict__ mlp, int f)
{
    ml_galerkin_kernel_body(D, mlp, f);
}

// ---- per LM trial: the sibling-block smoothers and the top level: dense SPD inverses of 24 .. 96 rows.
// In-place Gauss-Jordan (no pivoting: SPD) of a matrix held as 6 x 6 tiles IN REGISTERS: lane (I, J) of an LW x LW square of lanes owns tile
// (I, J); of step k only row k and column k travel, through two LDS buffers (one barrier per step).  6 nt dependent steps of ~0.1 us - the
// round-3 kernels (block pivots of 6 with the matrix in LDS, four barriers and a redundant 6 x 6 inverse per pivot block) took 50 us for a
// 48-row block and 390 us for 96 rows, which is why the top level stopped at 48 rows and the levels between were built by ten more launches.
template <int LW>
__device__ __forceinline__ void gj_tiles(double (&a)[36], int I, int J, int nt, double* __restrict__ sRow, double* __restrict__ sCol)
{
    for (int K = 0; K < nt; K++) {
#pragma unroll
        for (int kk = 0; kk < 6; kk++) {
            double* __restrict__ rb = sRow + (kk & 1) * 6 * LW;            // (6 K + kk) & 1 = kk & 1
            double* __restrict__ cb = sCol + (kk & 1) * 6 * LW;
            if (I == K) {
#pragma unroll
                for (int c = 0; c < 6; c++) rb[6 * J + c] = a[kk * 6 + c];
            }
            if (J == K) {
#pragma unroll
                for (int r = 0; r < 6; r++) cb[6 * I + r] = a[r * 6 + kk];
            }
            __syncthreads();
            const double p = 1. / rb[6 * K + kk];
            double rk[6], m[6];
#pragma unroll
            for (int c = 0; c < 6; c++) rk[c] = rb[6 * J + c];
#pragma unroll
            for (int r = 0; r < 6; r++) m[r] = cb[6 * I + r] * p;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++) a[r * 6 + c] = fma(-m[r], rk[c], a[r * 6 + c]);
            if (I == K) {
#pragma unroll
                for (int c = 0; c < 6; c++) a[kk * 6 + c] = rk[c] * p;
            }
            if (J == K) {
#pragma unroll
                for (int r = 0; r < 6; r++) a[r * 6 + kk] = -m[r];
            }
            if (I == K && J == K) a[kk * 6 + kk] = p;
        }
    }
}

//      Sibling blocks: one WAVE per (level l < L, aggregate A of level l+1), four to a workgroup.  A_l(lambda) restricted to A's children
//      ((6 fan)^2 <= 48^2: diagonal blocks plus every off-diagonal block whose column is a sibling; multi-edges add up in slot order) is
//      gathered tile by tile - lane (I, J) = children (I, J) - inverted in registers and written to Winv[l][A].  Missing children (last
//      aggregate of a level, fan-out 4) are padded with identity rows: every wave takes the same 48 steps, so the workgroup's barriers
//      line up whatever its four aggregates are.
//      Top level: the LAST workgroup of the same launch (the two depend on nothing but the Galerkin products): <= kMlTopWide aggregates,
//      16 x 16 lanes, one tile each.
constexpr int kSibPerBlk = kBlk / 64;                   // aggregates per workgroup
constexpr int kSibCols = 1024;                          // slot columns of one aggregate staged in LDS
constexpr int kSibHits = 2;                             // blocks of one tile fetched in one round trip

__device__ __forceinline__ void ml_top_tiles(PgoDev D, const MlDev* __restrict__ mlp, double* __restrict__ sRow, double* __restrict__ sCol);

__device__ __forceinline__ void ml_inverses_kernel_body(PgoDev D, const MlDev* __restrict__ mlp)
{
    __shared__ double sRow[kSibPerBlk * 2 * 48], sCol[kSibPerBlk * 2 * 48];      // (the top level uses the first 2 x 96 of each)
    __shared__ int scol_all[kSibPerBlk * kSibCols];
    if (blockIdx.x == gridDim.x - 1) { ml_top_tiles(D, mlp, sRow, sCol); return; }
    const MlDev& ml = *mlp;
    const double lambda = D.scal[3];
    const int wv = threadIdx.x >> 6, t = threadIdx.x & 63, I = t >> 3, J = t & 7;
    int* __restrict__ scol = scol_all + wv * kSibCols;
    int A = blockIdx.x * kSibPerBlk + wv, l = 0;
    while (l < ml.levels && A >= ml.lv[l + 1].n) { A -= ml.lv[l + 1].n; l++; }
    const bool live = l < ml.levels;                    // (a batch launches the largest graph's grid; the last workgroup may be short)
    if (!live) { l = 0; A = 0; }
    const MlLevel& F = ml.lv[l];
    const int fan = ml.lv[l + 1].fan, m = 6 * fan, nc = F.n;
    // column indices of the aggregate's rows (one contiguous slot range): all loads in flight at once, so the
    // per-row scan below walks LDS instead of paying a memory round trip per slot (hub rows have dozens)
    const int cfirst = A * fan, clast = (cfirst + fan < nc) ? cfirst + fan : nc;
    const int sbeg = F.row_ptr[cfirst], send = live ? F.row_ptr[clast] : sbeg;
    for (int i = sbeg + t; i < send && i - sbeg < kSibCols; i += 64) scol[i - sbeg] = F.col[i];
    const int ci = cfirst + I, cj = cfirst + J;
    const bool act = live && I < fan && J < fan;
    double a[36];
#pragma unroll
    for (int k = 0; k < 36; k++) a[k] = 0.;
    int r0 = 0, r1 = 0;
    if (act && ci < nc && cj < nc && I != J && (l > 0 || D.sibling0)) { r0 = F.row_ptr[ci]; r1 = F.row_ptr[ci + 1]; }
    if (I == J) {
        if (!act || ci >= nc) {
#pragma unroll
            for (int k = 0; k < 6; k++) a[k * 7] = 1.;
        } else {
            double any = 0.;
#pragma unroll
            for (int k = 0; k < 36; k++) {
                double v = F.G[(size_t)ci * 36 + k];
                if (l == 0) { if (k % 7 == 0) v += lambda; } else v += lambda * F.M[(size_t)ci * 36 + k];
                a[k] = v; any = fmax(any, fabs(v));
            }
            if (any == 0.) {                             // an aggregate of empty rows only (strong-aggregate numbering): nothing to correct
#pragma unroll
                for (int k = 0; k < 6; k++) a[k * 7] = 1.;
            }
        }
    }
    __syncthreads();
    // the slots of row ci whose column is cj: the first kSibHits fetched together, in slot order
    int hit[kSibHits], nh = 0;
#pragma unroll
    for (int q = 0; q < kSibHits; q++) hit[q] = -1;
    int s = r0;
    for (; s < r1 && nh < kSibHits; s++) {
        const int cc = (s - sbeg < kSibCols) ? scol[s - sbeg] : F.col[s];
        if (cc == cj) {
#pragma unroll
            for (int q = 0; q < kSibHits; q++) if (q == nh) hit[q] = s;
            nh++;
        }
    }
    double hb[kSibHits][36];
#pragma unroll
    for (int q = 0; q < kSibHits; q++)
#pragma unroll
        for (int k = 0; k < 36; k++) hb[q][k] = (hit[q] >= 0) ? F.blk[(size_t)hit[q] * 36 + k] : 0.;
#pragma unroll
    for (int q = 0; q < kSibHits; q++)
#pragma unroll
        for (int k = 0; k < 36; k++) a[k] += hb[q][k];
    for (; s < r1; s++) {                                           // more parallel edges than that: one at a time
        const int cc = (s - sbeg < kSibCols) ? scol[s - sbeg] : F.col[s];
        if (cc == cj) {
#pragma unroll
            for (int k = 0; k < 36; k++) a[k] += F.blk[(size_t)s * 36 + k];
        }
    }
    gj_tiles<8>(a, I, J, 8, sRow + wv * 2 * 48, sCol + wv * 2 * 48);
    if (act) {
        double* __restrict__ out = F.Winv + (size_t)A * m * m + (size_t)(6 * I) * m + 6 * J;
#pragma unroll
        for (int k = 0; k < 36; k++) out[(k / 6) * m + k % 6] = a[k];
    }
}
__device__ __forceinline__ double prolong_comp(const double* __restrict__ d, const double* __restrict__ yp, int k);

// ---- composite path (small graphs), per LM trial, for l = L-1 .. 1: the whole hierarchy above level l as ONE dense
//      operator  Y_l = blockdiag(W_l^-1 over sibling groups) + P_{l+1} Y_{l+1} P_{l+1}^T   ((6 n_l)^2, Y_L = A_L^-1).
//      Y_1 is what ml_cg applies: the coarse correction of an aggregate is 6 rows of Y_1 times the gather-level
//      residual - one latency-flat dot product instead of a restrict / solve / prolong walk through LDS.
//      One lane per 6x6 block (B, B').
__device__ __forceinline__ void ml_dense_level_kernel_body(const MlDev* __restrict__ mlp, int l)
{
    const MlDev& ml = *mlp;
    const int n = ml.lv[l].n, t = blockIdx.x * kBlk + threadIdx.x;
    if (t >= n * n) return;
    const int B = t / n, Bp = t % n;
    const int fan = ml.lv[l + 1].fan, np6 = 6 * ml.lv[l + 1].n, m = 6 * fan;
    const double* __restrict__ Yp = (l + 1 == ml.levels) ? ml.top_inv : ml.Ydense[l + 1];
    const int pB = B / fan, pBp = Bp / fan;
    double Yb[36], T1[36];
#pragma unroll
    for (int i = 0; i < 36; i++) Yb[i] = Yp[(size_t)(6 * pB + i / 6) * np6 + 6 * pBp + i % 6];
    const double* dB = ml.lv[l].geo + (size_t)B * 3;
    const double* dBp = ml.lv[l].geo + (size_t)Bp * 3;
    const double d0[3] = {dB[0], dB[1], dB[2]}, d1[3] = {dBp[0], dBp[1], dBp[2]};
    // T1 = Yb P(B')^T : row r of T1 = P(B') applied to row r of Yb
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) T1[r * 6 + k] = prolong_comp(d1, Yb + r * 6, k);
    // out = P(B) T1 : column c of out = P(B) applied to column c of T1
    double out[36];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const double col[6] = {T1[c], T1[6 + c], T1[12 + c], T1[18 + c], T1[24 + c], T1[30 + c]};
#pragma unroll
        for (int k = 0; k < 6; k++) out[k * 6 + c] = prolong_comp(d0, col, k);
    }
    if (pB == pBp) {
        const double* __restrict__ W = ml.lv[l].Winv + (size_t)pB * m * m + (size_t)((B % fan) * 6) * m + (Bp % fan) * 6;
#pragma unroll
        for (int i = 0; i < 36; i++) out[i] += W[(i / 6) * m + i % 6];
    }
    double* __restrict__ Y = ml.Ydense[l];
    const int n6 = 6 * n;
#pragma unroll
    for (int i = 0; i < 36; i++) Y[(size_t)(6 * B + i / 6) * n6 + 6 * Bp + i % 6] = out[i];
}
__global__ __launch_bounds__(kBlk) void ml_dense_level_kernel(const MlDev* __restrict__ mlp, int l) { ml_dense_level_kernel_body(mlp, l); }

// ---- composite path, level 1: MULTIPLICATIVE coupling of the level-1 smoother with the levels above,
//          Y_1 = 2 S - S A S + Q Y_2 Q^T,   Q = P - S A P                     (S = blockdiag of the sibling inverses W_1^-1,
//      A = A_1(lambda), P = P_2, Y_2 = the dense operator of level 2): the symmetric pre-smooth / coarse-correct /
//      post-smooth cycle written as one matrix.  It costs ~0.25 GFLOP of 6x6 block products per rebuild and nothing per
//      iteration (ml_cg_comp applies whatever Y_1 holds), and takes a third off the PCG iteration count compared with the
//      additive S + P Y_2 P^T (numpy prototype: 41 -> 29 and 57 -> 38 on config 2).  ml_mult_pair_kernel forms the cycle's operands per pair of
//      sibling groups, ml_mult_qy_kernel and ml_mult_qyqt_kernel add the coarse term.
__device__ __forceinline__ void mr6_acc(const double* __restrict__ Arow, const double* __restrict__ B, int ldb, double* c6, double sgn)
{                                                                              // c6 += sgn * Arow(1x6) * B(6x6, ld ldb)
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < 6; k++) s += Arow[k] * B[k * ldb + c];
        c6[c] += sgn * s;
    }
}

// QY[i][p] = sum_p' Q[i][p'] Y_2[p'][p]                                           (one lane per block row)
__device__ __forceinline__ void ml_mult_qy_kernel_body(const MlDev* __restrict__ mlp, int cl)
{
    const MlDev& ml = *mlp;
    const int n = ml.lv[cl].n, np = ml.lv[cl + 1].n, np6 = 6 * np;
    const int tt = blockIdx.x * kBlk + threadIdx.x;
    if (tt >= n * np * 6) return;
    const int t = tt / 6, r = tt % 6;
    const int i = t / np, p = t % np;
    const double* __restrict__ Y2 = (ml.levels == cl + 1) ? ml.top_inv : ml.Ydense[cl + 1];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int pp = 0; pp < np; pp++) mr6_acc(ml.mQ + ((size_t)i * np + pp) * 36 + r * 6, Y2 + (size_t)(6 * pp) * np6 + 6 * p, np6, acc, 1.);
    double* o = ml.mQY + (size_t)t * 36 + r * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) o[c] = acc[c];
}
__global__ __launch_bounds__(kBlk) void ml_mult_qy_kernel(const MlDev* __restrict__ mlp, int cl)
{
    ml_mult_qy_kernel_body(mlp, cl);
}

// ---- the cycle's operands per PAIR of sibling groups, in one launch (round 4; was A P | A S, then Q | 2 S - S (A S): two launches of
//      one lane per 6 x 6 block, 57 us at config 2).  Workgroup (g, p): the block A_gp of A_l(lambda) between the children of level-(l+1)
//      aggregates g and p (48 x 48, assembled in LDS from the slot ranges grp_beg / grp_end), the sibling inverses S_g and S_p, and
//          Y_l[g, p]  = [g == p] 2 S_g - (S_g A_gp) S_p                  (the cycle without its coarse term, which ml_mult_qyqt adds)
//          Q[g][p]    = [g == p] P_p  - S_g (A_gp P_p)                    (48 x 6; P_p = the children's prolongation blocks stacked)
//      Pairs without a block between them (and g != p) only write zeros.  Dense products of 48 x 48 LDS tiles, 256 lanes.
constexpr int kPairLd = 49;
__device__ __forceinline__ void ml_mult_pair_kernel_body(PgoDev D, const MlDev* __restrict__ mlp, int cl)
{
    __shared__ double sA[48 * kPairLd], sSg[48 * kPairLd], sSp[48 * kPairLd], sT[48 * kPairLd];
    __shared__ double sPp[48 * 6], sAP[48 * 6];
    __shared__ int s_any;
    const MlDev& ml = *mlp;
    const MlLevel& F = ml.lv[cl];
    const int n = F.n, np = ml.lv[cl + 1].n, fan = ml.lv[cl + 1].fan, m = 6 * fan, n6 = 6 * n;
    const int g = blockIdx.x / np, p = blockIdx.x % np, tid = threadIdx.x;
    if ((int)blockIdx.x >= np * np) return;
    const double lambda = D.scal[3];
    const int* __restrict__ gb = ml.grp_beg[cl];
    const int* __restrict__ ge = ml.grp_end[cl];
    if (tid == 0) s_any = (g == p) ? 1 : 0;
    __syncthreads();
    if (tid < fan) {
        const int i = g * fan + tid;
        if (i < n && gb[(size_t)i * np + p] < ge[(size_t)i * np + p]) s_any = 1;
    }
    __syncthreads();
    double* __restrict__ Y = ml.Ydense[cl];
    if (!s_any) {                                                        // no coupling: Y tile and Q blocks are zero
        for (int e = tid; e < m * m; e += kBlk) {
            const int r = e / m, c = e % m, gi = g * fan + r / 6, pi = p * fan + c / 6;
            if (gi < n && pi < n) Y[(size_t)(6 * gi + r % 6) * n6 + 6 * pi + c % 6] = 0.;
        }
        for (int e = tid; e < fan * 36; e += kBlk) {
            const int i = g * fan + e / 36;
            if (i < n) ml.mQ[((size_t)i * np + p) * 36 + e % 36] = 0.;
        }
        return;
    }
    // ---- operands into LDS
    for (int e = tid; e < m * m; e += kBlk) {
        const int r = e / m, c = e % m;
        sSg[r * kPairLd + c] = F.Winv[(size_t)g * m * m + e];
        sSp[r * kPairLd + c] = F.Winv[(size_t)p * m * m + e];
        sA[r * kPairLd + c] = 0.;
    }
    for (int e = tid; e < fan * 36; e += kBlk) {                         // P_p: pmat6 of every child of p (zero rows for missing children)
        const int j = e / 36, k = e % 36, c = p * fan + j;
        double v = 0.;
        if (c < n) {
            const double* __restrict__ d = F.geo + (size_t)c * 3;
            const int r = k / 6, q = k % 6;
            v = (r == q) ? 1. : 0.;
            if (r == 0 && q == 4) v = d[2];  if (r == 0 && q == 5) v = -d[1];
            if (r == 1 && q == 3) v = -d[2]; if (r == 1 && q == 5) v = d[0];
            if (r == 2 && q == 3) v = d[1];  if (r == 2 && q == 4) v = -d[0];
        }
        sPp[(j * 6 + k / 6) * 6 + k % 6] = v;
    }
    __syncthreads();
    if (g == p) {                                                        // diagonal blocks D_i = G_i + lambda M_i
        for (int e = tid; e < fan * 36; e += kBlk) {
            const int j = e / 36, k = e % 36, i = g * fan + j;
            if (i < n) sA[(j * 6 + k / 6) * kPairLd + j * 6 + k % 6] = F.G[(size_t)i * 36 + k] + lambda * F.M[(size_t)i * 36 + k];
        }
    }
    for (int j = 0; j < fan; j++) {                                      // off-diagonal blocks of row i whose column is a child of p
        const int i = g * fan + j;
        if (i >= n) break;
        const int s0 = gb[(size_t)i * np + p], s1 = ge[(size_t)i * np + p];
        for (int e = tid; e < (s1 - s0) * 36; e += kBlk) {
            const int s_ = s0 + e / 36, k = e % 36, jc = F.col[s_] - p * fan;
            sA[(j * 6 + k / 6) * kPairLd + jc * 6 + k % 6] = F.blk[(size_t)s_ * 36 + k];       // (blocks of a coarse level are unique per (row, column))
        }
    }
    __syncthreads();
    // ---- T = S_g A, AP = A P_p.  The two m x m x m products in 3 x 3 register tiles: lane (tr, tc) of (m / 3)^2 <= 256
    const int m3 = m / 3, tr = tid / m3, tc = tid % m3;
    const bool tile = tid < m3 * m3;
    if (tile) {
        double t[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int k = 0; k < m; k++) {
            const double a0 = sSg[(3 * tr) * kPairLd + k], a1 = sSg[(3 * tr + 1) * kPairLd + k], a2 = sSg[(3 * tr + 2) * kPairLd + k];
            const double b0 = sA[k * kPairLd + 3 * tc], b1 = sA[k * kPairLd + 3 * tc + 1], b2 = sA[k * kPairLd + 3 * tc + 2];
            t[0][0] = fma(a0, b0, t[0][0]); t[0][1] = fma(a0, b1, t[0][1]); t[0][2] = fma(a0, b2, t[0][2]);
            t[1][0] = fma(a1, b0, t[1][0]); t[1][1] = fma(a1, b1, t[1][1]); t[1][2] = fma(a1, b2, t[1][2]);
            t[2][0] = fma(a2, b0, t[2][0]); t[2][1] = fma(a2, b1, t[2][1]); t[2][2] = fma(a2, b2, t[2][2]);
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) sT[(3 * tr + i) * kPairLd + 3 * tc + j] = t[i][j];
    }
    for (int e = tid; e < m * 6; e += kBlk) {
        const int r = e / 6, c = e % 6;
        double t = 0.;
        for (int k = 0; k < m; k++) t = fma(sA[r * kPairLd + k], sPp[k * 6 + c], t);
        sAP[e] = t;
    }
    __syncthreads();
    // ---- Y tile = [g == p] 2 S_g - T S_p;  Q = [g == p] P_p - S_g AP
    if (tile) {
        double t[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int k = 0; k < m; k++) {
            const double a0 = sT[(3 * tr) * kPairLd + k], a1 = sT[(3 * tr + 1) * kPairLd + k], a2 = sT[(3 * tr + 2) * kPairLd + k];
            const double b0 = sSp[k * kPairLd + 3 * tc], b1 = sSp[k * kPairLd + 3 * tc + 1], b2 = sSp[k * kPairLd + 3 * tc + 2];
            t[0][0] = fma(a0, b0, t[0][0]); t[0][1] = fma(a0, b1, t[0][1]); t[0][2] = fma(a0, b2, t[0][2]);
            t[1][0] = fma(a1, b0, t[1][0]); t[1][1] = fma(a1, b1, t[1][1]); t[1][2] = fma(a1, b2, t[1][2]);
            t[2][0] = fma(a2, b0, t[2][0]); t[2][1] = fma(a2, b1, t[2][1]); t[2][2] = fma(a2, b2, t[2][2]);
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int r = 3 * tr + i, c = 3 * tc + j, gi = g * fan + r / 6, pi = p * fan + c / 6;
                if (gi < n && pi < n) Y[(size_t)(6 * gi + r % 6) * n6 + 6 * pi + c % 6] = ((g == p) ? 2. * sSg[r * kPairLd + c] : 0.) - t[i][j];
            }
    }
    for (int e = tid; e < m * 6; e += kBlk) {
        const int r = e / 6, c = e % 6, i = g * fan + r / 6;
        if (i >= n) continue;
        double t = 0.;
        for (int k = 0; k < m; k++) t = fma(sSg[r * kPairLd + k], sAP[k * 6 + c], t);
        ml.mQ[((size_t)i * np + p) * 36 + (r % 6) * 6 + c] = ((g == p) ? sPp[r * 6 + c] : 0.) - t;
    }
}
__global__ __launch_bounds__(kBlk) void ml_mult_pair_kernel(PgoDev D, const MlDev* __restrict__ mlp, int cl)
{
    ml_mult_pair_kernel_body(D, mlp, cl);
}

// ---- composite path: Newton-Schulz refinement  X <- 2 X - X A_1 X  of the dense level-1 operator (X = Y_1 is already a
//      good approximate inverse of A_1: one step squares the error of the cycle, two make it exact to PCG's eyes).
//      ml_ns_ax_kernel:   T = A_1 X        block-sparse (6x6 blocks) times dense, one lane per (row block, column)
//      ml_ns_gemm_kernel: X' = 2 X - X T   dense f64 GEMM on the matrix cores (2 n^3 flops, n = 6 n_1 <= 960)
// one workgroup per (row block i, <= 256 columns): the row's 6x6 blocks go through LDS once (broadcast reads), every lane owns
// one column of X and walks the row's neighbours (X rows are read coalesced across the lanes).
// XCD-aware: workgroup b runs on XCD b % 8, and X - written by the previous kernel on all eight - comes out of the Infinity Cache into
// that XCD's L2.  With the column range as the slow index every XCD pulled ALL of X (8 x 4.5 MB at n = 750: the kernel's 20 us);
// with columns split into eight slabs, slab b % 8, every XCD pulls one eighth.
constexpr int kAxChunk = 16;          // off-diagonal blocks staged per pass
constexpr int kXcds = 8;
__host__ __device__ __forceinline__ int ax_slab(int n6) { return ((n6 + kXcds - 1) / kXcds + 31) & ~31; }          // columns per XCD
__host__ __device__ __forceinline__ int ax_parts(int n6) { return (ax_slab(n6) + kBlk - 1) / kBlk; }             // workgroups per (row block, slab)
// (a batch fills the chip anyway and its graphs' X live in different places: there the plain mapping - row block, 256 columns per
//  workgroup, every lane busy - is the faster one: `by_xcd` = false)
__device__ __forceinline__ void ml_ns_ax_kernel_body(PgoDev D, const MlDev* __restrict__ mlp, int cl, const double* __restrict__ X,
                                                       double* __restrict__ T, bool by_xcd = true)
{
    __shared__ double sb[(kAxChunk + 1) * 36];
    __shared__ int sc[kAxChunk + 1];
    const MlDev& ml = *mlp;
    const MlLevel& F = ml.lv[cl];
    const int n6 = 6 * F.n, slab = ax_slab(n6), parts = ax_parts(n6), tid = threadIdx.x;
    int i, c;
    bool act;
    if (by_xcd) {
        const int xcd = blockIdx.x % kXcds, rest = blockIdx.x / kXcds, part = rest % parts, cin = part * kBlk + tid;
        i = rest / parts; c = xcd * slab + cin; act = cin < slab && c < n6;
    } else {
        const int chunks = (n6 + kBlk - 1) / kBlk;
        i = blockIdx.x / chunks; c = (blockIdx.x % chunks) * kBlk + tid; act = c < n6;
    }
    if (i >= F.n) return;                                          // (a batch launches the largest graph's grid)
    const double lambda = D.scal[3];
    const int s0 = F.row_ptr[i], s1 = F.row_ptr[i + 1];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int base = s0 - 1; base < s1; base += kAxChunk + 1) {          // "slot" s0 - 1 stands for the diagonal block
        const int cnt = min(kAxChunk + 1, s1 - base);
        __syncthreads();
        for (int e = tid; e < cnt * 36; e += kBlk) {
            const int q = e / 36, k = e % 36, s = base + q;
            sb[e] = (s < s0) ? F.G[(size_t)i * 36 + k] + lambda * F.M[(size_t)i * 36 + k] : F.blk[(size_t)s * 36 + k];
        }
        if (tid < cnt) sc[tid] = (base + tid < s0) ? i : F.col[base + tid];
        __syncthreads();
        if (act) {
            for (int q = 0; q < cnt; q++) {
                const int j = sc[q];
                const double* __restrict__ bq = sb + q * 36;
                double x[6];
#pragma unroll
                for (int k = 0; k < 6; k++) x[k] = X[(size_t)(6 * j + k) * n6 + c];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int k = 0; k < 6; k++) acc[r] += bq[r * 6 + k] * x[k];
            }
        }
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 6; r++) T[(size_t)(6 * i + r) * n6 + c] = acc[r];
    }
}
__global__ __launch_bounds__(kBlk) void ml_ns_ax_kernel(PgoDev D, const MlDev* __restrict__ mlp, int cl, const double* __restrict__ X, double* __restrict__ T)
{
    ml_ns_ax_kernel_body(D, mlp, cl, X, T);
}

constexpr int kGemmTile = 64, kGemmK = 64;
typedef double v4f64 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ int gemm_slabs_per_quarter(int n) { const int slabs = (n + kGemmK - 1) / kGemmK; return (slabs + 3) / 4; }
// tile (ti <= tj) number `b` of the gt (gt + 1) / 2 tiles on and above the diagonal, counted row by row
__device__ __forceinline__ void tri_tile(int gt, int b, int& ti, int& tj)
{
    int t = (int)(((double)(2 * gt + 1) - sqrt((double)(2 * gt + 1) * (double)(2 * gt + 1) - 8. * (double)b)) * 0.5);
    t = t < 0 ? 0 : (t > gt - 1 ? gt - 1 : t);
    while (t > 0 && t * gt - t * (t - 1) / 2 > b) t--;                       // first tile of row t: t gt - t (t - 1) / 2
    while (t + 1 < gt && (t + 1) * gt - (t + 1) * t / 2 <= b) t++;
    ti = t; tj = t + (b - (t * gt - t * (t - 1) / 2));
}
// The same tiles in an XCD-aware order.  Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2, and a tile streams
// its row slab of X and its column slab of T (2 x 64 x n doubles) once: with the tiles dealt out row by row every XCD touched every slab
// (round 4: 501 MB of L2 fills per launch at n = 1878 against 85 MB of operands).  Here the tiles are listed super-block by super-block
// (S x S tiles, S ~ the side of a square holding one XCD's share) and XCD x takes a contiguous run of that list: its workgroups, which
// walk K side by side, share ~S row slabs and ~S column slabs instead of ~T / 8 of each.  A permutation of the tile numbers: no entry
// of the result changes.
__device__ __forceinline__ void tri_tile_xcd(int gt, int b, int& ti, int& tj)
{
    const int T = gt * (gt + 1) / 2;
    const int x = b % kXcds;
    int p = b / kXcds;
    for (int y = 0; y < x; y++) p += (T - y + kXcds - 1) / kXcds;          // blocks y, y + 8, ... < T run on XCD y
    int S = (int)ceilf(sqrtf((float)T / (float)kXcds));
    S = S < 2 ? 2 : (S > 16 ? 16 : S);
    const int gs = (gt + S - 1) / S;
    for (int I = 0; I < gs; I++) {
        const int rI = min(S, gt - I * S);
        for (int J = I; J < gs; J++) {
            const int rJ = min(S, gt - J * S);
            const int cnt = (I == J) ? rI * (rI + 1) / 2 : rI * rJ;
            if (p < cnt) {
                if (I == J) { int a, c; tri_tile(rI, p, a, c); ti = I * S + a; tj = J * S + c; }
                else { ti = I * S + p / rJ; tj = J * S + p % rJ; }
                return;
            }
            p -= cnt;
        }
    }
    ti = tj = gt - 1;                                                       // (not reached: the counts add up to T)
}
// X' = 2 X - X T on the f64 matrix cores: v_mfma_f64_16x16x4_f64 (lane l feeds A[row l&15][k l>>4] and B[k l>>4][col l&15];
// the four results of a lane are C[row (l>>4) + 4 r][col l&15], r = 0..3).  A 256-lane workgroup owns a 64 x 64 tile, each
// of its four waves a 32 x 32 quarter as 2 x 2 MFMA tiles; K is staged through LDS in slabs of 64 (coalesced global
// reads; one LDS double per MFMA operand, padded rows: no bank pile-up); the next slab is fetched into registers
// while the matrix cores work on the current one.  One workgroup per CU at this size (144 tiles at n = 750), so nothing
// but the slab's own MFMAs hides the global-load latency of the next slab: with slabs of 16 (47 dependent slabs, 0.2 us
// of MFMA each) about 1 us per slab stayed exposed (55 us); slabs of 64 leave 12 exposures.
// TILE: 0 = a tile inside the matrix (bounds-free loads for whole slabs), 1 = a tile on the matrix edge, n even (clamped loads),
// 2 = n odd (element-wise: diagnostic sizes only, n = 6 n_c is even).  One instantiation per class, chosen per workgroup: with the
// three kinds of load behind branches of ONE loop the interior tiles lost 4 - 15 % (n = 3750: 1138 -> 1340 us).
template <int TILE>
__device__ __forceinline__ void ml_ns_gemm_tile(int n, int ti, int tj, const double* __restrict__ X, const double* __restrict__ T,
                                                double* __restrict__ Xn, float* __restrict__ c32, int c32_stride,
                                                double (*__restrict__ sA)[kGemmTile], double (*__restrict__ sB)[kGemmTile])
{
    // Both operand tiles sit k-major in LDS, s[k][i ^ 16 (k & 1)]: the 16 x 4 (row or column, k) doubles one MFMA operand read takes
    // then fall into 64 different banks per half wave (rows padded to 65 doubles put (i, k) and (i + 1, k - 1) on the same bank: 2- to
    // 4-way conflicts on every read).  The X tile is read through X's symmetry (X[k][row], lanes along the row: contiguous in memory
    // and in LDS) - X is symmetric in the refinement.  Measured: no faster than the padded layout (1.32 ms at n = 3750, 40 TFLOP/s) -
    // with 64 x 64 tiles the kernel moves 8 flops per operand byte, 5.4 TB/s out of the Infinity Cache at that rate: the operand
    // stream, not the LDS and not the matrix pipe, sets the pace (slabs of 32 with four workgroups per CU: 1.21 ms, and slower at n = 750).
    // (sA[k][row ^ swz(k)] = X[row][k], sB[k][col ^ swz(k)] = T[k][col]: the caller's LDS, shared by the three instantiations)
    constexpr int kPer = kGemmTile * kGemmK / 256;     // values per lane and operand per slab
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // X, A and therefore X (A X) are symmetric: only the tiles on and above the diagonal are computed (blockIdx.x counts them row by
    // row), every result is stored twice.  Half the flops of the rebuild's dominant kernel, and X' is symmetric to the last bit
    // outside the diagonal tiles - which PCG wants from its preconditioner anyway.
    const int gt = (n + kGemmTile - 1) / kGemmTile;
    const int row0 = ti * kGemmTile, col0 = tj * kGemmTile;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;          // this wave's quarter
    const int li = lane & 15, lk = lane >> 4;
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = v4f64{0., 0., 0., 0.};
    // A lane fetches PAIRS of neighbouring rows / columns (16-byte loads; n = 6 n_c is even and the tiles start at multiples of 64, so a
    // pair is aligned and never straddles the matrix edge); slabs and tiles that lie inside the matrix - all but the last of each - load
    // without a bounds test (round 4's element-wise predicated loads were 32 branches per slab and operand).
    double2 pa[kPer / 2], pb[kPer / 2];
    // Operand fetch: 16-byte loads (a pair of neighbouring rows / columns; n = 6 n_c is even and tiles start at multiples of 64, so a
    // pair is aligned and lies inside or outside the matrix as a whole).  On the matrix edge (TILE 1) and in an interior tile's last,
    // partial slab the address of a pair outside the matrix is CLAMPED to the matrix's first element and the value replaced by zero.
    // Rounds 4-5 took 32 predicated 8-byte loads there, each in a branch of its own with a wait behind it - and since a 10k-vertex
    // graph's 465 tiles are two to a CU, the launch lasted as long as its slowest EDGE tile (n = 1878, 59 edge tiles: 189 us against
    // 160 us at n = 1920, which has none; now 165 us; tests/diag/ns_gemm_lab.hip).
    auto fetch = [&](int k0) {
        if (TILE == 0 && k0 + kGemmK <= n) {
#pragma unroll
            for (int u = 0; u < kPer / 2; u++) {
                const int e = u * 256 + tid;
                const int ek = e / (kGemmTile / 2), ei = 2 * (e % (kGemmTile / 2));      // consecutive lanes walk the row / column index
                pa[u] = *reinterpret_cast<const double2*>(X + (size_t)(k0 + ek) * n + row0 + ei);      // = X[row][k], X symmetric
                pb[u] = *reinterpret_cast<const double2*>(T + (size_t)(k0 + ek) * n + col0 + ei);
            }
        } else if (TILE != 2) {
#pragma unroll
            for (int u = 0; u < kPer / 2; u++) {
                const int e = u * 256 + tid;
                const int ek = e / (kGemmTile / 2), ei = 2 * (e % (kGemmTile / 2));
                const int gk = k0 + ek, gr = row0 + ei, gc = col0 + ei;
                const bool oka = gk < n && gr < n, okb = gk < n && gc < n;
                const double2 va = *reinterpret_cast<const double2*>(X + (oka ? (size_t)gk * n + gr : 0));
                const double2 vb = *reinterpret_cast<const double2*>(T + (okb ? (size_t)gk * n + gc : 0));
                pa[u] = oka ? va : make_double2(0., 0.);
                pb[u] = okb ? vb : make_double2(0., 0.);
            }
        } else {
#pragma unroll
            for (int u = 0; u < kPer / 2; u++) {
                const int e = u * 256 + tid;
                const int ek = e / (kGemmTile / 2), ei = 2 * (e % (kGemmTile / 2));
                const int gk = k0 + ek, gr = row0 + ei, gc = col0 + ei;
                pa[u].x = (gk < n && gr < n) ? X[(size_t)gk * n + gr] : 0.;
                pa[u].y = (gk < n && gr + 1 < n) ? X[(size_t)gk * n + gr + 1] : 0.;
                pb[u].x = (gk < n && gc < n) ? T[(size_t)gk * n + gc] : 0.;
                pb[u].y = (gk < n && gc + 1 < n) ? T[(size_t)gk * n + gc + 1] : 0.;
            }
        }
    };
    fetch(0);
    const int sw = (lk & 1) << 4;                       // k4 is a multiple of 4: (k4 + lk) & 1 = lk & 1
    // ONE summation order for this kernel and ml_ns_gemm32_kernel (small n, one graph): K in four quarters of whole slabs, each summed on
    // its own, then ((q0 + q1) + q2) + q3 - the 32 x 32 kernel gives a quarter to each of its four waves.  Same bits from both, so a
    // graph solved alone and in a batch (which keeps this kernel) agree.
    const int spq = gemm_slabs_per_quarter(n);
    v4f64 tot[2][2];
    int slab = 0;
    for (int k0 = 0; k0 < n; k0 += kGemmK, slab++) {
#pragma unroll
        for (int u = 0; u < kPer / 2; u++) {
            const int e = u * 256 + tid;
            const int ek = e / (kGemmTile / 2), ei = (2 * (e % (kGemmTile / 2))) ^ ((ek & 1) << 4);      // (the swizzle moves pairs as pairs)
            *reinterpret_cast<double2*>(&sA[ek][ei]) = pa[u];
            *reinterpret_cast<double2*>(&sB[ek][ei]) = pb[u];
        }
        __syncthreads();
        if (k0 + kGemmK < n) fetch(k0 + kGemmK);
        // (Measured and not kept: the next pair of k-steps' operands read from LDS ahead of this pair's MFMAs, the order pinned with
        //  __builtin_amdgcn_sched_group_barrier - eight LDS reads in flight per wave instead of four: 200 -> 253 us at n = 1878, 1253 -> 1523 us
        //  at n = 3750.  With two workgroups per CU the other wave's MFMAs already cover a wave's LDS round trip.)
#pragma unroll
        for (int k4 = 0; k4 < kGemmK; k4 += 4) {
            const double a0 = sA[k4 + lk][(wr + li) ^ sw], a1 = sA[k4 + lk][(wr + 16 + li) ^ sw];
            const double b0 = sB[k4 + lk][(wc + li) ^ sw], b1 = sB[k4 + lk][(wc + 16 + li) ^ sw];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if ((slab + 1) % spq == 0 || k0 + kGemmK >= n) {          // a quarter is complete
            const bool first = slab < spq;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) { tot[a][b] = first ? acc[a][b] : tot[a][b] + acc[a][b]; acc[a][b] = v4f64{0., 0., 0., 0.}; }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = tot[a][b];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gr = row0 + wr + a * 16 + lk + 4 * r, gc = col0 + wc + b * 16 + li;
                // (in a diagonal tile the quarter BELOW the diagonal is the mirror image of the one above - what the 32 x 32 tiling of
                //  ml_ns_gemm32_kernel stores there: the two kernels agree entry for entry)
                if (gr < n && gc < n && !(ti == tj && wr > wc)) {
                    const bool mirror = ti != tj || wr < wc;
                    const double v = 2. * X[(size_t)gr * n + gc] - acc[a][b][r];
                    Xn[(size_t)gr * n + gc] = v;
                    if (mirror) Xn[(size_t)gc * n + gr] = v;
                    if (c32) {                                                   // last step of a rebuild: the f32 copy the PCG kernels read (ml_cmat32_body)
                        c32[(size_t)gr * c32_stride + gc] = (float)v;
                        if (mirror) c32[(size_t)gc * c32_stride + gr] = (float)v;
                    }
                }
            }
    if (c32 && tj == gt - 1 && tid < kGemmTile && row0 + tid < n)               // pad columns [n, stride) stay zero
        for (int q = n; q < c32_stride; q++) c32[(size_t)(row0 + tid) * c32_stride + q] = 0.f;
}
__device__ __forceinline__ void ml_ns_gemm_kernel_body(int n, const double* __restrict__ X, const double* __restrict__ T,
                                                        double* __restrict__ Xn, float* __restrict__ c32 = nullptr, int c32_stride = 0)
{
    const int gt = (n + kGemmTile - 1) / kGemmTile;
    __shared__ double sA[kGemmK][kGemmTile];
    __shared__ double sB[kGemmK][kGemmTile];
    int ti, tj;
    tri_tile_xcd(gt, (int)blockIdx.x, ti, tj);
    if (n & 1) ml_ns_gemm_tile<2>(n, ti, tj, X, T, Xn, c32, c32_stride, sA, sB);                       // (uniform in the workgroup)
    else if ((tj + 1) * kGemmTile <= n) ml_ns_gemm_tile<0>(n, ti, tj, X, T, Xn, c32, c32_stride, sA, sB);      // (ti <= tj: the rows are inside too)
    else ml_ns_gemm_tile<1>(n, ti, tj, X, T, Xn, c32, c32_stride, sA, sB);
}
__global__ __launch_bounds__(256) void ml_ns_gemm_kernel(int n, const double* __restrict__ X, const double* __restrict__ T, double* __restrict__ Xn,
                                                        float* __restrict__ c32, int c32_stride)
{
    ml_ns_gemm_kernel_body(n, X, T, Xn, c32, c32_stride);
}

// The same product for ONE small graph (n <= kGemm32Max): 78 tiles of 64 x 64 leave two thirds of the chip idle at n = 750 and every
// workgroup walks all of K alone (53 us = 0.11 of the f64 matrix-core peak).  Here a workgroup owns a 32 x 32 tile and its four waves a
// QUARTER OF K each (the quarters of ml_ns_gemm_kernel_body, summed in the same order: same bits): 300 workgroups at n = 750, a quarter
// of the dependent steps per wave.  Operands go straight from global memory to the MFMA registers (lane (li, lk) of a 16 x 4 operand reads
// X[k + lk][row + li]: 128-byte runs), kGemm32Ahead steps ahead; the partial tiles meet in LDS.  Tiles are dealt to the XCDs by tile
// column, serpentine, so that an XCD's L2 holds its few columns of T and streams X.
constexpr int kGemm32Max = 960, kGemm32Ahead = 8;
__device__ __forceinline__ bool gemm32_tile(int gt, int b, int& ti, int& tj)
{
    const int x = b % kXcds;
    int idx = b / kXcds;
    for (int m = 0; ; m++) {
        const int c = (m & 1) ? 8 * m + 7 - x : 8 * m + x;             // columns of XCD x: x, 15 - x, 16 + x, 31 - x, ...
        if (8 * m >= gt) return false;
        if (c >= gt) continue;
        if (idx <= c) { ti = idx; tj = c; return true; }
        idx -= c + 1;
    }
}
__host__ __device__ __forceinline__ int gemm32_grid(int gt)      // workgroups: 8 x the most tiles any XCD gets
{
    int most = 0;
    for (int x = 0; x < kXcds; x++) {
        int cnt = 0;
        for (int m = 0; 8 * m < gt; m++) { const int c = (m & 1) ? 8 * m + 7 - x : 8 * m + x; if (c < gt) cnt += c + 1; }
        most = cnt > most ? cnt : most;
    }
    return kXcds * most;
}
// EDGE: the tile touches the matrix edge (rows or columns beyond n).  Operands beyond the matrix - an edge tile's, and every tile's k
// beyond n in the zero-padded last slab - come through BUFFER loads over the matrix (`buffer_load_dwordx2`: an offset beyond the
// resource's n^2 doubles returns zero, and a lane whose row or column lies beyond n is given such an offset): no branch, so the
// kGemm32Ahead steps of loads really are in flight together.  (Rounds 4-5 predicated every load: each sat in a branch of its own and
// the compiler put a full `s_waitcnt vmcnt(0)` in front of every step's MFMAs - the read-ahead never happened; address selects in front
// of plain loads were turned back into branches.)
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
template <bool EDGE>
__device__ __forceinline__ void ml_ns_gemm32_tile(int n, int ti, int tj, const double* __restrict__ X, const double* __restrict__ T,
                                                  double* __restrict__ Xn, float* __restrict__ c32, int c32_stride, double (*__restrict__ sP)[32][33])
{
    const int gt = (n + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar: the wave's K range and the branches on it are uniform)
    const int row0 = ti * 32, col0 = tj * 32;
    const int li = lane & 15, lk = lane >> 4;
    const int spq = gemm_slabs_per_quarter(n), slabs = (n + kGemmK - 1) / kGemmK;
    const int kbeg = wv * spq * kGemmK, kend = min((wv + 1) * spq, slabs) * kGemmK;          // whole slabs, zero-padded like the 64 x 64 kernel's
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = v4f64{0., 0., 0., 0.};
    const bool ra0 = !EDGE || row0 + li < n, ra1 = !EDGE || row0 + 16 + li < n, cb0 = !EDGE || col0 + li < n, cb1 = !EDGE || col0 + 16 + li < n;
    // (word 3 of the resource: raw buffer, 32-bit data format - the value the gfx90a / gfx942 / gfx950 family takes)
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(X), 0, n * n * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T), 0, n * n * 8, 0x00020000);
    double pa0[kGemm32Ahead], pa1[kGemm32Ahead], pb0[kGemm32Ahead], pb1[kGemm32Ahead];
    // (one kind of load for every step - a branch between plain and buffer loads inside the loop made the compiler wait for ALL loads in
    //  flight at every step: its count of outstanding loads does not survive a join)
    auto fetch = [&](int u, int k4) {
        const int k = k4 + lk;
        constexpr unsigned kOut = 0xFFFFFFF8u;                                // beyond any matrix: reads as zero
        const unsigned o = (k < n) ? (unsigned)k * (unsigned)n * 8u : kOut;     // (n <= 960: the matrix is < 8 MB)
        const unsigned oa = o + (unsigned)(row0 + li) * 8u, ob = o + (unsigned)(col0 + li) * 8u;
        pa0[u] = buf_load_f64(rX, (o != kOut && ra0) ? oa : kOut); pa1[u] = buf_load_f64(rX, (o != kOut && ra1) ? oa + 128u : kOut);      // = X[row][k] through X's symmetry
        pb0[u] = buf_load_f64(rT, (o != kOut && cb0) ? ob : kOut); pb1[u] = buf_load_f64(rT, (o != kOut && cb1) ? ob + 128u : kOut);
    };
#pragma unroll
    for (int u = 0; u < kGemm32Ahead; u++) fetch(u, kbeg + 4 * u);
    // the tile's own entries of X (for 2 X - X T at the end) are asked for now: behind the K range they were one more round trip
    double xown[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = tid + 256 * u, r = e >> 5, c = e & 31, gr = row0 + r, gc = col0 + c;
        xown[u] = buf_load_f64(rX, (gr < n && gc < n) ? ((unsigned)gr * (unsigned)n + (unsigned)gc) * 8u : 0xFFFFFFF8u);
    }
    // A wave's K range is whole 64-slabs: spq <= 4 of them (n <= kGemm32Max), i.e. 16 spq steps - unrolled COMPLETELY, one
    // instantiation per spq.  As a loop of 8-step trips the compiler drained all 32 loads in flight at every trip's head (its count of
    // loads in flight does not survive the loop's back edge); straight-line, every step waits for its own four loads and nothing else.
    // No test per step either (a wave whose range ends early - the last quarter - multiplies zeros: buffer loads beyond the matrix
    // read as zero, and the read-ahead past the range fetches operands nobody uses).
    static_assert(kGemm32Max <= 960 + 64, "spq <= 4");
    const int steps = (kend > kbeg) ? (kend - kbeg) / 4 : 0;
    auto run = [&](auto nsteps_c) {
        constexpr int NS = decltype(nsteps_c)::value;
#pragma unroll
        for (int st = 0; st < NS; st++) {
            const int u = st % kGemm32Ahead;
            // the step's MFMAs FIRST, then the fetch into the registers they have just read (with the fetch in front the old operands
            // lived on in copies and the loads went into fresh registers that had to be moved back)
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa0[u], pb0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa0[u], pb1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1[u], pb0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1[u], pb1[u], acc[1][1], 0, 0, 0);
            if (st + kGemm32Ahead < NS) fetch(u, kbeg + 4 * (st + kGemm32Ahead));
        }
    };
    if (steps == 64) run(std::integral_constant<int, 64>{});
    else if (steps == 48) run(std::integral_constant<int, 48>{});
    else if (steps == 32) run(std::integral_constant<int, 32>{});
    else if (steps == 16) run(std::integral_constant<int, 16>{});
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) sP[wv][a * 16 + lk + 4 * r][b * 16 + li] = acc[a][b][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = tid + 256 * u, r = e >> 5, c = e & 31, gr = row0 + r, gc = col0 + c;
        if (gr < n && gc < n) {
            const double sum = ((sP[0][r][c] + sP[1][r][c]) + sP[2][r][c]) + sP[3][r][c];
            const double v = 2. * xown[u] - sum;
            Xn[(size_t)gr * n + gc] = v;
            if (ti != tj) Xn[(size_t)gc * n + gr] = v;
            if (c32) {
                c32[(size_t)gr * c32_stride + gc] = (float)v;
                if (ti != tj) c32[(size_t)gc * c32_stride + gr] = (float)v;
            }
        }
    }
    if (c32 && tj == gt - 1 && tid < 32 && row0 + tid < n)                       // pad columns [n, stride) stay zero
        for (int q = n; q < c32_stride; q++) c32[(size_t)(row0 + tid) * c32_stride + q] = 0.f;
}
__device__ __forceinline__ void ml_ns_gemm32_kernel_body(int n, const double* __restrict__ X, const double* __restrict__ T,
                                                          double* __restrict__ Xn, float* __restrict__ c32, int c32_stride)
{
    __shared__ double sP[4][32][33];
    const int gt = (n + 31) / 32;
    int ti, tj;
    if (!gemm32_tile(gt, (int)blockIdx.x, ti, tj)) return;
    if ((tj + 1) * 32 <= n) ml_ns_gemm32_tile<false>(n, ti, tj, X, T, Xn, c32, c32_stride, sP);       // (ti <= tj: the rows are inside too)
    else ml_ns_gemm32_tile<true>(n, ti, tj, X, T, Xn, c32, c32_stride, sP);
}
__global__ __launch_bounds__(256) void ml_ns_gemm32_kernel(int n, const double* __restrict__ X, const double* __restrict__ T, double* __restrict__ Xn,
                                                          float* __restrict__ c32, int c32_stride)
{
    ml_ns_gemm32_kernel_body(n, X, T, Xn, c32, c32_stride);
}

// Y_cl += QY Q^T on the f64 matrix cores - the last term of the multiplicative cycle, 2 (6 n_cl)^2 (6 n_{cl+1}) flops (13 GFLOP at 20k
// vertices, where one lane per 6 x 6 block took 1.3 ms per rebuild).  Same tiling as ml_ns_gemm_kernel: 64 x 64 tiles on and above the
// diagonal, mirrored (the cycle's operator is symmetric).  QY and Q are stored as [entity][parent][6 x 6] blocks, so
//   A(row 6 i + r, k 6 p + c) = QY[i][p][r][c],   B(k 6 p + c, col 6 i' + r) = Q[i'][p][r][c];
// both are walked along k by consecutive lanes (runs of 6 doubles; the operands are a few MB and stay in L2).
__device__ __forceinline__ void ml_mult_qyqt_kernel_body(const MlDev* __restrict__ mlp, int cl)
{
    __shared__ double sA[kGemmTile][kGemmK + 1];      // sA[row][k]
    __shared__ double sB[kGemmTile][kGemmK + 1];      // sB[col][k]
    constexpr int kPer = kGemmTile * kGemmK / 256;
    const MlDev& ml = *mlp;
    const int np = ml.lv[cl + 1].n, n = 6 * ml.lv[cl].n, kd = 6 * np;
    const double* __restrict__ QY = ml.mQY;
    const double* __restrict__ Q = ml.mQ;
    double* __restrict__ Y = ml.Ydense[cl];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int gt = (n + kGemmTile - 1) / kGemmTile;
    int ti, tj;
    tri_tile_xcd(gt, (int)blockIdx.x, ti, tj);
    const int row0 = ti * kGemmTile, col0 = tj * kGemmTile;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = v4f64{0., 0., 0., 0.};
    double pa[kPer], pb[kPer];
    const unsigned qbytes = (unsigned)ml.lv[cl].n * (unsigned)np * 288u;           // both operands: n_cl x n_{cl+1} blocks of 6 x 6 (< 4 GB: n_cl n_{cl+1} < 14.9 M)
    const __amdgpu_buffer_rsrc_t rQY = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(QY), 0, (int)qbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Q), 0, (int)qbytes, 0x00020000);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < kPer; u++) {
            const int e = u * 256 + tid;
            const int er = e / kGemmK, ek = e % kGemmK;                           // consecutive lanes walk k
            const int gk = k0 + ek, p = gk / 6, c = gk % 6;
            const int gr = row0 + er, gc = col0 + er;
            // (buffer loads: an entry outside the operands is given an offset beyond them and reads as zero - no branch per element,
            //  all 32 loads of a slab in flight together)
            const unsigned oa = (unsigned)((gr / 6) * np + p) * 288u + (unsigned)((gr % 6) * 6 + c) * 8u;
            const unsigned ob = (unsigned)((gc / 6) * np + p) * 288u + (unsigned)((gc % 6) * 6 + c) * 8u;
            pa[u] = buf_load_f64(rQY, (gr < n && gk < kd) ? oa : 0xFFFFFFF8u);
            pb[u] = buf_load_f64(rQ, (gc < n && gk < kd) ? ob : 0xFFFFFFF8u);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kd; k0 += kGemmK) {
#pragma unroll
        for (int u = 0; u < kPer; u++) {
            const int e = u * 256 + tid;
            sA[e / kGemmK][e % kGemmK] = pa[u];
            sB[e / kGemmK][e % kGemmK] = pb[u];
        }
        __syncthreads();
        if (k0 + kGemmK < kd) fetch(k0 + kGemmK);
#pragma unroll 4
        for (int k4 = 0; k4 < kGemmK; k4 += 4) {
            const double a0 = sA[wr + li][k4 + lk], a1 = sA[wr + 16 + li][k4 + lk];
            const double b0 = sB[wc + li][k4 + lk], b1 = sB[wc + 16 + li][k4 + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gr = row0 + wr + a * 16 + lk + 4 * r, gc = col0 + wc + b * 16 + li;
                if (gr < n && gc < n) {
                    const double v = Y[(size_t)gr * n + gc] + acc[a][b][r];
                    Y[(size_t)gr * n + gc] = v;
                    if (ti != tj) Y[(size_t)gc * n + gr] = v;
                }
            }
}
__global__ __launch_bounds__(256) void ml_mult_qyqt_kernel(const MlDev* __restrict__ mlp, int cl)
{
    ml_mult_qyqt_kernel_body(mlp, cl);
}

// ---- per LM trial: dense inverse of the top level A_L(lambda) (<= 96 x 96: kMlTopWide aggregates), one workgroup of 16 x 16 lanes,
//      one 6 x 6 tile per lane (gj_tiles above); the last workgroup of ml_inverses_kernel
__device__ __forceinline__ void ml_top_tiles(PgoDev D, const MlDev* __restrict__ mlp, double* __restrict__ sRow, double* __restrict__ sCol)
{
    static_assert(kSibPerBlk * 48 >= 6 * kMlTopWide, "the pivot row / column buffers are shared with the sibling blocks");
    const MlDev& ml = *mlp;
    const MlLevel& L = ml.lv[ml.levels];
    const double lambda = D.scal[3];
    const int nt = L.n, n = 6 * nt;
    const int I = threadIdx.x >> 4, J = threadIdx.x & 15;
    const bool act = I < nt && J < nt;
    double a[36];
#pragma unroll
    for (int k = 0; k < 36; k++) a[k] = 0.;
    if (act && I == J) {
#pragma unroll
        for (int k = 0; k < 36; k++) a[k] = L.G[(size_t)I * 36 + k] + lambda * L.M[(size_t)I * 36 + k];
    } else if (act) {                                    // blocks of a coarse level are unique per (row, column)
        const int s0 = L.row_ptr[I], s1 = L.row_ptr[I + 1];
        int hit = -1;
        for (int s = s0; s < s1; s++) if (L.col[s] == J) hit = s;
        if (hit >= 0) {
#pragma unroll
            for (int k = 0; k < 36; k++) a[k] = L.blk[(size_t)hit * 36 + k];
        }
    }
    gj_tiles<kMlTopWide>(a, I, J, nt, sRow, sCol);
    if (act) {
        double* __restrict__ out = ml.top_inv + (size_t)(6 * I) * n + 6 * J;
#pragma unroll
        for (int k = 0; k < 36; k++) out[(k / 6) * n + k % 6] = a[k];
    }
}
__global__ __launch_bounds__(kBlk) void ml_inverses_kernel(PgoDev D, const MlDev* __restrict__ mlp)
{
    ml_inverses_kernel_body(D, mlp);
}
// The dense operator the PCG kernels apply, rounded to f32 once per rebuild (MlHot::Cmat32).  It is the largest stream of an
// iteration (config 2: 4.5 of 10 MB; 20k vertices: 112 MB); a preconditioner does not need the last 29 bits, and being rounded once,
// outside the iteration, it is still one fixed linear operator for the whole solve.  Four columns per lane; pad columns are zero.
__device__ __forceinline__ void ml_cmat32_body(const double* __restrict__ src, float* __restrict__ dst, int n6, int stride)
{
    const int q4 = stride >> 2;
    const long t = (long)blockIdx.x * kBlk + threadIdx.x;
    if (t >= (long)n6 * q4) return;
    const int row = (int)(t / q4), c = (int)(t % q4) * 4;
    const double* __restrict__ sr = src + (size_t)row * n6 + c;
    float4 o;
    o.x = (float)sr[0];                       // c < n6 always (stride - n6 < 4)
    o.y = (c + 1 < n6) ? (float)sr[1] : 0.f;
    o.z = (c + 2 < n6) ? (float)sr[2] : 0.f;
    o.w = (c + 3 < n6) ? (float)sr[3] : 0.f;
    *reinterpret_cast<float4*>(dst + (size_t)row * stride + c) = o;
}
__global__ __launch_bounds__(kBlk) void ml_cmat32_kernel(const double* __restrict__ src, float* __restrict__ dst, int n6, int stride)
{
    ml_cmat32_body(src, dst, n6, stride);
}


// ------------------------------------------------------------------------------------------------
// PCG-iteration kernels: two launches per iteration with the full multilevel preconditioner.
//
// One workgroup of either kernel owns one level-2 aggregate = 4 level-1 aggregates = 32 consecutive rows.
//   ml_spmv : 8 waves x 4 rows.  beta from the r.z partials, p = z + beta p_old (own rows + recomputed for the
//             neighbour columns), Ap, p.Ap partial, and Sg[A] = restriction of Ap to the gather level g = min(2, L).
//   ml_cg   : alpha from the p.Ap partials.  Because r_new = r - alpha Ap, the gather-level residual of EVERY
//             aggregate is rg_old - alpha Sg: no global pass over r.  Every workgroup restricts that up to the top
//             level in LDS, applies the top inverse for its own ancestor, walks down its own chain; then x, r, z for
//             its 32 rows, and the EXACT r1 / r2 of its own aggregates (written to rg_new, so the recursion never
//             accumulates error).  r.z partial -> part_b.
// rg and p are double-buffered (other workgroups read the old buffer while the owner writes the new one).
// All global operands whose address does not depend on alpha/beta are loaded before the partial reduction so that
// one memory latency covers them: these kernels are latency-bound, not bandwidth-bound, at pose-graph sizes.
// ------------------------------------------------------------------------------------------------
// Two geometries, chosen by graph size (template parameter AGG = level-1 aggregates per workgroup):
//   AGG = 1 (small graphs, <= 2560 free vertices): a workgroup owns ONE level-1 aggregate (8 rows), the gather
//           level is 1, hierarchy fan-outs 8,8,8,..  -> 8x more workgroups, i.e. CUs, for the latency-bound kernels
//   AGG = 4 (large graphs): a workgroup owns one level-2 aggregate = 4 level-1 aggregates (32 rows), gather level 2,
//           hierarchy fan-outs 8,4,8,8,..             -> the gathered arrays stay small (n/32 entries)
constexpr int kCgBlk = 192;                             // 3 waves; the first 48*AGG threads own a (row, component)
constexpr int kGatherU = 16;                            // gather-level values per thread kept in registers
constexpr int kChain = 6 * 48 + 3;                      // per ancestor level: 6 rows of its sibling-block inverse + its offset d

// The stop test's share of ml_cg (kCgBlk = 3 waves; progress_decide_ml, pgo_device.hpp).  At a look: the largest movement of the
// workgroup's rows since the last look, in units of the accuracy asked for, and |r|^2 of its rows; at the first application of the
// preconditioner (r = b): |b|^2 of its rows.  Per wave into spm[0..2] / spm[3..5]; look_store (lane 0, behind a workgroup barrier)
// folds the waves - the sum in wave order - into part_c[blk].
template <bool LDS = false>
__device__ __forceinline__ void look_partials(double moved, double sq, int tid, double* spm)
{
    const double m = wave_max<LDS>(moved), s = wave_sum<LDS>(sq);
    if ((tid & 63) == 0) { spm[tid >> 6] = m; spm[3 + (tid >> 6)] = s; }
}
__device__ __forceinline__ void look_store(double* __restrict__ part_c, int blk, const double* spm)
{
    reinterpret_cast<double2*>(part_c)[blk] = make_double2(fmax(fmax(spm[0], spm[1]), spm[2]), (spm[3] + spm[4]) + spm[5]);
}
template <int NW, bool LDS = false>
__device__ __forceinline__ double block_sum_w(double v, double* sN)
{
    v = wave_sum<LDS>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sN[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.;
#pragma unroll
    for (int k = 0; k < NW; k++) t += sN[k];
    return t;
}
// component k of P^T r for a child with offset d (levels >= 1):  [r_v ; d x r_v + r_w]
__device__ __forceinline__ double restrict_comp(const double* __restrict__ d, const double* __restrict__ rc, int k)
{
    if (k < 3) return rc[k];
    const int j = k - 3;
    const double cr = (j == 0) ? d[1] * rc[2] - d[2] * rc[1] : (j == 1) ? d[2] * rc[0] - d[0] * rc[2] : d[0] * rc[1] - d[1] * rc[0];
    return cr + rc[k];
}
// component k of P y for a child with offset d:  [v + w x d ; w]
__device__ __forceinline__ double prolong_comp(const double* __restrict__ d, const double* __restrict__ yp, int k)
{
    if (k >= 3) return yp[k];
    const double cr = (k == 0) ? yp[4] * d[2] - yp[5] * d[1] : (k == 1) ? yp[5] * d[0] - yp[3] * d[2] : yp[3] * d[1] - yp[4] * d[0];
    return yp[k] + cr;
}
// Workgroup-wide sum of n block partials, lane `tid` of BLK taking elements tid, tid + BLK, ...: the first NU per lane are fetched
// together at kernel entry (independent loads: one round trip, and whatever is issued behind them is in flight at the same time),
// the rest - graphs beyond NU * BLK workgroups - one by one.  A `for (...) s += part[i]` loop waits for every load in turn AND
// holds back every load that follows it in program order: four serial round trips in front of ml_cg at 20k vertices.
// Same summation order as that loop.
template <int BLK, int NU>
__device__ __forceinline__ void part_issue(const double* __restrict__ part, int n, int tid, double (&v)[NU])
{
#pragma unroll
    for (int u = 0; u < NU; u++) { const int i = tid + u * BLK; const double x = part[i < n ? i : 0]; v[u] = (i < n) ? x : 0.; }
}
template <int BLK, int NU>
__device__ __forceinline__ double part_fold(const double* __restrict__ part, int n, int tid, const double (&v)[NU])
{
    double s = v[0];
#pragma unroll
    for (int u = 1; u < NU; u++) s += v[u];
    for (int i = tid + NU * BLK; i < n; i += BLK) s += part[i];
    return s;
}
// (the geometry of a row lives in registers: its entries are picked with selects, never with a computed index - a computed
//  index sends the array through scratch memory, one more dependent round trip in kernels that are nothing but round trips)
__device__ __forceinline__ double sel3(int k, double a, double b, double c) { return (k == 0) ? a : (k == 1) ? b : c; }
// (P1^T v)[r] for a row with geo = {R^T (9), d (3)} and v = (t0,t1,t2,q0,q1,q2)
__device__ __forceinline__ double p1t_comp(const double* geo, double t0, double t1, double t2, double q0, double q1, double q2, int r)
{
    const double u0 = fma(geo[6], t2, fma(geo[3], t1, geo[0] * t0));      // R = (R^T)^T
    const double u1 = fma(geo[7], t2, fma(geo[4], t1, geo[1] * t0));
    const double u2 = fma(geo[8], t2, fma(geo[5], t1, geo[2] * t0));
    if (r < 3) return (r == 0) ? u0 : (r == 1) ? u1 : u2;
    const int k = r - 3;
    const double rq = 0.5 * fma(sel3(k, geo[6], geo[7], geo[8]), q2, fma(sel3(k, geo[3], geo[4], geo[5]), q1, sel3(k, geo[0], geo[1], geo[2]) * q0));
    const double dx = geo[9], dy = geo[10], dz = geo[11];
    const double cr = (k == 0) ? fma(dy, u2, -(dz * u1)) : (k == 1) ? fma(dz, u0, -(dx * u2)) : fma(dx, u1, -(dy * u0));
    return cr + rq;
}
// (P1 y)[r]
__device__ __forceinline__ double p1_comp(const double* geo, const double* y, int r)
{
    const int k = (r < 3) ? r : r - 3;
    const double g0 = sel3(k, geo[0], geo[3], geo[6]), g1 = sel3(k, geo[1], geo[4], geo[7]), g2 = sel3(k, geo[2], geo[5], geo[8]);   // row k of R^T
    if (r < 3) {
        const double dx = geo[9], dy = geo[10], dz = geo[11];
        const double vx = y[0] + fma(y[4], dz, -(y[5] * dy));      // v + w x d
        const double vy = y[1] + fma(y[5], dx, -(y[3] * dz));
        const double vz = y[2] + fma(y[3], dy, -(y[4] * dx));
        return fma(g2, vz, fma(g1, vy, g0 * vx));       // R^T (.)
    }
    return 0.5 * fma(g2, y[5], fma(g1, y[4], g0 * y[3]));
}

// x = 0, r = b, p0 = p1 = 0, flags cleared; exact gather-level residual of the own aggregates -> rg
template <int AGG>
__device__ __forceinline__ void ml_init_kernel_body(PgoDev D, MlHot H, double* __restrict__ p0, double* __restrict__ p1,
                                                        double* __restrict__ rg)
{
    constexpr int kRowsPerBlk = kMlFanout * AGG, kAggPerBlk = AGG;
    __shared__ double sv[kCgBlk];
    __shared__ double sw[kCgBlk];
    __shared__ double sr1[kAggPerBlk * 6];
    const int tid = threadIdx.x;
    const int gl = (AGG == 1 || H.levels < 2) ? 1 : 2;
    const int a = blockIdx.x * kRowsPerBlk + tid / 6, r = tid % 6;
    const bool act = tid < kRowsPerBlk * 6 && a < D.nb;
    double rv = 0., geo[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const size_t i = (size_t)a * 6 + r;
        rv = D.b[i];
        D.r[i] = rv; D.x[i] = 0.; D.xs[i] = 0.; p0[i] = 0.; p1[i] = 0.;
        const double* __restrict__ gg = H.geo0 + (size_t)a * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
    sv[tid] = rv;
    __syncthreads();
    const int g0 = tid - r;
    sw[tid] = act ? p1t_comp(geo, sv[g0], sv[g0 + 1], sv[g0 + 2], sv[g0 + 3], sv[g0 + 4], sv[g0 + 5], r) : 0.;
    __syncthreads();
    const int n1 = H.n[1];
    if (tid < kAggPerBlk * 6) {
        const int la = tid / 6, k = tid % 6, A1 = blockIdx.x * kAggPerBlk + la;
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[(la * kMlFanout + j) * 6 + k];
        sr1[tid] = s;
        if (gl == 1 && A1 < n1) rg[(size_t)A1 * 6 + k] = s;
    }
    __syncthreads();
    if (gl == 2 && tid < 6) {
        double s = 0.;
        for (int la = 0; la < kAggPerBlk; la++) {
            const int A1 = blockIdx.x * kAggPerBlk + la;
            if (A1 < n1) s += restrict_comp(H.geo[1] + (size_t)A1 * 3, sr1 + la * 6, tid);
        }
        rg[(size_t)blockIdx.x * 6 + tid] = s;
    }
    if (blockIdx.x == 0 && tid == 0) { D.flags[0] = 0; D.flags[1] = 0; D.flags[2] = 0; D.flags[3] = 0; D.scal[2] = 1.; }
}
template <int AGG>
__global__ __launch_bounds__(kCgBlk) void ml_init_kernel(PgoDev D, MlHot H, double* __restrict__ p0, double* __restrict__ p1, double* __restrict__ rg)
{
    ml_init_kernel_body<AGG>(D, H, p0, p1, rg);
}

// This file is compiled with -ffp-contract=off (two geometries of one kernel body must give the same bits): the products that matter
// for speed say fma() themselves.
__device__ __forceinline__ double dot6(double2 a0, double2 a1, double2 a2, double v0, double v1, double v2, double v3, double v4, double v5)
{
    double t = a0.x * v0;
    t = fma(a0.y, v1, t); t = fma(a1.x, v2, t); t = fma(a1.y, v3, t); t = fma(a2.x, v4, t); t = fma(a2.y, v5, t);
    return t;
}
// restricted A p of gather-level entity t / 6, component t % 6.  Gather level 2 (AGG = 4): the sum of ml_spmv's two half-aggregate parts
template <int AGG>
__device__ __forceinline__ double sg_at(const double* __restrict__ Sg, int gl, int t)
{
    if (AGG == 1 || gl != 2) return Sg[t];
    const double2 v = reinterpret_cast<const double2*>(Sg)[t];          // [aggregate][component][half]: one 16-byte load, no index arithmetic
    return v.x + v.y;
}
// AGG = 1: 8 waves per workgroup, one row per wave (8 rows = one level-1 aggregate).
// AGG = 4: four rows per wave, 4 waves per workgroup (kSpmvWaves4): 16 rows = HALF a level-2 aggregate; the two halves' restricted
//   A p go to Sg[aggregate][component][half] and ml_cg adds them (sg_at).  The kernel's time is set by the busiest CU (one 32-row workgroup
//   per CU took 15.5 us, two 23.7): 313 workgroups of a 10k-vertex graph on 256 CUs left 57 CUs with twice the work of the rest;
//   626 half workgroups put at most 1.5 times the mean on one CU.
constexpr int kSpmvWaves4 = 4;
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx.x mod 8), each with an L2 of its own.  Rows that are neighbours on the
// trajectory share vectors and odometry blocks, so every XCD gets a CONTIGUOUS range of row groups instead of every eighth one:
// logical index = start(blockIdx.x mod 8) + blockIdx.x / 8.  A bijection on [0, gridDim.x); partials stay indexed by the logical
// index, so every sum is taken in the same order as before - results do not change.
__device__ __forceinline__ int xcd_contiguous(int b, int g)
{
    const int q = g >> 3, rem = g & 7, x = b & 7;
    return x * q + (x < rem ? x : rem) + (b >> 3);
}
// Geometry of the workgroup (RPW rows per wave x WAVES waves) is separate from the hierarchy it serves (AGG): the batched solve runs
// the AGG = 1 hierarchy with four rows per wave in 128-lane workgroups (ml_spmv_batch_kernel) - four times the bytes in flight per
// wave when sixteen graphs fill the chip - where a single small graph wants one row per wave for the shortest chain.  Both give the
// same bits: every sum that crosses rows or lanes is taken in an order that does not depend on the geometry (row sums of six
// components, then rows pairwise by index; the r.z partials in groups of 64, then groups in order).
template <int AGG, int RPW = AGG, int WAVES = (AGG == 1 ? 8 : kSpmvWaves4)>
__device__ __forceinline__ void ml_spmv_kernel_body(PgoDev D, MlHot H, const double* __restrict__ p_old,
                                                     double* __restrict__ p_new, int n_part, double tol2)
{
    constexpr int kWaves = WAVES, kRowsPerWave = RPW;
    constexpr bool kLds = (WAVES == 2);          // the batched geometry: wave sums through the LDS crossbar (wave_sum, pgo_device.hpp)
    constexpr int kRowsPerBlk = kWaves * kRowsPerWave, kAggPerBlk = kRowsPerBlk / kMlFanout;
    constexpr int kGrpU = (kWaves >= 8) ? 2 : 4;          // groups of 64 r.z partials a wave fetches up front (1024 / 1024 / 512 partials in all)
    __shared__ double sgrp[kMaxPartials / 64];
    __shared__ double slook[2][kMaxPartials / 64];      // block 0: the stop test's maxima per group of 64 ml_cg workgroups
    // The six lanes of a group need the same 6-vector, each holds one component: it goes through a 512-byte LDS line of the wave (one
    // 8-byte store, three 16-byte broadcast reads; a wave's LDS traffic is processed in order) instead of twelve ds_bpermute - the
    // kernel is bound by instruction issue as much as by its round trips.  One line per gather: nothing is reused inside a launch.
    constexpr int kGat = 2 + 3 * kRowsPerWave;
    __shared__ __attribute__((aligned(16))) double sgat[kWaves][kGat][64];
    __shared__ double sd[kRowsPerBlk * 6];
    __shared__ double sw[kRowsPerBlk * 6];
    __shared__ double ss1[kAggPerBlk * 6];
    __shared__ double sg1[kAggPerBlk * 3];
    const int done = D.flags[0];            // looked at behind the first loads (below): its round trip runs beside theirs, not in front
    STAMP_DECL
    const int gl = (AGG == 1 || H.levels < 2) ? 1 : 2;
    // (the wave index as a scalar: rows, row headers and the loops over partial groups are then uniform to the compiler as well)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane / 6, r = lane % 6;
    const bool lact = lane < 60;
    const int bx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);      // this workgroup's row group
    const int row0 = bx * kRowsPerBlk + wv * kRowsPerWave;
    // ---- prefetch (independent of beta); the r.z partials first: they gate everything else
    double vpart[kGrpU];                      // wave wv takes groups wv, wv + kWaves, ...; lane = element of the group
#pragma unroll
    for (int u = 0; u < kGrpU; u++) {
        const int i = (wv + u * kWaves) * 64 + lane;
        const double x = D.part_b[i < n_part ? i : 0];
        vpart[u] = (i < n_part) ? x : 0.;
    }
    // the stop test (progress_decide, pgo_device.hpp): block 0 folds what the last ml_cg left in part_c - fetched whether or not that
    // was a look iteration (the iteration count would be a round trip in front of these loads), used only if it was
    double2 vlook[kGrpU];
    if (bx == 0) {
#pragma unroll
        for (int u = 0; u < kGrpU; u++) {
            const int i = (wv + u * kWaves) * 64 + lane;
            const double2 x = reinterpret_cast<const double2*>(D.part_c)[i < n_part ? i : 0];
            vlook[u] = (i < n_part) ? x : make_double2(0., 0.);
        }
    }
    const int it = D.flags[1];
    const double rz_prev = D.scal[2], thr_old = D.scal[1], lambda = D.scal[3], rz_last = D.scal[0];
    // slot range and this lane group's columns of the first TWO slot passes: ONE hop (row header).  Rows have ~10 slots on the
    // BASELINE graphs, i.e. four in ten need a second pass; with its column already here the second pass is one more round trip
    // instead of two per row (column index, then block and vectors), and rows of a wave take it together instead of one by one.
    int s0[kRowsPerWave], s1[kRowsPerWave], cf[kRowsPerWave], cf2[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
        const int a = row0 + q;
        const int32_t* __restrict__ hd = D.rowhdr + (size_t)(a < D.nb ? a : 0) * kRowHdr;
        s0[q] = (a < D.nb) ? hd[0] : 0;
        s1[q] = (a < D.nb) ? hd[1] : 0;
        cf[q] = (a < D.nb && lact) ? hd[2 + g] : -1;
        cf2[q] = (a < D.nb && lact) ? hd[12 + g] : -1;
    }
    // (a launch after convergence must stay cheap: 16-iteration graph batches overshoot.  Leaving here costs it the issue of the loads
    //  above - nothing waits for them - and saves every working launch the done flag's round trip in front of its first load.)
    if (done) return;
    // offsets of the workgroup's level-1 aggregates (the restriction of A p at the end): the LOAD here, the store into LDS behind the
    // prefetch below.  (Rounds 1-5 stored at once: a full `s_waitcnt vmcnt(0)` between the row headers and the diagonal blocks, and - the
    // counts of loads in flight do not survive the divergent `dact` block - a second one behind the diagonal blocks' loads, in front of
    // the first slot pass: one whole memory round trip of the AGG = 4 kernel spent waiting for the diagonal block alone.)
    double g1v = 0.;
    if (tid < kAggPerBlk * 3) {
        const int A1 = bx * kAggPerBlk + tid / 3;
        if (gl == 2 && A1 < H.n[1]) g1v = H.geo[1][(size_t)A1 * 3 + tid % 3];
    }
    // Every lane (g, r) multiplies row r of a 6x6 block with the 6-vector z + beta p_old of the block's column.  The six lanes of a
    // group need the same vector: each loads ONE component (the six loads of a group are one contiguous 48 B) and the vector is
    // assembled by shuffles - 4 VGPRs per (row, pass) instead of 24, which is what lets a second workgroup share the CU (the kernel
    // is latency-bound: 313 workgroups of a 10k graph on 256 CUs took two rounds).  Same products, same summation order.
    const int gbase = g * 6;                      // first lane of this lane's group
    auto gather6 = [&](int slot, double v, int base, double& v0, double& v1, double& v2, double& v3, double& v4, double& v5) {
        double* line = sgat[wv][slot];
        line[lane] = v;
        __builtin_amdgcn_wave_barrier();
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(line + base);
        const double2 a = p2[0], b = p2[1], c = p2[2];
        v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y; v4 = c.x; v5 = c.y;
    };
    // lane group q (< AGG) owns the diagonal block, z, p_old and geometry of row q
    double hrow[6] = {0, 0, 0, 0, 0, 0}, zo_r = 0., po_r = 0.;
    const int arow = row0 + g;
    const bool dact = g < kRowsPerWave && arow < D.nb;
    if (dact) {
        const double* __restrict__ h = D.hdiag + (size_t)arow * 36 + r * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) hrow[c] = h[c];
        zo_r = D.z[(size_t)arow * 6 + r]; po_r = p_old[(size_t)arow * 6 + r];
    }
    // first slot pass of every row (rows have ~10 slots: most rows need exactly this pass)
    double2 b0[kRowsPerWave], b1[kRowsPerWave], b2[kRowsPerWave];
    double zr[kRowsPerWave], orr[kRowsPerWave];
    bool have[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
        have[q] = false; zr[q] = 0.; orr[q] = 0.;
        const int s = s0[q] + g;
        if (lact && s < s1[q]) {
            const int c = cf[q];
            if (c >= 0) {
                have[q] = true;
                const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                b0[q] = bk[0]; b1[q] = bk[1]; b2[q] = bk[2];
                zr[q] = D.z[(size_t)c * 6 + r]; orr[q] = p_old[(size_t)c * 6 + r];
            }
        }
    }
    // second slot pass: one row per wave (AGG = 1) has the registers to fetch it up front as well
    double2 c0[kRowsPerWave], c1[kRowsPerWave], c2[kRowsPerWave];
    double yr[kRowsPerWave], qr[kRowsPerWave];
    bool have2[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
        have2[q] = false; yr[q] = 0.; qr[q] = 0.;
        if (kRowsPerWave == 1) {
            const int s = s0[q] + g + 10;
            const int c = cf2[q];
            if (lact && s < s1[q] && c >= 0) {
                have2[q] = true;
                const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                c0[q] = bk[0]; c1[q] = bk[1]; c2[q] = bk[2];
                yr[q] = D.z[(size_t)c * 6 + r]; qr[q] = p_old[(size_t)c * 6 + r];
            }
        }
    }
    STAMP(16);     // 16: prefetch issue
    if (tid < kAggPerBlk * 3) sg1[tid] = g1v;      // (read behind the barriers below)
    // ---- beta
    const int n_grp = (n_part + 63) >> 6;
#pragma unroll
    for (int u = 0; u < kGrpU; u++) {
        const int gi = wv + u * kWaves;
        if (gi < n_grp) {                                  // (uniform in the wave)
            const double sgi = wave_sum<kLds>(vpart[u]);
            if (lane == 0) sgrp[gi] = sgi;
        }
    }
    for (int gi = wv + kGrpU * kWaves; gi < n_grp; gi += kWaves) {       // beyond kGrpU * kWaves * 64 partials: one group per round trip
        const int i = gi * 64 + lane;
        const double sgi = wave_sum<kLds>(i < n_part ? D.part_b[i] : 0.);
        if (lane == 0) sgrp[gi] = sgi;
    }
    if (bx == 0) {                                        // (.x: a maximum - any order gives the same bits; .y: a sum, taken like r.z)
#pragma unroll
        for (int u = 0; u < kGrpU; u++) {
            const int gi = wv + u * kWaves;
            if (gi < n_grp) {
                const double a = wave_max<kLds>(vlook[u].x), b = wave_sum<kLds>(vlook[u].y);
                if (lane == 0) { slook[0][gi] = a; slook[1][gi] = b; }
            }
        }
        for (int gi = wv + kGrpU * kWaves; gi < n_grp; gi += kWaves) {
            const int i = gi * 64 + lane;
            const double2 x = (i < n_part) ? reinterpret_cast<const double2*>(D.part_c)[i] : make_double2(0., 0.);
            const double a = wave_max<kLds>(x.x), b = wave_sum<kLds>(x.y);
            if (lane == 0) { slook[0][gi] = a; slook[1][gi] = b; }
        }
    }
    __syncthreads();
    double rz = 0.;
    for (int gi = 0; gi < n_grp; gi++) rz += sgrp[gi];
    const double beta = (it == 0) ? 0. : rz / rz_prev;
    const double thresh = (it == 0) ? (tol2 * D.scal[8]) * rz : thr_old;       // scal[8]: the LM iteration's tightening of pcg_tol^2 (uzl_pgo.hip)
    STAMP(16);     // 17: partial reduction (prefetch landed)
    // ---- row products: diagonal block + first slot pass of every row ...
    const double pr = fma(beta, po_r, zo_r);          // component r of the new direction of row `arow` (lanes with dact)
    double acc[kRowsPerWave];
    {
        double d0, d1, d2, d3, d4, d5;
        gather6(0, pr, gbase, d0, d1, d2, d3, d4, d5);
#pragma unroll
        for (int q = 0; q < kRowsPerWave; q++) {
            double dterm = 0.;                        // (H_aa + lambda I) p of row q, in the lanes of group q (they hold H_aa and p)
            if (dact && g == q) {
                dterm = hrow[0] * d0; dterm = fma(hrow[1], d1, dterm); dterm = fma(hrow[2], d2, dterm); dterm = fma(hrow[3], d3, dterm);
                dterm = fma(hrow[4], d4, dterm); dterm = fma(hrow[5], d5, dterm);
                dterm = fma(lambda, pr, dterm);
                if (!D.diag_owner) dterm = 0.;        // sharded solve: the diagonal term is added by one rank only
                p_new[(size_t)arow * 6 + r] = pr;
            }
            // it enters the row's fold in lane group 0 whatever group computed it (with one row per wave that is the same group):
            // the fold adds the ten groups in a fixed tree, so the place decides the rounding
            const double dmov = (kRowsPerWave == 1) ? dterm : __shfl(dterm, 6 * q + (lane < 6 ? lane : 0));
            double aq = (g == 0) ? dmov : 0.;
            const double pc = fma(beta, orr[q], zr[q]);
            double v0, v1, v2, v3, v4, v5;
            gather6(2 + q, pc, gbase, v0, v1, v2, v3, v4, v5);
            if (have[q]) aq += dot6(b0[q], b1[q], b2[q], v0, v1, v2, v3, v4, v5);
            acc[q] = aq;
        }
    }
    // ... the second pass of all rows of the wave in one round trip (several rows per wave: into the registers the first pass has just freed) ...
    if (kRowsPerWave != 1) {
#pragma unroll
        for (int q = 0; q < kRowsPerWave; q++) {
            const int s = s0[q] + g + 10;
            const int c = cf2[q];
            if (lact && s < s1[q] && c >= 0) {
                have2[q] = true;
                const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                c0[q] = bk[0]; c1[q] = bk[1]; c2[q] = bk[2];
                yr[q] = D.z[(size_t)c * 6 + r]; qr[q] = p_old[(size_t)c * 6 + r];
            }
        }
    }
    // ... then whatever a long row has beyond 20 slots, and the folds (same summation order as a slot-by-slot walk)
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
        double aq = acc[q];
        {
            const double pc = fma(beta, qr[q], yr[q]);
            double v0, v1, v2, v3, v4, v5;
            gather6(2 + kRowsPerWave + q, pc, gbase, v0, v1, v2, v3, v4, v5);
            if (have2[q]) aq += dot6(c0[q], c1[q], c2[q], v0, v1, v2, v3, v4, v5);
        }
        // rows with more than 20 slots are rare (hubs): the wave walks their remaining passes together (a uniform trip count, so
        // that the shuffles stay convergent)
        const int smax = __builtin_amdgcn_readfirstlane(s1[q]);
        for (int sb = s0[q] + 20; sb < smax; sb += 10) {
            const int s = sb + g;
            double2 e0 = make_double2(0., 0.), e1 = e0, e2 = e0;
            double wz = 0., wp = 0.;
            bool hv = false;
            if (lact && s < s1[q]) {
                const int c = D.col[s];
                if (c >= 0) {
                    hv = true;
                    const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                    e0 = bk[0]; e1 = bk[1]; e2 = bk[2];
                    wz = D.z[(size_t)c * 6 + r]; wp = p_old[(size_t)c * 6 + r];
                }
            }
            const double pc = fma(beta, wp, wz);
            double v0, v1, v2, v3, v4, v5;
            __builtin_amdgcn_wave_barrier();                 // (line 1 is reused from pass to pass: the reads of the last one are done)
            gather6(1, pc, gbase, v0, v1, v2, v3, v4, v5);
            if (hv) aq += dot6(e0, e1, e2, v0, v1, v2, v3, v4, v5);
        }
        double t;
        t = __shfl_down(aq, 48); if (lane + 48 < 60) aq += t;
        t = __shfl_down(aq, 24); if (lane + 24 < 48) aq += t;
        t = __shfl_down(aq, 12); if (lane + 12 < 24) aq += t;
        t = __shfl_down(aq, 6);  if (lane + 6 < 12) aq += t;
        acc[q] = aq;                               // lanes 0..5: (A p)[row q][0..5]
    }
    STAMP(16);     // 18: row products + folds
    double geo[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};         // geometry of row `arow`: fetched here, not held across the row products
    if (dact) {
        const double* __restrict__ gg = H.geo0 + (size_t)arow * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
#pragma unroll
    for (int q = 0; q < kRowsPerWave; q++) {
        const int a = row0 + q;
        if (lane < 6 && a < D.nb) D.ap[(size_t)a * 6 + lane] = acc[q];
        double t0, t1, t2, q0, q1, q2;
        gather6(2 + 2 * kRowsPerWave + q, acc[q], 0, t0, t1, t2, q0, q1, q2);
        if (g == q) {                               // the lanes that own row q's p and geometry
            double wq = 0., dq = 0.;
            if (dact) {
                const double apr = (r == 0) ? t0 : (r == 1) ? t1 : (r == 2) ? t2 : (r == 3) ? q0 : (r == 4) ? q1 : q2;
                dq = apr * pr;
                wq = p1t_comp(geo, t0, t1, t2, q0, q1, q2, r);
            }
            sw[(wv * kRowsPerWave + q) * 6 + r] = wq;
            sd[(wv * kRowsPerWave + q) * 6 + r] = dq;
        }
    }
    __syncthreads();                                    // sw, sd complete
    // p.Ap of the workgroup's rows: six components per row, then the rows pairwise by row index (a tree that does not depend on how
    // rows map to waves; every lane < kRowsPerBlk ends with the same bits)
    double dtot = 0.;
    if (wv == 0) {
        if (lane < kRowsPerBlk) {
            const double* dd = sd + lane * 6;
            dtot = ((dd[0] + dd[1]) + (dd[2] + dd[3])) + (dd[4] + dd[5]);
        }
#pragma unroll
        for (int o = 1; o < kRowsPerBlk; o <<= 1) dtot += __shfl_xor(dtot, o);
    }
    if (tid < kAggPerBlk * 6) {
        const int la = tid / 6, k = tid % 6, A1 = bx * kAggPerBlk + la;
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[(la * kMlFanout + j) * 6 + k];
        ss1[tid] = s;
        if (gl == 1 && A1 < H.n[1]) H.Sg[(size_t)A1 * 6 + k] = s;
    }
    if (AGG != 1) __syncthreads();
    if (AGG != 1 && gl == 2 && tid < 6) {
        double s = 0.;
        for (int la = 0; la < kAggPerBlk; la++)
            if (bx * kAggPerBlk + la < H.n[1]) s += restrict_comp(sg1 + la * 3, ss1 + la * 6, tid);
        H.Sg[((size_t)(bx / (8 / kSpmvWaves4)) * 6 + tid) * (8 / kSpmvWaves4) + bx % (8 / kSpmvWaves4)] = s;     // [aggregate][component][half]
    }
    if (tid == 0) {
        D.part_a[bx] = dtot;
        if (bx == 0) {
            if (it == 0) {                               // the first ml_cg (r = b) left |b|^2
                double bb = 0.;
                for (int gi = 0; gi < n_grp; gi++) bb += slook[1][gi];
                D.scal[14] = bb;
            } else if (it % kProgressEvery == 0) {       // the ml_cg that ended iteration `it` was a look: how far has x moved since the last one
                double m = 0., rr = 0.;
                for (int gi = 0; gi < n_grp; gi++) { m = fmax(m, slook[0][gi]); rr += slook[1][gi]; }
                progress_decide_ml(D, m, rr, rz_last);   // rz_last: r.M^-1 r at the START of that iteration, as this launch found it in scal[0]
            }
            D.flags[3] = ((it + 1) % kProgressEvery == 0) ? 1 : 0;         // is the ml_cg behind this launch a look
            D.scal[0] = rz;
            if (it == 0) { D.scal[1] = thresh; D.scal[11] = rz; D.scal[15] = 0.; }      // ([15]: no movement seen yet)
            if (!(rz > thresh)) D.flags[0] = 1;
            if (!(rz >= 0.)) D.flags[2] = 1;      // r.M^-1 r < 0 (or NaN): M^-1 is not positive definite - breakdown, not convergence
        }
    }
    STAMP(16);     // 19: restriction of Ap + stores
#ifdef UZL_STAMPS
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_stamps[47], 1ull);
#endif
}
// AGG = 1: 4 waves per SIMD = two 512-lane workgroups per CU (<= 128 VGPRs; the body needs 98).  AGG = 4: 140 VGPRs, 3 waves per SIMD =
// three 256-lane workgroups per CU (held at 128 it spilled 44 B per lane and was slower: 18.5 vs 17.3 us at 10k vertices)
template <int AGG>
__global__ __launch_bounds__(AGG == 1 ? 512 : 64 * kSpmvWaves4) __attribute__((amdgpu_waves_per_eu(AGG == 1 ? 4 : 3))) void ml_spmv_kernel(PgoDev D, MlHot H, const double* __restrict__ p_old, double* __restrict__ p_new, int n_part, double tol2)
{
    ml_spmv_kernel_body<AGG>(D, H, p_old, p_new, n_part, tol2);
}
constexpr int kSpmvBatchRpw = 4, kSpmvBatchWaves = 2;      // ml_spmv_batch_kernel: 8 rows = one level-1 aggregate in 128 lanes

// init = 1: first application (r = b stored, exact rg in rg_old): only the preconditioner part runs.
// Dynamic LDS (doubles): res[levels g..L] | geo[levels g..L-1] | top rows | own-chain sibling rows + offsets   (ml_cg_lds_bytes)
// COMP (AGG = 4 only): the dense level-2 operator is present - a compile-time fact, so that the registers and LDS staging of the
// restrict / top-solve / sibling-chain walk it replaces are not allocated (230 -> fewer VGPRs: more workgroups per CU for a kernel
// whose level-2 product streams 6 rows of Y_2 per workgroup)
// YPRE (COMP, 6 n_2 <= 2304 = up to ~12k free vertices): the workgroup's six rows of Y_2 fit the register file (18 x 16 B per lane), so
// they are fetched at entry with everything else and the level-2 product runs out of registers the moment alpha is known - one
// memory round trip per launch instead of two (10k/50k: 12.7 -> ~9 us).  ~220 VGPRs: two workgroups per CU, enough for the <= 375
// workgroups of such a graph.
// VPRE (COMP, larger graphs: 12k .. 21.8k vertices): alpha and the gather-level residual estimate v = rg - alpha Sg come from
// ml_alpha_kernel, launched in front.  Without it every one of the 400 - 680 workgroups reduces the same partials and stages the same
// two (with the half-aggregate parts: three) 30-KB vectors through its registers - at 20k vertices that, not the product with Y_2,
// was most of the kernel (33 us per launch).
constexpr int kYU = 18;
template <int AGG, bool COMP = false, bool YPRE = false, bool VPRE = false>
__device__ __forceinline__ void ml_cg_kernel_body(PgoDev D, MlHot H, const double* __restrict__ p,
                                                      const double* __restrict__ rg_old, double* __restrict__ rg_new,
                                                      int n_part, int init)
{
    constexpr int kRowsPerBlk = kMlFanout * AGG, kAggPerBlk = AGG;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double s3[3];
    __shared__ double sv[kCgBlk];
    __shared__ double sw[kCgBlk];
    __shared__ double sr1[kAggPerBlk * 6];
    __shared__ double sy[kAggPerBlk * 6];
    __shared__ double syc[6];
    __shared__ double szj[kRowsPerBlk * 6];
    __shared__ double spm[6];
    constexpr int kFan2 = (AGG == 1) ? kMlFanout : kMlFanout2;    // children of a level-2 aggregate (build_ml)
    if (D.flags[0]) return;               // (kept first: post-convergence launches of a graph batch must stay cheap no-ops)
    STAMP_DECL
    const int tid = threadIdx.x;
    const int Lt = H.levels;
    const int gl = (AGG == 1 || Lt < 2) ? 1 : 2;               // gather level
    // AGG = 4 with the dense operator of level 2 (H.Cmat = Y_2, rows 6 A .. 6 A + 5 belong to workgroup A): the level-2
    // correction is six rows of Y_2 times the gather-level residual; the restrict / top-solve / sibling-chain walk is skipped
    constexpr bool comp = (AGG == 4) && COMP;
    constexpr int kGU = comp ? 12 : kGatherU;      // gather-level values per thread held in registers (COMP: 12 x 192 covers 12k vertices at 168 VGPRs without a spill)
    const int a = blockIdx.x * kRowsPerBlk + tid / 6, r = tid % 6;
    const bool act = tid < kRowsPerBlk * 6 && a < D.nb;
    const int n1 = H.n[1], ng = H.n[gl];
    const int ntop = 6 * H.n[Lt];
    // ---- LDS carve-up
    // (COMP uses none of this: the gather-level vector sits at offset 0 and nothing else is staged.  Left in, the tables - computed
    //  indices, so scratch memory - and the integer divisions of the ancestor chain run in front of the kernel's first load.)
    int roff[kMlMaxLevels + 2], goff[kMlMaxLevels + 2], anc[kMlMaxLevels + 2];
    int o = 0, top_off = 0, chain_off = 0;
    const int n_top_rows = (Lt == 1) ? kAggPerBlk * 6 : 6;
    if (!comp) {
        for (int l = gl; l <= Lt; l++) { roff[l] = o; o += 6 * H.n[l]; }
        for (int l = gl; l < Lt; l++) { goff[l] = o; o += 3 * H.n[l]; }
        top_off = o;
        o += n_top_rows * ntop;
        chain_off = o;                                        // (L-2) x kChain
        anc[1] = blockIdx.x * kAggPerBlk;                      // (only its parent chain is used)
        anc[2] = (gl == 2) ? (int)blockIdx.x : anc[1] / H.fan[2 <= Lt ? 2 : 1];
        for (int l = 3; l <= Lt; l++) anc[l] = anc[l - 1] / H.fan[l];
    }
    STAMP(0);      // 0: entry
    // ---- every global load whose address is known now, before any barrier
    double vpart[4] = {0., 0., 0., 0.};
    if (!init && !VPRE) part_issue<kCgBlk, 4>(D.part_a, n_part, tid, vpart);
    double xv = 0., xsv = 0., rv0 = 0., apv = 0., pv = 0., geo[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int look = init ? 1 : D.flags[3];       // the stop test wants this workgroup's partials (ml_spmv said so; always with r = b: look_partials)
    const double unit = progress_unit(D.scal, r);
    if (act) {
        const size_t i = (size_t)a * 6 + r;
        rv0 = D.r[i];
        if (!init) { xv = D.x[i]; xsv = D.xs[i]; apv = D.ap[i]; pv = p[i]; }
        const double* __restrict__ gg = H.geo0 + (size_t)a * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
    // level-0 smoother.  AGG = 1: W0^-1 of the own level-1 aggregate (48 x 48), 4 threads per output row, 12 columns
    // each.  AGG = 4 (large graphs): the level-0 blocks are built block-diagonal (PgoDev::sibling0 = 0: four dense
    // 48 x 48 slices per workgroup and iteration would cost more traffic than the iterations they save), so every
    // (row, component) thread needs only the 6 entries of its own diagonal block.
    double w0[(AGG == 1) ? 12 : 6];
    if (AGG == 1) {
        const int out = tid >> 2, part = tid & 3;
        const int Aq = blockIdx.x;
        const double2* __restrict__ src = reinterpret_cast<const double2*>(H.Winv[0] + ((size_t)(Aq < n1 ? Aq : 0) * 48 + out) * 48 + part * 12);
#pragma unroll
        for (int c = 0; c < 6; c++) { const double2 v = src[c]; w0[2 * c] = v.x; w0[2 * c + 1] = v.y; }
    } else {
#pragma unroll
        for (int c = 0; c < 6; c++) w0[c] = 0.;
        if (act) {
            const int rl = a % kMlFanout;
            const double* __restrict__ src = H.Winv[0] + ((size_t)(a / kMlFanout) * 48 + rl * 6 + r) * 48 + rl * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) w0[c] = src[c];
        }
    }
    // level-1 smoother: the own aggregates' rows of W1^-1 (sibling block of the level-2 parent): (6 AGG) outputs x kFan2 parts
    double g1own[3] = {0, 0, 0}, w1[6] = {0, 0, 0, 0, 0, 0}, g1x[3] = {0, 0, 0};
    const int A1 = blockIdx.x * kAggPerBlk + tid / 6;
    if (Lt >= 2 && tid < kAggPerBlk * 6 && A1 < n1) {
        const double* gq = H.geo[1] + (size_t)A1 * 3;
        g1own[0] = gq[0]; g1own[1] = gq[1]; g1own[2] = gq[2];
    }
    const int o1 = tid / kFan2, part1 = tid % kFan2;
    const int A1x = blockIdx.x * kAggPerBlk + o1 / 6;
    const bool l1thr = Lt >= 2 && tid < kAggPerBlk * 6 * kFan2;
    const bool l1act = l1thr && A1x < n1;
    if (l1act) {
        const int m1 = 6 * kFan2, A2 = A1x / kFan2, rowin = (A1x % kFan2) * 6 + o1 % 6;
        const double* __restrict__ src = H.Winv[1] + ((size_t)A2 * m1 + rowin) * m1 + part1 * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) w1[c] = src[c];
        const double* gq = H.geo[1] + (size_t)A1x * 3;
        g1x[0] = gq[0]; g1x[1] = gq[1]; g1x[2] = gq[2];
    }
    const double rz = init ? 0. : D.scal[0];
    // gather-level residual and restricted Ap of ALL aggregates: first kGU x 192 values in registers
    // (every workgroup reads the same two vectors: started at the same element they would all pull the same cache line from the
    //  same L2 channel at the same moment - each starts at a different 1.5-KB piece instead)
    double rgreg[kGU], sgreg[kGU];
    const int urot = (blockIdx.x >> 3) % kGU;              // workgroups of one XCD (blockIdx.x mod 8) get all kGU offsets
#pragma unroll
    for (int u = 0; u < kGU; u++) {
        const int uu = (u + urot >= kGU) ? u + urot - kGU : u + urot;
        const int t = uu * kCgBlk + tid;
        rgreg[u] = (t < 6 * ng) ? (VPRE ? H.Vg[t] : rg_old[t]) : 0.;
        sgreg[u] = (!VPRE && !init && t < 6 * ng) ? sg_at<AGG>(H.Sg, gl, t) : 0.;
    }
    // small arrays (offsets of levels >= g: one contiguous blob in the arena; the own ancestor's top-inverse rows; the
    // own-chain sibling rows): staged through registers so that EVERY load is in flight before anything waits
    constexpr int kGeoU = 6, kTopU = 2, kChainLv = kMlMaxLevels - 2;
    const int g_tot = (!comp && Lt > gl) ? (goff[Lt - 1] + 3 * H.n[Lt - 1] - goff[gl]) : 0;
    const int top_n = n_top_rows * ntop;
    const size_t top_base = comp ? 0 : (size_t)((Lt == 1) ? blockIdx.x * kAggPerBlk * 6 : 6 * anc[Lt]) * ntop;
    double gv[kGeoU], tv[kTopU], cv[kChainLv][2];
    {
        const double* __restrict__ gsrc = H.geo[gl];
#pragma unroll
        for (int u = 0; u < kGeoU; u++) { const int t = u * kCgBlk + tid; gv[u] = (!comp && t < g_tot) ? gsrc[t] : 0.; }
#pragma unroll
        for (int u = 0; u < kTopU; u++) {
            const int t = u * kCgBlk + tid;
            tv[u] = (!comp && t < top_n && top_base + t < (size_t)ntop * ntop) ? H.top_inv[top_base + t] : 0.;
        }
#pragma unroll
        for (int q = 0; q < kChainLv; q++) {       // level l = q + 2; fan-out of levels >= 3 is kMlFanout: 48 x 48 sibling blocks
            const int l = q + 2;
            cv[q][0] = 0.; cv[q][1] = 0.;
            if (l < Lt && !comp) {
                const double* __restrict__ wl = H.Winv[l] + ((size_t)anc[l + 1] * 48 + (size_t)(anc[l] % kMlFanout) * 6) * 48;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int t = u * kCgBlk + tid;
                    if (t < kChain) cv[q][u] = (t < 288) ? wl[t] : H.geo[l][(size_t)anc[l] * 3 + (t - 288)];
                }
            }
        }
    }
    float4 yreg[YPRE ? kYU : 1];
    float2 ytail = make_float2(0.f, 0.f);
    if (YPRE) {                                   // issued last: everything the alpha path needs returns first
        const int row = tid >> 5, j = tid & 31, n6 = 6 * ng, n4 = n6 >> 2;
        const float* __restrict__ yrow = H.Cmat32 + ((size_t)blockIdx.x * 6 + (row < 6 ? row : 0)) * H.c32_stride;
        const float4* __restrict__ yr = reinterpret_cast<const float4*>(yrow);
#pragma unroll
        for (int u = 0; u < (YPRE ? kYU : 1); u++) {
            yreg[u] = yr[j + 32 * u];             // (past the row's end: never used; the buffer carries 16 KB of slack behind its last row)
        }
        if (n6 & 2) ytail = *reinterpret_cast<const float2*>(yrow + 4 * n4);
    }
    STAMP(0);      // 1: prefetch issue
    double alpha = 0.;
    bool bad = false;
    if (VPRE) {
        alpha = D.scal[9]; bad = D.scal[10] != 0.;
    } else if (!init) {
        const double pAp = block_sum_w<3>(part_fold<kCgBlk, 4>(D.part_a, n_part, tid, vpart), s3);     // barriers: everything above has landed
        bad = !(pAp > 0.);
        alpha = bad ? 0. : rz / pAp;
    }
    STAMP(0);      // 2: partial reduction
#pragma unroll
    for (int u = 0; u < kGU; u++) {
        const int uu = (u + urot >= kGU) ? u + urot - kGU : u + urot;
        const int t = uu * kCgBlk + tid;
        if (t < 6 * ng) dyn[t] = VPRE ? rgreg[u] : fma(-alpha, sgreg[u], rgreg[u]);          // (roff[gl] = 0)
    }
    for (int t0 = kGU * kCgBlk + tid; t0 < 6 * ng; t0 += 4 * kCgBlk) {     // graphs beyond 12k free vertices: four values per lane and round trip
        double ra[4], sa[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u * kCgBlk, tt = (t < 6 * ng) ? t : 0;
            ra[u] = VPRE ? H.Vg[tt] : rg_old[tt]; sa[u] = (VPRE || init) ? 0. : sg_at<AGG>(H.Sg, gl, tt);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u * kCgBlk;
            if (t < 6 * ng) dyn[t] = VPRE ? ra[u] : ra[u] - alpha * sa[u];
        }
    }
    if (!comp) {
#pragma unroll
        for (int u = 0; u < kGeoU; u++) { const int t = u * kCgBlk + tid; if (t < g_tot) dyn[goff[gl] + t] = gv[u]; }
        for (int t = kGeoU * kCgBlk + tid; t < g_tot; t += kCgBlk) dyn[goff[gl] + t] = H.geo[gl][t];      // very large graphs
#pragma unroll
        for (int u = 0; u < kTopU; u++) { const int t = u * kCgBlk + tid; if (t < top_n) dyn[top_off + t] = tv[u]; }
        for (int t = kTopU * kCgBlk + tid; t < top_n; t += kCgBlk)
            dyn[top_off + t] = (top_base + t < (size_t)ntop * ntop) ? H.top_inv[top_base + t] : 0.;
#pragma unroll
        for (int q = 0; q < kChainLv; q++) {
            if (q + 2 < Lt) {
#pragma unroll
                for (int u = 0; u < 2; u++) { const int t = u * kCgBlk + tid; if (t < kChain) dyn[chain_off + q * kChain + t] = cv[q][u]; }
            }
        }
    }
    __syncthreads();
    STAMP(0);      // 3: gather-level residual estimate
    if (comp) {
        // y_2[own aggregate] = Y_2[rows 6 A .. 6 A + 5] . r_2   (32 lanes per row, fixed summation order)
        const int row = tid >> 5, j = tid & 31, n6 = 6 * ng;
        double sacc = 0.;
        if (YPRE) {
            const double2* __restrict__ rr = reinterpret_cast<const double2*>(dyn);
            double s0 = 0., s1 = 0., s2 = 0., s3q = 0.;
            const int n4 = n6 >> 2;
#pragma unroll
            for (int u = 0; u < (YPRE ? kYU : 1); u++) {
                const int t = j + 32 * u;
                if (t < n4) {
                    const float4 y = yreg[u];
                    const double2 xa = rr[2 * t], xb = rr[2 * t + 1];
                    s0 = fma((double)y.x, xa.x, s0); s1 = fma((double)y.y, xa.y, s1); s2 = fma((double)y.z, xb.x, s2); s3q = fma((double)y.w, xb.y, s3q);
                }
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of at most four steps in registers at a time
            }
            if ((n6 & 2) && j == 0) {
                const double2 xa = rr[2 * n4];
                s0 += (double)ytail.x * xa.x; s1 += (double)ytail.y * xa.y;
            }
            s0 += s2; s1 += s3q;
            sacc = (row < 6) ? s0 + s1 : 0.;
        } else if (row < 6) {
            // Y_2 as f32 (H.Cmat32), 16-byte loads = four columns, eight in flight per lane; accumulation in f64.  At 20k vertices the
            // f64 operator (112 MB, and a second hierarchy copy beside it) did not fit the Infinity Cache; half of it does.
            const float* __restrict__ yrow = H.Cmat32 + ((size_t)blockIdx.x * 6 + row) * H.c32_stride;
            const float4* __restrict__ yr = reinterpret_cast<const float4*>(yrow);
            const double2* __restrict__ rr = reinterpret_cast<const double2*>(dyn);          // roff[gl] = 0
            double s0 = 0., s1 = 0., s2 = 0., s3q = 0.;
            const int n4 = n6 >> 2;
#pragma unroll 16
            for (int t = j; t < n4; t += 32) {
                const float4 y = yr[t];
                const double2 xa = rr[2 * t], xb = rr[2 * t + 1];
                s0 = fma((double)y.x, xa.x, s0); s1 = fma((double)y.y, xa.y, s1); s2 = fma((double)y.z, xb.x, s2); s3q = fma((double)y.w, xb.y, s3q);
            }
            if ((n6 & 2) && j == 0) {                 // 6 n_2 is even: at most one pair beyond the last full quad
                const float2 y = *reinterpret_cast<const float2*>(yrow + 4 * n4);
                const double2 xa = rr[2 * n4];
                s0 += (double)y.x * xa.x; s1 += (double)y.y * xa.y;
            }
            s0 += s2; s1 += s3q;
            sacc = s0 + s1;
        }
        sacc = xsum16(sacc); sacc += __shfl_xor(sacc, 16);
        if (row < 6 && j == 0) syc[row] = sacc;
        __syncthreads();
    }
    // ---- restrict up to the top level: 8 lanes per (parent, component), one child each, xor-shuffle fold
    for (int l = gl + 1; l <= Lt && !comp; l++) {
        const int nC = H.n[l - 1], nP = H.n[l], fan = H.fan[l];
        const int tasks = nP * 6 * 8;
        for (int t0 = 0; t0 < tasks; t0 += kCgBlk) {
            const int t = t0 + tid;
            const int j = t & 7, ak = t >> 3;
            const int A = ak / 6, k = ak % 6;
            const int c = A * fan + j;
            double sacc = 0.;
            if (t < tasks && j < fan && c < nC) sacc = restrict_comp(dyn + goff[l - 1] + c * 3, dyn + roff[l - 1] + c * 6, k);
            sacc = xsum8(sacc);
            if (t < tasks && j == 0) dyn[roff[l] + ak] = sacc;
        }
        __syncthreads();
    }
    // (COMP: none of the offset tables is used - with computed indices they would live in scratch memory)
    const double* rtop = dyn + (comp ? 0 : roff[Lt]);
    if (!comp && Lt == 1) {
        if (tid < kAggPerBlk * 6) {
            double sacc = 0.;
            for (int c = 0; c < ntop; c++) sacc += dyn[top_off + tid * ntop + c] * rtop[c];
            sy[tid] = (A1 < n1) ? sacc : 0.;
        }
    } else if (!comp) {
        {   // 8 lanes per top row
            const int row = tid >> 3, j = tid & 7;
            double sacc = 0.;
            if (row < 6) for (int c = j; c < ntop; c += 8) sacc += dyn[top_off + row * ntop + c] * rtop[c];
            sacc = xsum8(sacc);
            if (row < 6 && j == 0) syc[row] = sacc;
        }
        __syncthreads();
        for (int l = Lt - 1; l >= 2; l--) {          // y_l = W_l^-1 [own ancestor's rows] r_l[siblings] + P_{l+1} y_{l+1}
            const double* ch = dyn + chain_off + (l - 2) * kChain;
            const int k = tid >> 3, part = tid & 7, sib = anc[l + 1] * kMlFanout + part;
            double sacc = 0.;
            if (tid < 48 && sib < H.n[l]) {
                const double* rr = dyn + roff[l] + sib * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) sacc += ch[k * 48 + part * 6 + c] * rr[c];
            }
            sacc = xsum8(sacc);
            if (tid < 48 && part == 0) sacc += prolong_comp(ch + 288, syc, k);
            __syncthreads();
            if (tid < 48 && part == 0) syc[k] = sacc;
            __syncthreads();
        }
    }
    STAMP(0);      // 4: restrict + top + down chain
    // ---- own rows: x, r, block-Jacobi part, exact r1 / r2 of the own aggregates
    double rv = rv0;
    if (act && !init) {
        const size_t i = (size_t)a * 6 + r;
        rv = fma(-alpha, apv, rv0);
        const double xn = fma(alpha, pv, xv);
        D.x[i] = xn;
        D.r[i] = rv;
        if (look) { D.xs[i] = xn; xsv = fabs(xn - xsv) * unit; }   // (xsv: from here on the distance moved since the last look)
    }
    if (look) look_partials((act && !init) ? xsv : 0., act ? rv * rv : 0., tid, spm);
    sv[tid] = act ? rv : 0.;
    __syncthreads();
    double zz = 0., w = 0.;
    if (AGG == 1) {   // zJ = W0^-1 r over the own aggregate: 4 partial sums per output row
        const int part = tid & 3;
        double ps = 0.;
#pragma unroll
        for (int c = 0; c < 12; c++) ps = fma(w0[c], sv[part * 12 + c], ps);
        ps = xsum4(ps);
        if (part == 0) szj[tid >> 2] = ps;
    } else if (act) {
        const int g0 = tid - r;
#pragma unroll
        for (int c = 0; c < 6; c++) zz = fma(w0[c], sv[g0 + c], zz);
    }
    if (act) {
        const int g0 = tid - r;
        w = p1t_comp(geo, sv[g0], sv[g0 + 1], sv[g0 + 2], sv[g0 + 3], sv[g0 + 4], sv[g0 + 5], r);
    }
    sw[tid] = w;
    __syncthreads();
    if (AGG == 1 && act) zz = szj[tid];
    STAMP(0);      // 5: x, r, zJ, w
    if (tid < kAggPerBlk * 6) {
        const int la = tid / 6, k = tid % 6;
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[(la * kMlFanout + j) * 6 + k];
        sr1[tid] = s;
        if (gl == 1 && A1 < n1) rg_new[(size_t)A1 * 6 + k] = s;          // exact, for the next iteration's recursion
    }
    __syncthreads();
    STAMP(0);      // 6: exact r1
    if (Lt >= 2) {
        double c2 = 0.;
        if (tid < kAggPerBlk * 6 && A1 < n1) c2 = restrict_comp(g1own, sr1 + (tid / 6) * 6, tid % 6);   // this child's share of the exact r2
        if (gl == 2) {
            c2 += __shfl_down(c2, 12);                                     // lanes (la, k): fold la = 0..3
            c2 += __shfl_down(c2, 6);
            if (tid < 6) rg_new[(size_t)blockIdx.x * 6 + tid] = c2;       // exact r2 of the own aggregate
        }
        // y1 = W1^-1 [own rows] r1[siblings] + P2 y2.  AGG = 1: the siblings' r1 come from the gather-level vector in
        // LDS (own aggregate: the exact value); AGG = 4: the four siblings are this workgroup's own aggregates.
        double ps = 0.;
        if (l1act) {
            const int sib = (A1x / kFan2) * kFan2 + part1;
            const double* rs = (AGG == 1) ? ((sib == A1x) ? sr1 : dyn + roff[1] + (size_t)sib * 6) : (sr1 + part1 * 6);
            if (AGG != 1 || sib < n1) {
#pragma unroll
                for (int c = 0; c < 6; c++) ps = fma(w1[c], rs[c], ps);
            }
        }
        ps = xsum4(ps);
        if (kFan2 == 8) ps += dpp_mov_f64<0x141, 0xf>(0., ps);
        if (l1thr && part1 == 0) sy[o1] = l1act ? ps + prolong_comp(g1x, syc, o1 % 6) : 0.;
    }
    __syncthreads();
    double acc = 0.;
    if (act) {
        zz += p1_comp(geo, sy + ((tid / 6) / kMlFanout) * 6, r);
        D.z[(size_t)a * 6 + r] = zz;
        acc = rv * zz;
    }
    STAMP(0);      // 7: y1, z
    const double tot = block_sum_w<3>(acc, s3);
    if (tid == 0) {
        D.part_b[blockIdx.x] = tot;
        if (look) look_store(D.part_c, blockIdx.x, spm);
        if (blockIdx.x == 0 && !init) {
            D.scal[2] = rz;
            D.flags[1] += 1;
            if (bad) { D.flags[0] = 1; D.flags[2] = 1; }
        }
    }
    STAMP(0);      // 8: block sum + stores
#ifdef UZL_STAMPS
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_stamps[31], 1ull);
#endif
}
template <int AGG, bool COMP = false, bool YPRE = false, bool VPRE = false>
__global__ __launch_bounds__(kCgBlk) __attribute__((amdgpu_waves_per_eu(COMP ? (YPRE ? 2 : 3) : 1))) void ml_cg_kernel(PgoDev D, MlHot H, const double* __restrict__ p,
                                                      const double* __restrict__ rg_old, double* __restrict__ rg_new,
                                                      int n_part, int init)
{
    ml_cg_kernel_body<AGG, COMP, YPRE, VPRE>(D, H, p, rg_old, rg_new, n_part, init);
}

// alpha = r.z / p.Ap and v = rg - alpha Sg for ml_cg_kernel<4, true, false, true>: every workgroup sums the p.Ap partials (same order,
// same alpha), each writes 256 entries of v; workgroup 0 leaves alpha and the breakdown flag in scal[9], scal[10].
__device__ __forceinline__ void ml_alpha_kernel_body(PgoDev D, MlHot H, const double* __restrict__ rg_old, int n_part)
{
    __shared__ double s4[4];
    const int done = D.flags[0];                  // checked behind the loads: 15 workgroups, nothing to save by leaving before them
    const int tid = threadIdx.x, t = blockIdx.x * 256 + tid, n6 = 6 * H.n[2];
    double vp[4];
    part_issue<256, 4>(D.part_a, n_part, tid, vp);
    double rgv = 0., sgv = 0.;
    if (t < n6) { rgv = rg_old[t]; sgv = sg_at<4>(H.Sg, 2, t); }
    const double rz = D.scal[0];
    if (done) return;
    const double pAp = block_sum_w<4>(part_fold<256, 4>(D.part_a, n_part, tid, vp), s4);
    const bool bad = !(pAp > 0.);
    const double alpha = bad ? 0. : rz / pAp;
    if (t < n6) H.Vg[t] = rgv - alpha * sgv;
    if (blockIdx.x == 0 && tid == 0) { D.scal[9] = alpha; D.scal[10] = bad ? 1. : 0.; }
}
__global__ __launch_bounds__(256) void ml_alpha_kernel(PgoDev D, MlHot H, const double* __restrict__ rg_old, int n_part) { ml_alpha_kernel_body(D, H, rg_old, n_part); }

// ------------------------------------------------------------------------------------------------
// ml_cg for small graphs (<= 1280 free vertices, one level-1 aggregate per workgroup): the hierarchy above level 1
// has been folded into the dense operator Y_1 (ml_dense_level_kernel), so the coarse correction of the own aggregate
// is  y1 = Y_1[rows 6A..6A+5] (rg_old - alpha Sg): every operand is loaded at entry (one memory latency), the
// restrict / top-solve / prolong walk through LDS and its barriers are gone.  Same preconditioner, same results up to
// rounding as ml_cg_kernel<1>.
// ------------------------------------------------------------------------------------------------
// kCompU gather-level values per lane: 5 covers 6 n_1 <= 960 (<= 1280 free vertices), 8 covers 6 n_1 <= 1536 (<= 2048), 12 <= 2304 (3072), 16 <= 3072 (4096)
template <int kCompU, bool kLds = false>
__device__ __forceinline__ void ml_cg_comp_kernel_body(PgoDev D, MlHot H, const double* __restrict__ p,
                                                           const double* __restrict__ rg_old, double* __restrict__ rg_new,
                                                           int n_part, int init)
{
    __shared__ double s3[3];
    __shared__ double sv[kCgBlk];
    __shared__ double sw[kCgBlk];
    __shared__ double sy[6];
    __shared__ double szj[48];
    __shared__ double scomp[3][6];
    __shared__ double spm[6];
    if (D.flags[0]) return;
    const int tid = threadIdx.x;
    const int a = blockIdx.x * kMlFanout + tid / 6, r = tid % 6;
    const bool act = tid < 48 && a < D.nb;
    const int n1 = H.n[1], ng6 = 6 * n1;
    // ---- every global load, before any barrier
    double vpart[4] = {0., 0., 0., 0.};
    if (!init) part_issue<kCgBlk, 4>(D.part_a, n_part, tid, vpart);
    double xv = 0., xsv = 0., rv0 = 0., apv = 0., pv = 0., geo[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int look = init ? 1 : D.flags[3];       // the stop test wants this workgroup's partials (ml_spmv said so; always with r = b: look_partials)
    const double unit = progress_unit(D.scal, r);
    if (act) {
        const size_t i = (size_t)a * 6 + r;
        rv0 = D.r[i];
        if (!init) { xv = D.x[i]; xsv = D.xs[i]; apv = D.ap[i]; pv = p[i]; }
        const double* __restrict__ gg = H.geo0 + (size_t)a * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
    double w0[12];
    {
        const int out = tid >> 2, part4 = tid & 3;
        const double2* __restrict__ src = reinterpret_cast<const double2*>(H.Winv[0] + ((size_t)blockIdx.x * 48 + out) * 48 + part4 * 12);
#pragma unroll
        for (int c = 0; c < 6; c++) { const double2 v = src[c]; w0[2 * c] = v.x; w0[2 * c + 1] = v.y; }
    }
    const double rz = init ? 0. : D.scal[0];
    double rgreg[kCompU], sgreg[kCompU];
    float cm[6][kCompU];                   // six rows of Y_1 (f32 copy): half the bytes and half the registers of the f64 operator
    const int cst = H.c32_stride;
    const float* __restrict__ crow = H.Cmat32 + (size_t)blockIdx.x * 6 * cst;
#pragma unroll
    for (int u = 0; u < kCompU; u++) {
        const int t = u * kCgBlk + tid;
        const bool in = t < ng6;
        rgreg[u] = in ? rg_old[t] : 0.;
        sgreg[u] = (!init && in) ? H.Sg[t] : 0.;
#pragma unroll
        for (int q = 0; q < 6; q++) { const float y = crow[(size_t)q * cst + (in ? t : 0)]; cm[q][u] = in ? y : 0.f; }   // unconditional loads: all 30 in flight together
    }
    double alpha = 0.;
    bool bad = false;
    if (!init) {
        const double pAp = block_sum_w<3, kLds>(part_fold<kCgBlk, 4>(D.part_a, n_part, tid, vpart), s3);
        bad = !(pAp > 0.);
        alpha = bad ? 0. : rz / pAp;
    }
    // ---- coarse correction of the own aggregate: 6 rows of Y_1 times the gather-level residual estimate
    {
        double ps[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < kCompU; u++) {
            const double v = fma(-alpha, sgreg[u], rgreg[u]);
#pragma unroll
            for (int q = 0; q < 6; q++) ps[q] = fma((double)cm[q][u], v, ps[q]);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) ps[q] = wave_sum<kLds>(ps[q]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 6; q++) scomp[tid >> 6][q] = ps[q];
        }
    }
    // ---- own rows: x, r, level-0 smoother, exact r1 of the own aggregate
    double rv = rv0;
    if (act && !init) {
        const size_t i = (size_t)a * 6 + r;
        rv = fma(-alpha, apv, rv0);
        const double xn = fma(alpha, pv, xv);
        D.x[i] = xn;
        D.r[i] = rv;
        if (look) { D.xs[i] = xn; xsv = fabs(xn - xsv) * unit; }
    }
    if (look) look_partials<kLds>((act && !init) ? xsv : 0., act ? rv * rv : 0., tid, spm);
    sv[tid] = act ? rv : 0.;
    __syncthreads();
    double zz = 0., w = 0.;
    {
        const int part4 = tid & 3;
        double ps = 0.;
#pragma unroll
        for (int c = 0; c < 12; c++) ps = fma(w0[c], sv[part4 * 12 + c], ps);
        ps = xsum4(ps);
        if (part4 == 0) szj[tid >> 2] = ps;
    }
    if (act) {
        const int g0 = tid - r;
        w = p1t_comp(geo, sv[g0], sv[g0 + 1], sv[g0 + 2], sv[g0 + 3], sv[g0 + 4], sv[g0 + 5], r);
    }
    sw[tid] = w;
    if (tid < 6) sy[tid] = (scomp[0][tid] + scomp[1][tid]) + scomp[2][tid];
    __syncthreads();
    if (act) zz = szj[tid];
    if (tid < 6) {
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[j * 6 + tid];
        if ((int)blockIdx.x < n1) rg_new[(size_t)blockIdx.x * 6 + tid] = s;       // exact r1: the recursion never accumulates error
    }
    double acc = 0.;
    if (act) {
        zz += p1_comp(geo, sy, r);
        D.z[(size_t)a * 6 + r] = zz;
        acc = rv * zz;
    }
    const double tot = block_sum_w<3, kLds>(acc, s3);
    if (tid == 0) {
        D.part_b[blockIdx.x] = tot;
        if (look) look_store(D.part_c, blockIdx.x, spm);
        if (blockIdx.x == 0 && !init) {
            D.scal[2] = rz;
            D.flags[1] += 1;
            if (bad) { D.flags[0] = 1; D.flags[2] = 1; }
        }
    }
}
template <int kCompU>
__global__ __launch_bounds__(kCgBlk) void ml_cg_comp_kernel(PgoDev D, MlHot H, const double* __restrict__ p, const double* __restrict__ rg_old, double* __restrict__ rg_new, int n_part, int init)
{
    ml_cg_comp_kernel_body<kCompU>(D, H, p, rg_old, rg_new, n_part, init);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void k_ml_geometry(const PgoDev& D, const MlDev* ml, const double* pose, int l, int n_l, hipStream_t s)
{
    hipLaunchKernelGGL(ml_geometry_kernel, dim3(l == 0 ? 1 : (n_l + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D, ml, pose, l);
}
void k_ml_galerkin(const PgoDev& D, const MlDev* ml, int f, int n_chunks, hipStream_t s)
{
    if (n_chunks > 0) hipLaunchKernelGGL(ml_galerkin_kernel, dim3(n_chunks), dim3(kBlk), 0, s, D, ml, f);
}
void k_ml_dense_level(const MlDev* ml, int l, int n_l, hipStream_t s)
{
    hipLaunchKernelGGL(ml_dense_level_kernel, dim3((n_l * n_l + kBlk - 1) / kBlk), dim3(kBlk), 0, s, ml, l);
}
void k_ml_mult_level(const PgoDev& D, const MlDev* ml, int lev, int n1, int n2, hipStream_t s)
{
    hipLaunchKernelGGL(ml_mult_pair_kernel, dim3(n2 * n2), dim3(kBlk), 0, s, D, ml, lev);
    const int g12r = (n1 * n2 * 6 + kBlk - 1) / kBlk;
    hipLaunchKernelGGL(ml_mult_qy_kernel, dim3(g12r), dim3(kBlk), 0, s, ml, lev);
    const int gt = (6 * n1 + kGemmTile - 1) / kGemmTile;
    hipLaunchKernelGGL(ml_mult_qyqt_kernel, dim3(gt * (gt + 1) / 2), dim3(256), 0, s, ml, lev);
}
// one Newton-Schulz step at level `lev`: Xn = 2 X - X (A_lev X); T is scratch
void k_ml_ns_step(const PgoDev& D, const MlDev* ml, int lev, int n1, const double* X, double* T, double* Xn, hipStream_t s,
                  hipEvent_t ev_a, hipEvent_t ev_b, float* c32, int c32_stride)
{
    const int n6 = 6 * n1;
    hipLaunchKernelGGL(ml_ns_ax_kernel, dim3(kXcds * n1 * ax_parts(n6)), dim3(kBlk), 0, s, D, ml, lev, X, T);
    const int g = (n6 + kGemmTile - 1) / kGemmTile, gtri = g * (g + 1) / 2;     // tiles on and above the diagonal
    if (n6 <= kGemm32Max) {                                                       // one small graph: 32 x 32 tiles, K split over the waves (same bits)
        const int g32 = gemm32_grid((n6 + 31) / 32);
        if (ev_a) hipExtLaunchKernelGGL(ml_ns_gemm32_kernel, dim3(g32), dim3(256), 0, s, ev_a, ev_b, 0, n6, X, T, Xn, c32, c32_stride);
        else hipLaunchKernelGGL(ml_ns_gemm32_kernel, dim3(g32), dim3(256), 0, s, n6, X, T, Xn, c32, c32_stride);
        return;
    }
    if (ev_a) hipExtLaunchKernelGGL(ml_ns_gemm_kernel, dim3(gtri), dim3(256), 0, s, ev_a, ev_b, 0, n6, X, T, Xn, c32, c32_stride);     // dispatch timestamps of the GEMM alone
    else hipLaunchKernelGGL(ml_ns_gemm_kernel, dim3(gtri), dim3(256), 0, s, n6, X, T, Xn, c32, c32_stride);
}
void k_ml_cmat32(const MlHot& hot, int n6, hipStream_t s)
{
    if (!hot.Cmat || !hot.Cmat32) return;
    const long work = (long)n6 * (hot.c32_stride >> 2);
    hipLaunchKernelGGL(ml_cmat32_kernel, dim3((unsigned)((work + kBlk - 1) / kBlk)), dim3(kBlk), 0, s, hot.Cmat, const_cast<float*>(hot.Cmat32), n6, hot.c32_stride);
}
void k_ml_sibling(const PgoDev& D, const MlDev* ml, int total_aggs, hipStream_t s)
{
    hipLaunchKernelGGL(ml_inverses_kernel, dim3((total_aggs + kSibPerBlk - 1) / kSibPerBlk + 1), dim3(kBlk), 0, s, D, ml);
}
int g_ml_rows(int nb, int agg) { return (nb + kMlFanout * agg - 1) / (kMlFanout * agg); }
// workgroups of ml_spmv (= p.Ap partials ml_cg sums): AGG = 4 runs two half workgroups per level-2 aggregate
int g_ml_spmv(int nb, int agg) { return agg == 1 ? g_ml_rows(nb, 1) : g_ml_rows(nb, agg) * (8 / kSpmvWaves4); }
// dynamic LDS of ml_cg_kernel for a hierarchy (n[0..levels]) and workgroup geometry agg
size_t ml_cg_lds_bytes(const int* n, int levels, int agg)
{
    const int g = (agg == 1 || levels < 2) ? 1 : 2;
    size_t d = 0;
    for (int l = g; l <= levels; l++) d += 6 * (size_t)n[l];
    for (int l = g; l < levels; l++) d += 3 * (size_t)n[l];
    const size_t ntop = 6 * (size_t)n[levels];
    d += ((levels == 1) ? (size_t)agg * 6 : 6) * ntop;
    if (levels > 2) d += (size_t)(levels - 2) * kChain;
    return d * 8;
}
bool ml_fits_lds(const int* n_per_level, int levels, int agg)
{
    return ml_cg_lds_bytes(n_per_level, levels, agg) <= kMlLdsLimit && g_ml_spmv(n_per_level[0], agg) <= kMaxPartials;
}
bool ml_comp4_fits(int nb, int n2) { return ml_comp4_lds(n2) <= kMlLdsLimit && g_ml_spmv(nb, 4) <= kMaxPartials; }
void k_ml_init(const PgoDev& D, const MlHot& ml, int agg, double* p0, double* p1, double* rg, hipStream_t s)
{
    if (agg == 1) hipLaunchKernelGGL(ml_init_kernel<1>, dim3(g_ml_rows(D.nb, 1)), dim3(kCgBlk), 0, s, D, ml, p0, p1, rg);
    else hipLaunchKernelGGL(ml_init_kernel<4>, dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), 0, s, D, ml, p0, p1, rg);
}
// ev_a / ev_b (profiling only): the dispatch's own start / stop timestamps
void k_ml_spmv(const PgoDev& D, const MlHot& ml, int agg, const double* p_old, double* p_new, int n_part, double tol2, hipStream_t s,
               hipEvent_t ev_a, hipEvent_t ev_b)
{
    if (ev_a) {
        if (agg == 1) hipExtLaunchKernelGGL(ml_spmv_kernel<1>, dim3(g_ml_rows(D.nb, 1)), dim3(512), 0, s, ev_a, ev_b, 0, D, ml, p_old, p_new, n_part, tol2);
        else hipExtLaunchKernelGGL(ml_spmv_kernel<4>, dim3(g_ml_spmv(D.nb, 4)), dim3(64 * kSpmvWaves4), 0, s, ev_a, ev_b, 0, D, ml, p_old, p_new, n_part, tol2);
        return;
    }
    if (agg == 1) hipLaunchKernelGGL(ml_spmv_kernel<1>, dim3(g_ml_rows(D.nb, 1)), dim3(512), 0, s, D, ml, p_old, p_new, n_part, tol2);
    else hipLaunchKernelGGL(ml_spmv_kernel<4>, dim3(g_ml_spmv(D.nb, 4)), dim3(64 * kSpmvWaves4), 0, s, D, ml, p_old, p_new, n_part, tol2);
}
hipError_t k_ml_cg(const PgoDev& D, const MlHot& ml, int agg, const double* p, const double* rg_old, double* rg_new, int n_part,
                   int init, size_t lds, hipStream_t s, hipEvent_t ev_a, hipEvent_t ev_b)
{
    // largest dynamic-LDS size each kernel variant has been raised to, PER DEVICE (a function attribute is per device), under a lock
    // (handles of several threads / devices share this table)
    constexpr int kMaxDev = 16;
    static size_t configured_tab[kMaxDev][5] = {};
    static std::mutex configured_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    const bool comp4 = agg != 1 && ml.Cmat != nullptr;
    const bool ypre = comp4 && ml.levels >= 2 && 6 * ml.n[2] <= 4 * 32 * kYU;       // the six rows of Y_2 fit the registers
    static const bool no_vpre = diag_flag("UZL_NO_VPRE");                            // A/B switch (diagnostic build)
    const bool vpre = comp4 && !ypre && !init && ml.Vg != nullptr && !no_vpre;      // alpha and rg - alpha Sg prepared once, by ml_alpha_kernel
    // COMP stages nothing but the gather-level vector: asking for the LDS of the full restrict / top / chain walk (52 KB at 20k
    // vertices) held the kernel at two workgroups per CU - 625 workgroups ran in two rounds
    if (comp4) lds = ml_comp4_lds(ml.n[2]);
    if (lds > kMlLdsLimit) return hipErrorInvalidValue;                             // (build_ml admits no such hierarchy)
    const int ci = agg == 1 ? 0 : (comp4 ? (ypre ? 3 : (vpre ? 4 : 2)) : 1);
    std::unique_lock<std::mutex> cfg_lock(configured_mu);
    size_t* configured = configured_tab[dev];
    if (lds > configured[ci]) {
        const void* fn = agg == 1 ? reinterpret_cast<const void*>(&ml_cg_kernel<1>)
                                  : (comp4 ? (ypre ? reinterpret_cast<const void*>(&ml_cg_kernel<4, true, true>)
                                                   : (vpre ? reinterpret_cast<const void*>(&ml_cg_kernel<4, true, false, true>) : reinterpret_cast<const void*>(&ml_cg_kernel<4, true>)))
                                           : reinterpret_cast<const void*>(&ml_cg_kernel<4>));
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[ci] = lds;
    }
    cfg_lock.unlock();
    if (agg == 1 && ml.Cmat) {           // small graphs: composite coarse operator
        // columns of the dense level-1 operator a lane holds: 5 (6 n_1 <= 960), 8 (<= 1536), 12 (<= 2304), 16 (<= 3072: 4096 free vertices)
        const int cols = 6 * ml.n[1];
        const dim3 g(g_ml_rows(D.nb, 1)), t(kCgBlk);
#define UZL_COMP_LAUNCH(U)                                                                                                              \
        do { if (ev_a) hipExtLaunchKernelGGL(ml_cg_comp_kernel<U>, g, t, 0, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);  \
             else hipLaunchKernelGGL(ml_cg_comp_kernel<U>, g, t, 0, s, D, ml, p, rg_old, rg_new, n_part, init); } while (0)
        if (cols <= 5 * kCgBlk) UZL_COMP_LAUNCH(5);
        else if (cols <= 8 * kCgBlk) UZL_COMP_LAUNCH(8);
        else if (cols <= 12 * kCgBlk) UZL_COMP_LAUNCH(12);
        else UZL_COMP_LAUNCH(16);
#undef UZL_COMP_LAUNCH
        return hipSuccess;
    }
    if (ev_a) {
        if (agg == 1) hipExtLaunchKernelGGL(ml_cg_kernel<1>, dim3(g_ml_rows(D.nb, 1)), dim3(kCgBlk), lds, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);
        else if (ypre) hipExtLaunchKernelGGL((ml_cg_kernel<4, true, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);
        else if (vpre) {
            hipLaunchKernelGGL(ml_alpha_kernel, dim3((6 * ml.n[2] + 255) / 256), dim3(256), 0, s, D, ml, rg_old, n_part);
            hipExtLaunchKernelGGL((ml_cg_kernel<4, true, false, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);
        }
        else if (comp4) hipExtLaunchKernelGGL((ml_cg_kernel<4, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);
        else hipExtLaunchKernelGGL(ml_cg_kernel<4>, dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, ev_a, ev_b, 0, D, ml, p, rg_old, rg_new, n_part, init);
        return hipSuccess;
    }
    if (agg == 1) hipLaunchKernelGGL(ml_cg_kernel<1>, dim3(g_ml_rows(D.nb, 1)), dim3(kCgBlk), lds, s, D, ml, p, rg_old, rg_new, n_part, init);
    else if (ypre) hipLaunchKernelGGL((ml_cg_kernel<4, true, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, D, ml, p, rg_old, rg_new, n_part, init);
    else if (vpre) {
        hipLaunchKernelGGL(ml_alpha_kernel, dim3((6 * ml.n[2] + 255) / 256), dim3(256), 0, s, D, ml, rg_old, n_part);
        hipLaunchKernelGGL((ml_cg_kernel<4, true, false, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, D, ml, p, rg_old, rg_new, n_part, init);
    }
    else if (comp4) hipLaunchKernelGGL((ml_cg_kernel<4, true>), dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, D, ml, p, rg_old, rg_new, n_part, init);
    else hipLaunchKernelGGL(ml_cg_kernel<4>, dim3(g_ml_rows(D.nb, 4)), dim3(kCgBlk), lds, s, D, ml, p, rg_old, rg_new, n_part, init);
    return hipSuccess;
}


// ------------------------------------------------------------------------------------------------
// slot twins of the device-resident LM loop (pgo_types.hpp: LmSlot / LmDev; uzl_pgo_lm.hip): graph = blockIdx.z, arguments from its
// slot, every kernel predicated on the graph's LM state.  Same bodies as the by-value kernels above: same arithmetic, same bits.
// ------------------------------------------------------------------------------------------------
// Set-up kernels serve two segments of a pass (`which`):
//   0 = rebuild of copy LmDev::build_ix AHEAD of the trial loop (second stream): numeric + trial part, lambda from LmDev::scal2, poses
//       of buffer build_cur - all three snapshots lm_head_kernel took, because the main stream moves cur / ix on while this runs;
//   1 = set-up of the copy in use: numeric part in the pass stamped numeric_pass, trial part in the pass stamped trial_pass.
#define UZL_LM_SETUP(NUMERIC)                                                                             \
    const LmSlot& S = slots[blockIdx.z];                                                                  \
    const LmDev* lm = S.lm;                                                                               \
    if ((which == 0 ? lm->build_pass : ((NUMERIC) ? lm->numeric_pass : lm->trial_pass)) != lm->pass) return;      \
    const int c = which == 0 ? lm->build_ix : lm->ix;                                                     \
    PgoDev D = S.Dp;                                                                                      \
    if (which == 0) D.scal = const_cast<double*>(lm->scal2);

__global__ __launch_bounds__(kBlk) void ml_geometry_lm_kernel(const LmSlot* __restrict__ slots, int which, int l)
{
    UZL_LM_SETUP(true)
    ml_geometry_kernel_body(D, S.dml[c], S.pose[which == 0 ? lm->build_cur : lm->cur], l);
}
__global__ __launch_bounds__(kBlk) void ml_galerkin_lm_kernel(const LmSlot* __restrict__ slots, int which, int f)
{
    UZL_LM_SETUP(true)
    ml_galerkin_kernel_body(D, S.dml[c], f);
}
__global__ __launch_bounds__(kBlk) void ml_inverses_lm_kernel(const LmSlot* __restrict__ slots, int which)
{
    UZL_LM_SETUP(false)
    ml_inverses_kernel_body(D, S.dml[c]);
}
__global__ __launch_bounds__(kBlk) void ml_dense_level_lm_kernel(const LmSlot* __restrict__ slots, int which, int l)
{
    UZL_LM_SETUP(false)
    (void)D;
    ml_dense_level_kernel_body(S.dml[c], l);
}
__global__ __launch_bounds__(kBlk) void ml_mult_pair_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev)
{
    UZL_LM_SETUP(false)
    ml_mult_pair_kernel_body(D, S.dml[c], lev);
}
__global__ __launch_bounds__(kBlk) void ml_mult_qy_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev)
{
    UZL_LM_SETUP(false)
    (void)D;
    ml_mult_qy_kernel_body(S.dml[c], lev);
}
__global__ __launch_bounds__(256) void ml_mult_qyqt_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev)
{
    UZL_LM_SETUP(false)
    (void)D;
    { const int gt = (6 * S.hot[0].n[lev] + kGemmTile - 1) / kGemmTile; if ((int)blockIdx.x >= gt * (gt + 1) / 2) return; }      // (a batch launches the largest graph's grid)
    ml_mult_qyqt_kernel_body(S.dml[c], lev);
}
// Newton-Schulz step k at level lev: X ping-pongs between Ydense[lev] and nsX, starting in Ydense[lev]
__global__ __launch_bounds__(kBlk) void ml_ns_ax_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev, int k)
{
    UZL_LM_SETUP(false)
    const bool by_xcd = gridDim.z == 1;
    if (by_xcd && (int)blockIdx.x >= kXcds * S.hot[0].n[lev] * ax_parts(6 * S.hot[0].n[lev])) return;
    const double* X = (k & 1) ? S.nsX[c] : S.dense[c][lev];
    ml_ns_ax_kernel_body(D, S.dml[c], lev, X, S.nsT[c], by_xcd);
}
__global__ __launch_bounds__(256) void ml_ns_gemm_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev, int k, int last)
{
    UZL_LM_SETUP(false)
    (void)D;
    const int n6 = 6 * S.hot[0].n[lev];
    { const int gt = (n6 + kGemmTile - 1) / kGemmTile; if ((int)blockIdx.x >= gt * (gt + 1) / 2) return; }
    const double* X = (k & 1) ? S.nsX[c] : S.dense[c][lev];
    double* Xn = (k & 1) ? S.dense[c][lev] : S.nsX[c];
    const MlHot& H = S.hot[c];
    ml_ns_gemm_kernel_body(n6, X, S.nsT[c], Xn, last ? const_cast<float*>(H.Cmat32) : nullptr, H.c32_stride);      // (last step: Xn = H.Cmat)
}
__global__ __launch_bounds__(256) void ml_ns_gemm32_lm_kernel(const LmSlot* __restrict__ slots, int which, int lev, int k, int last)
{
    UZL_LM_SETUP(false)
    (void)D;
    const int n6 = 6 * S.hot[0].n[lev];
    const double* X = (k & 1) ? S.nsX[c] : S.dense[c][lev];
    double* Xn = (k & 1) ? S.dense[c][lev] : S.nsX[c];
    const MlHot& H = S.hot[c];
    ml_ns_gemm32_kernel_body(n6, X, S.nsT[c], Xn, last ? const_cast<float*>(H.Cmat32) : nullptr, H.c32_stride);
}
__global__ __launch_bounds__(kBlk) void ml_cmat32_lm_kernel(const LmSlot* __restrict__ slots, int which, int cl)
{
    UZL_LM_SETUP(false)
    (void)D;
    const MlHot& H = S.hot[c];
    if (!H.Cmat || !H.Cmat32) return;
    const int n6 = 6 * H.n[cl];
    ml_cmat32_body(H.Cmat, const_cast<float*>(H.Cmat32), n6, H.c32_stride);
}

// numeric part (geometry, Galerkin products level by level): the launch sequence of ml_setup_numeric (uzl_pgo.hip)
void kl_ml_numeric(const LmSlot* sl, const LmShape& sh, int which, hipStream_t s)
{
    const int B = sh.nslots;
    if (sh.n_lv[1] <= kGeoAllMax) hipLaunchKernelGGL(ml_geometry_lm_kernel, dim3(1, 1, B), dim3(kBlk), 0, s, sl, which, 0);     // all levels, one workgroup per graph
    else
        for (int l = 1; l <= sh.levels; l++)
            hipLaunchKernelGGL(ml_geometry_lm_kernel, dim3((sh.n_lv[l] + kBlk - 1) / kBlk, 1, B), dim3(kBlk), 0, s, sl, which, l);
    for (int f = 0; f < sh.levels; f++)
        if (sh.chunks[f + 1] > 0) hipLaunchKernelGGL(ml_galerkin_lm_kernel, dim3(sh.chunks[f + 1], 1, B), dim3(kBlk), 0, s, sl, which, f);
}
// lambda-dependent part: the launch sequence of ml_setup_trial (uzl_pgo.hip)
void kl_ml_trial(const LmSlot* sl, const LmShape& sh, int which, hipStream_t s)
{
    const int B = sh.nslots, L = sh.levels, cl = sh.cl;
    hipLaunchKernelGGL(ml_inverses_lm_kernel, dim3((sh.inner_aggs + kSibPerBlk - 1) / kSibPerBlk + 1, 1, B), dim3(kBlk), 0, s, sl, which);
    if (cl == 0) return;                                                       // no dense operator
    const int n6c = 6 * sh.n_lv[cl];
    const long work32 = (long)n6c * (((n6c + 3) & ~3) >> 2);
    if (!sh.mult) {                                                            // additive operator: Y_l = blockdiag(W_l^-1) + P Y_{l+1} P^T
        for (int l = L - 1; l >= cl; l--)
            hipLaunchKernelGGL(ml_dense_level_lm_kernel, dim3((sh.n_lv[l] * sh.n_lv[l] + kBlk - 1) / kBlk, 1, B), dim3(kBlk), 0, s, sl, which, l);
        hipLaunchKernelGGL(ml_cmat32_lm_kernel, dim3((unsigned)((work32 + kBlk - 1) / kBlk), 1, B), dim3(kBlk), 0, s, sl, which, cl);
        return;
    }
    for (int l = L - 1; l >= cl; l--) {                                        // multiplicative cycle + Newton-Schulz, from the top down
        const int n1 = sh.n_lv[l], n2 = sh.n_lv[l + 1];
        hipLaunchKernelGGL(ml_mult_pair_lm_kernel, dim3(n2 * n2, 1, B), dim3(kBlk), 0, s, sl, which, l);
        const int g12r = (n1 * n2 * 6 + kBlk - 1) / kBlk;
        hipLaunchKernelGGL(ml_mult_qy_lm_kernel, dim3(g12r, 1, B), dim3(kBlk), 0, s, sl, which, l);
        const int n6 = 6 * n1, gt = (n6 + kGemmTile - 1) / kGemmTile;
        hipLaunchKernelGGL(ml_mult_qyqt_lm_kernel, dim3(gt * (gt + 1) / 2, 1, B), dim3(256), 0, s, sl, which, l);
        const int steps = l > cl ? sh.upper_ns : sh.ns_steps;
        for (int k = 0; k < steps; k++) {
            hipLaunchKernelGGL(ml_ns_ax_lm_kernel, dim3(B == 1 ? kXcds * n1 * ax_parts(n6) : n1 * ((n6 + kBlk - 1) / kBlk), 1, B), dim3(kBlk), 0, s, sl, which, l, k);
            const int last = (l == cl && k == steps - 1) ? 1 : 0;
            if (B == 1 && n6 <= kGemm32Max) hipLaunchKernelGGL(ml_ns_gemm32_lm_kernel, dim3(gemm32_grid((n6 + 31) / 32), 1, 1), dim3(256), 0, s, sl, which, l, k, last);
            else hipLaunchKernelGGL(ml_ns_gemm_lm_kernel, dim3(gt * (gt + 1) / 2, 1, B), dim3(256), 0, s, sl, which, l, k, last);
        }
    }
    if (sh.ns_steps == 0) hipLaunchKernelGGL(ml_cmat32_lm_kernel, dim3((unsigned)((work32 + kBlk - 1) / kBlk), 1, B), dim3(kBlk), 0, s, sl, which, cl);
}

// ---- PCG: init + the two iteration kernels.  SLOT = `const LmSlot*` (blockIdx.z picks the graph) or `LmSlot` BY VALUE for a pass of
// one graph - the slot then sits in the kernel-argument segment like the by-value kernels' arguments: no pointer hop in front of the
// first loads of kernels that are a chain of round trips.
__device__ __forceinline__ const LmSlot& slot_of(const LmSlot* __restrict__ slots) { return slots[blockIdx.z]; }
__device__ __forceinline__ const LmSlot& slot_of(const LmSlot& slot) { return slot; }

// The hot subset of hierarchy copy `ix`.  Both copies sit in one arena, `copy_stride` bytes apart, and share every size: copy 0's
// struct with its arena pointers moved - an add behind the ix load, where indexing hot[] with ix would be a second, dependent load of the slot
// (kernel-argument memory) in front of kernels that are chains of round trips already (10k/50k: 1.0 us per PCG iteration).
template <class T>
__device__ __forceinline__ const T* moved(const T* p, int64_t bytes) { return p ? reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + bytes) : nullptr; }
__device__ __forceinline__ MlHot hot_of(const LmSlot& S, int ix)
{
    MlHot H = S.hot[0];
    const int64_t off = ix ? S.copy_stride : 0;
    H.geo0 = moved(H.geo0, off); H.top_inv = moved(H.top_inv, off); H.Cmat = moved(H.Cmat, off); H.Cmat32 = moved(H.Cmat32, off);
    H.Vg = const_cast<double*>(moved(const_cast<const double*>(H.Vg), off));
#pragma unroll
    for (int l = 0; l <= kMlMaxLevels; l++) { H.geo[l] = moved(H.geo[l], off); H.Winv[l] = moved(H.Winv[l], off); }
    return H;                                   // (Sg lives in the PCG vectors' buffer: the same for both copies)
}
__device__ __forceinline__ double* rg_of(const LmSlot& S, int ix, int k) { return reinterpret_cast<double*>(reinterpret_cast<char*>(S.rg[0][k]) + (ix ? S.copy_stride : 0)); }

template <int AGG, class SLOT>
__global__ __launch_bounds__(kCgBlk) void ml_init_lm_kernel(const SLOT slots)
{
    const LmSlot& S = slot_of(slots);
    const LmDev* lm = S.lm;
    if (lm->phase != kLmSolve || lm->init_pass != lm->pass || (int)blockIdx.x >= S.g_rows) return;
    ml_init_kernel_body<AGG>(S.Dp, hot_of(S, lm->ix), S.pbuf[0], S.pbuf[1], rg_of(S, lm->ix, 0));
}
// PCG iteration i of a replay (parity = i & 1): p_old = pbuf[parity], p_new = pbuf[parity ^ 1]; a no-op once flags[0] is set
template <int AGG, int RPW, int WAVES, class SLOT>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 4 : 3))) void ml_spmv_lm_kernel(const SLOT slots, int parity)
{
    const LmSlot& S = slot_of(slots);
    const LmDev* lm = S.lm;
    ml_spmv_kernel_body<AGG, RPW, WAVES>(S.Dp, hot_of(S, lm->ix), S.pbuf[parity], S.pbuf[parity ^ 1], S.g_rows, lm->tol2);
}
// init = 1: the first application of the preconditioner (r = b), in the pass that starts the solve
template <int kCompU, bool kLds, class SLOT>
__global__ __launch_bounds__(kCgBlk) void ml_cg_comp_lm_kernel(const SLOT slots, int parity, int init)
{
    const LmSlot& S = slot_of(slots);
    if ((int)blockIdx.x >= S.g_rows) return;                 // (a batch launches the largest graph's grid)
    const LmDev* lm = S.lm;
    const int ix = lm->ix;
    if (init) {
        if (lm->phase != kLmSolve || lm->init_pass != lm->pass) return;
        ml_cg_comp_kernel_body<kCompU, kLds>(S.Dp, hot_of(S, ix), S.pbuf[0], rg_of(S, ix, 0), rg_of(S, ix, 1), 0, 1);
    } else ml_cg_comp_kernel_body<kCompU, kLds>(S.Dp, hot_of(S, ix), S.pbuf[parity ^ 1], rg_of(S, ix, parity ^ 1), rg_of(S, ix, parity), S.g_spmv, 0);
}
template <int AGG, bool COMP, bool YPRE, bool VPRE, class SLOT>
__global__ __launch_bounds__(kCgBlk) __attribute__((amdgpu_waves_per_eu(COMP ? (YPRE ? 2 : 3) : 1))) void ml_cg_lm_kernel(const SLOT slots, int parity, int init)
{
    const LmSlot& S = slot_of(slots);
    if ((int)blockIdx.x >= S.g_rows) return;
    const LmDev* lm = S.lm;
    const int ix = lm->ix;
    if (init) {
        if (lm->phase != kLmSolve || lm->init_pass != lm->pass) return;
        ml_cg_kernel_body<AGG, COMP, YPRE, false>(S.Dp, hot_of(S, ix), S.pbuf[0], rg_of(S, ix, 0), rg_of(S, ix, 1), 0, 1);
    } else ml_cg_kernel_body<AGG, COMP, YPRE, VPRE>(S.Dp, hot_of(S, ix), S.pbuf[parity ^ 1], rg_of(S, ix, parity ^ 1), rg_of(S, ix, parity), S.g_spmv, 0);
}
template <class SLOT>
__global__ __launch_bounds__(256) void ml_alpha_lm_kernel(const SLOT slots, int parity)
{
    const LmSlot& S = slot_of(slots);
    const LmDev* lm = S.lm;
    ml_alpha_kernel_body(S.Dp, hot_of(S, lm->ix), rg_of(S, lm->ix, parity ^ 1), S.g_spmv);
}

// raises the dynamic-LDS limit of an ml_cg variant once per device (a function attribute is per device)
static hipError_t lm_cg_lds(const void* fn, int variant_ix, size_t lds)
{
    constexpr int kMaxDev = 16, kVar = 16;
    static size_t configured_tab[kMaxDev][kVar] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (lds > configured_tab[dev][variant_ix]) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured_tab[dev][variant_ix] = lds;
    }
    return hipSuccess;
}

// the ml_cg launch of iteration parity `parity` (init = 1: first application) for the shape's variant; SLOT as above
// (ev_a / ev_b, profiling only: the dispatch's own start / stop timestamps - the throughput-geometry variants a batch runs)
template <class SLOT>
static hipError_t kl_ml_cg_t(SLOT sl, const LmShape& sh, int parity, int init, hipStream_t s, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr)
{
    constexpr bool kPtr = std::is_pointer<SLOT>::value;
    const dim3 g(sh.g_rows, 1, sh.nslots), t(kCgBlk);
    size_t lds = (size_t)sh.cg_lds;
    switch (sh.cg_variant) {
    case kCgComp1:
#define UZL_LM_COMP(U, LDS) do { if (ev_a) hipExtLaunchKernelGGL((ml_cg_comp_lm_kernel<U, LDS, SLOT>), g, t, 0, s, ev_a, ev_b, 0, sl, parity, init); \
                                 else hipLaunchKernelGGL((ml_cg_comp_lm_kernel<U, LDS, SLOT>), g, t, 0, s, sl, parity, init); } while (0)
        if (sh.batch_geometry) { if (sh.comp_u <= 5) UZL_LM_COMP(5, true); else UZL_LM_COMP(8, true); }
        else if (sh.comp_u <= 5) UZL_LM_COMP(5, false);
        else if (sh.comp_u <= 8) UZL_LM_COMP(8, false);
        else if (sh.comp_u <= 12) UZL_LM_COMP(12, false);
        else UZL_LM_COMP(16, false);
#undef UZL_LM_COMP
        return hipSuccess;
    case kCgPlain1: {
        const hipError_t e = lm_cg_lds(reinterpret_cast<const void*>(&ml_cg_lm_kernel<1, false, false, false, SLOT>), kPtr ? 0 : 8, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ml_cg_lm_kernel<1, false, false, false, SLOT>), g, t, lds, s, sl, parity, init);
        return hipSuccess; }
    case kCgPlain4: {
        const hipError_t e = lm_cg_lds(reinterpret_cast<const void*>(&ml_cg_lm_kernel<4, false, false, false, SLOT>), kPtr ? 1 : 9, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ml_cg_lm_kernel<4, false, false, false, SLOT>), g, t, lds, s, sl, parity, init);
        return hipSuccess; }
    case kCgComp4: {
        const hipError_t e = lm_cg_lds(reinterpret_cast<const void*>(&ml_cg_lm_kernel<4, true, false, false, SLOT>), kPtr ? 2 : 10, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ml_cg_lm_kernel<4, true, false, false, SLOT>), g, t, lds, s, sl, parity, init);
        return hipSuccess; }
    case kCgComp4Ypre: {
        const hipError_t e = lm_cg_lds(reinterpret_cast<const void*>(&ml_cg_lm_kernel<4, true, true, false, SLOT>), kPtr ? 3 : 11, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ml_cg_lm_kernel<4, true, true, false, SLOT>), g, t, lds, s, sl, parity, init);
        return hipSuccess; }
    case kCgComp4Vpre: {
        const hipError_t e = lm_cg_lds(reinterpret_cast<const void*>(&ml_cg_lm_kernel<4, true, false, true, SLOT>), kPtr ? 4 : 12, lds);
        if (e != hipSuccess) return e;
        // alpha and rg - alpha Sg prepared once by ml_alpha_kernel (not with r = b: the init application takes the plain COMP path)
        if (!init) hipLaunchKernelGGL((ml_alpha_lm_kernel<SLOT>), dim3((6 * sh.n_lv[2] + 255) / 256, 1, sh.nslots), dim3(256), 0, s, sl, parity);
        hipLaunchKernelGGL((ml_cg_lm_kernel<4, true, false, true, SLOT>), g, t, lds, s, sl, parity, init);
        return hipSuccess; }
    }
    return hipErrorInvalidValue;
}
template <class SLOT>
static void kl_ml_spmv_t(SLOT sl, const LmShape& sh, int parity, hipStream_t s, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr)
{
    if (sh.batch_geometry && ev_a) hipExtLaunchKernelGGL((ml_spmv_lm_kernel<1, kSpmvBatchRpw, kSpmvBatchWaves, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(64 * kSpmvBatchWaves), 0, s, ev_a, ev_b, 0, sl, parity);
    else if (sh.batch_geometry) hipLaunchKernelGGL((ml_spmv_lm_kernel<1, kSpmvBatchRpw, kSpmvBatchWaves, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(64 * kSpmvBatchWaves), 0, s, sl, parity);
    else if (sh.agg == 1) {
        // one row per wave (8 waves) or two (4 waves: half the waves to launch, the same 8 rows per workgroup, the same bits);
        // UZL_SPMV1_RPW (diagnostic build) picks
        static const int rpw1 = diag_int("UZL_SPMV1_RPW", 1);
        if (rpw1 == 2) hipLaunchKernelGGL((ml_spmv_lm_kernel<1, 2, 4, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(256), 0, s, sl, parity);
        else hipLaunchKernelGGL((ml_spmv_lm_kernel<1, 1, 8, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(512), 0, s, sl, parity);
    }
    else {
        // rows per wave of the AGG = 4 geometry (16 rows per workgroup; the same bits either way).  Up to two such workgroups per CU the
        // kernel is a latency chain per row and twice the lanes per row shorten it (5k / 25k 23.4 -> 21.8 ms, 6k / 30k 25.5 -> 23.8, 8k / 24k
        // 35.8 -> 34.2); beyond, four rows per wave keep three workgroups on a CU (10k / 50k: 626 workgroups in one round; 47.0 -> 49.2 ms
        // at two rows per wave).  tests/diag/knob_sweep.sh; UZL_SPMV4_RPW (diagnostic build) fixes it
        static const int rpw_env = diag_int("UZL_SPMV4_RPW", 0);
        static const int two_per_cu = [] { int dev = 0, cu = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev); return 2 * cu; }();
        const int rpw = rpw_env ? rpw_env : (sh.g_spmv <= two_per_cu ? 2 : 4);
        if (rpw == 2) hipLaunchKernelGGL((ml_spmv_lm_kernel<4, 2, 8, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(512), 0, s, sl, parity);
        else if (rpw == 1) hipLaunchKernelGGL((ml_spmv_lm_kernel<4, 1, 16, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(1024), 0, s, sl, parity);
        else hipLaunchKernelGGL((ml_spmv_lm_kernel<4, 4, kSpmvWaves4, SLOT>), dim3(sh.g_spmv, 1, sh.nslots), dim3(64 * kSpmvWaves4), 0, s, sl, parity);
    }
}
template <class SLOT>
static hipError_t kl_ml_init_t(SLOT sl, const LmShape& sh, hipStream_t s)
{
    if (sh.agg == 1) hipLaunchKernelGGL((ml_init_lm_kernel<1, SLOT>), dim3(sh.g_rows, 1, sh.nslots), dim3(kCgBlk), 0, s, sl);
    else hipLaunchKernelGGL((ml_init_lm_kernel<4, SLOT>), dim3(sh.g_rows, 1, sh.nslots), dim3(kCgBlk), 0, s, sl);
    return kl_ml_cg_t<SLOT>(sl, sh, 0, 1, s);
}
// which ml_cg kernel serves a hierarchy (the choice k_ml_cg makes per launch), and its dynamic LDS
void ml_cg_variant(const MlHot& ml, int agg, size_t lds_full, int32_t* variant, int32_t* comp_u, uint64_t* lds)
{
    static const bool no_vpre = diag_flag("UZL_NO_VPRE");                            // A/B switch (diagnostic build)
    *comp_u = 0; *lds = lds_full;
    if (agg == 1) {
        if (!ml.Cmat) { *variant = kCgPlain1; return; }
        const int cols = 6 * ml.n[1];
        *variant = kCgComp1; *lds = 0;
        *comp_u = cols <= 5 * kCgBlk ? 5 : (cols <= 8 * kCgBlk ? 8 : (cols <= 12 * kCgBlk ? 12 : 16));
        return;
    }
    if (!ml.Cmat) { *variant = kCgPlain4; return; }
    *lds = ml_comp4_lds(ml.n[2]);                                                    // COMP stages nothing but the gather-level vector
    const bool ypre = ml.levels >= 2 && 6 * ml.n[2] <= 4 * 32 * kYU;
    *variant = ypre ? kCgComp4Ypre : ((ml.Vg != nullptr && !no_vpre) ? kCgComp4Vpre : kCgComp4);
}
// x = 0, r = b, first application of the preconditioner - for the graphs whose solve starts in this pass.  by_value: the pass has one
// graph and `host_slot` is its slot (the device table `sl` is what every other twin reads)
hipError_t kl_ml_init(const LmSlot* sl, const LmSlot* host_slot, const LmShape& sh, hipStream_t s)
{
    if (host_slot && sh.nslots == 1) return kl_ml_init_t<LmSlot>(*host_slot, sh, s);
    return kl_ml_init_t<const LmSlot*>(sl, sh, s);
}
// PCG iterations first .. first + n - 1 of a solve (iteration i: p_old = pbuf[i & 1], p_new = pbuf[(i & 1) ^ 1]).  ev (profiling only, may
// be null): 4 events per iteration - spmv start / stop, cg start / stop (dispatch timestamps; the batch's kernels)
hipError_t kl_ml_pcg_its(const LmSlot* sl, const LmSlot* host_slot, const LmShape& sh, int first, int n, hipStream_t s, hipEvent_t* ev)
{
    const bool by_value = host_slot && sh.nslots == 1 && !ev;
    for (int q = 0; q < n; q++) {
        const int par = (first + q) & 1;
        hipError_t e;
        if (ev) { kl_ml_spmv_t<const LmSlot*>(sl, sh, par, s, ev[4 * q], ev[4 * q + 1]); e = kl_ml_cg_t<const LmSlot*>(sl, sh, par, 0, s, ev[4 * q + 2], ev[4 * q + 3]); }
        else if (by_value) { kl_ml_spmv_t<LmSlot>(*host_slot, sh, par, s); e = kl_ml_cg_t<LmSlot>(*host_slot, sh, par, 0, s); }
        else { kl_ml_spmv_t<const LmSlot*>(sl, sh, par, s); e = kl_ml_cg_t<const LmSlot*>(sl, sh, par, 0, s); }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace uzl

#ifdef UZL_STAMPS
extern "C" UZL_DIAG_EXPORT int uzl_debug_read_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(uzl::g_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return -3;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(uzl::g_stamps), z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
#endif

// test hooks of the diagnostic build (not part of include/uzl_mi355x.h): out = 2 X - X T for host matrices, through ml_ns_gemm_kernel
#ifdef UZL_DIAG
extern "C" UZL_DIAG_EXPORT int uzl_debug_ns_gemm32(int n, const double* X, const double* T, double* out)
{
    if (n <= 0 || n > uzl::kGemm32Max || !X || !T || !out) return -1;
    double *dX = nullptr, *dT = nullptr, *dO = nullptr;
    const size_t b = (size_t)n * n * 8;
    if (hipMalloc((void**)&dX, b) != hipSuccess || hipMalloc((void**)&dT, b) != hipSuccess || hipMalloc((void**)&dO, b) != hipSuccess) return -3;
    (void)hipMemcpy(dX, X, b, hipMemcpyHostToDevice);
    (void)hipMemcpy(dT, T, b, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(uzl::ml_ns_gemm32_kernel, dim3(uzl::gemm32_grid((n + 31) / 32)), dim3(256), 0, nullptr, n, dX, dT, dO, (float*)nullptr, 0);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(out, dO, b, hipMemcpyDeviceToHost);
    (void)hipFree(dX); (void)hipFree(dT); (void)hipFree(dO);
    return e == hipSuccess ? 0 : -3;
}
extern "C" UZL_DIAG_EXPORT int uzl_debug_ns_gemm(int n, const double* X, const double* T, double* out)
{
    if (n <= 0 || !X || !T || !out) return -1;
    double *dX = nullptr, *dT = nullptr, *dO = nullptr;
    const size_t b = (size_t)n * n * 8;
    if (hipMalloc((void**)&dX, b) != hipSuccess || hipMalloc((void**)&dT, b) != hipSuccess || hipMalloc((void**)&dO, b) != hipSuccess) return -3;
    (void)hipMemcpy(dX, X, b, hipMemcpyHostToDevice);
    (void)hipMemcpy(dT, T, b, hipMemcpyHostToDevice);
    const int g = (n + uzl::kGemmTile - 1) / uzl::kGemmTile;
    hipLaunchKernelGGL(uzl::ml_ns_gemm_kernel, dim3(g * (g + 1) / 2), dim3(256), 0, nullptr, n, dX, dT, dO, (float*)nullptr, 0);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(out, dO, b, hipMemcpyDeviceToHost);
    (void)hipFree(dX); (void)hipFree(dT); (void)hipFree(dO);
    return e == hipSuccess ? 0 : -3;
}
// admission of the dense level-2 operator's PCG variant (host arithmetic only: runs without a device - tests/test_ml_admission.py):
// the LDS ml_cg asks for with n2 level-2 aggregates, ml_spmv's workgroups (= partials) for nb free vertices, and build_ml's verdict
extern "C" UZL_DIAG_EXPORT int uzl_debug_ml_admission(int nb, int n2, uint64_t* lds, int* spmv_groups, int* fits)
{
    if (nb <= 0 || n2 <= 0) return -1;
    if (lds) *lds = uzl::ml_comp4_lds(n2);
    if (spmv_groups) *spmv_groups = uzl::g_ml_spmv(nb, 4);
    if (fits) *fits = uzl::ml_comp4_fits(nb, n2) ? 1 : 0;
    return 0;
}
#endif
