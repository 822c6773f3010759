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
// Per PCG iteration : pcg_spmv -> ml_update (x, r, D0^-1 r, r1 = P1^T r, r2 = P2^T r1)
//                     -> ml_finish (levels >= 2 in LDS by every workgroup, z += P1 y1, r.z partials)
#include "pgo_device.hpp"

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

// number of level-0 blocks under entity i of level l
__device__ __forceinline__ int leaves_under(int l, int i, int nb)
{
    int span = 1;
    for (int k = 0; k < l; k++) span *= kMlFanout;
    const int lo = i * span;
    const int hi = lo + span < nb ? lo + span : nb;
    return hi > lo ? hi - lo : 0;
}

// ---- geometry of level l (>= 1): centroid of every aggregate, then the children's offsets d
__global__ __launch_bounds__(kBlk) void ml_geometry_kernel(PgoDev D, const MlDev* __restrict__ mlp,
                                                          const double* __restrict__ pose, int l)
{
    const MlDev& ml = *mlp;
    const int A = blockIdx.x * kBlk + threadIdx.x;
    const int n = ml.lv[l].n, nc = ml.lv[l - 1].n;
    if (A >= n) return;
    const int c0 = A * kMlFanout, c1 = (c0 + kMlFanout < nc) ? c0 + kMlFanout : nc;
    double cx = 0, cy = 0, cz = 0, wsum = 0;
    for (int c = c0; c < c1; c++) {
        double px, py, pz, w;
        if (l == 1) {
            const Pose P = load_pose(pose, D.b2v[c]);
            px = P.t.x; py = P.t.y; pz = P.t.z; w = 1.;
        } else {
            const double* cc = ml.lv[l - 1].cen + (size_t)c * 3;
            px = cc[0]; py = cc[1]; pz = cc[2]; w = (double)leaves_under(l - 1, c, D.nb);
        }
        cx += w * px; cy += w * py; cz += w * pz; wsum += w;
    }
    cx /= wsum; cy /= wsum; cz /= wsum;
    double* cen = ml.lv[l].cen + (size_t)A * 3;
    cen[0] = cx; cen[1] = cy; cen[2] = cz;
    for (int c = c0; c < c1; c++) {
        if (l == 1) {
            const Pose P = load_pose(pose, D.b2v[c]);
            const M33 R = qrot(P.q);
            double* g = ml.lv[0].geo + (size_t)c * 12;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) g[r * 3 + k] = R.m[k * 3 + r];       // R^T
            g[9] = P.t.x - cx; g[10] = P.t.y - cy; g[11] = P.t.z - cz;
        } else {
            const double* cc = ml.lv[l - 1].cen + (size_t)c * 3;
            double* g = ml.lv[l - 1].geo + (size_t)c * 3;
            g[0] = cc[0] - cx; g[1] = cc[1] - cy; g[2] = cc[2] - cz;
        }
    }
}

// ---- Galerkin transform of level f: every off-diagonal block and every diagonal block of A_f (and of M_f)
//      is mapped through its two prolongation blocks and dropped into its sorted contribution position
__global__ __launch_bounds__(kBlk) void ml_transform_kernel(PgoDev D, const MlDev* __restrict__ mlp, int f)
{
    const MlDev& ml = *mlp;
    const MlLevel& L = ml.lv[f];
    const int t = blockIdx.x * kBlk + threadIdx.x;
    const int ns = L.nslots, n = L.n;
    const double* geo = L.geo;
    if (t < ns) {
        const int c = (f == 0) ? D.col[t] : L.col[t];
        const int pos = L.tpos[t];
        if (c >= 0 && pos >= 0) {
            const int a = L.srow[t];
            P3 PL, PR;
            make_P(f, geo, a, PL);
            make_P(f, geo, c, PR);
            const double* F = ((f == 0) ? D.blk : L.blk) + (size_t)t * 36;
            galerkin(PL, F, PR, ml.tmp + (size_t)pos * 36);
        }
    } else if (t < ns + n) {
        const int i = t - ns;
        P3 P;
        make_P(f, geo, i, P);
        const double* G = ((f == 0) ? D.hdiag : L.G) + (size_t)i * 36;
        galerkin(P, G, P, ml.tmpG + (size_t)i * 36);
        if (f == 0) {
            const double I6[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
            galerkin(P, I6, P, ml.tmpM + (size_t)i * 36);
        } else {
            galerkin(P, L.M + (size_t)i * 36, P, ml.tmpM + (size_t)i * 36);
        }
    }
}

// ---- ordered reduction of the contributions into A_l (off-diagonal blocks), G_l and M_l
__global__ __launch_bounds__(kBlk) void ml_reduce_kernel(const MlDev* __restrict__ mlp, int l)
{
    const MlDev& ml = *mlp;
    const MlLevel& L = ml.lv[l];
    const int nc = ml.lv[l - 1].n;
    const int t = blockIdx.x * kBlk + threadIdx.x;
    const int blk_id = t / 36, k = t % 36;
    if (blk_id < L.nslots) {
        double s = 0.;
        for (int q = L.off_ptr[blk_id]; q < L.off_ptr[blk_id + 1]; q++) s += ml.tmp[(size_t)q * 36 + k];
        L.blk[(size_t)blk_id * 36 + k] = s;
    } else if (blk_id < L.nslots + L.n) {
        const int A = blk_id - L.nslots;
        double s = 0., m = 0.;
        for (int q = L.diag_ptr[A]; q < L.diag_ptr[A + 1]; q++) s += ml.tmp[(size_t)(L.n_off_contrib + q) * 36 + k];
        const int c0 = A * kMlFanout, c1 = (c0 + kMlFanout < nc) ? c0 + kMlFanout : nc;
        for (int c = c0; c < c1; c++) { s += ml.tmpG[(size_t)c * 36 + k]; m += ml.tmpM[(size_t)c * 36 + k]; }
        L.G[(size_t)A * 36 + k] = s;
        L.M[(size_t)A * 36 + k] = m;
    }
}

// ---- per LM trial: D_l(lambda)^-1 for the intermediate levels 1..L-1 (one lane per aggregate)
__global__ __launch_bounds__(kBlk) void ml_invert_kernel(PgoDev D, const MlDev* __restrict__ mlp)
{
    const MlDev& ml = *mlp;
    const double lambda = D.scal[3];
    int t = blockIdx.x * kBlk + threadIdx.x;
    for (int l = 1; l < ml.levels; l++) {
        const MlLevel& L = ml.lv[l];
        if (t < L.n) {
            double A[36], out[36];
#pragma unroll
            for (int k = 0; k < 36; k++) A[k] = L.G[(size_t)t * 36 + k] + lambda * L.M[(size_t)t * 36 + k];
            spd_inverse6(A, out);
#pragma unroll
            for (int k = 0; k < 36; k++) L.Dinv[(size_t)t * 36 + k] = out[k];
            return;
        }
        t -= L.n;
    }
}

// ---- per LM trial: dense inverse of the top level A_L(lambda) (<= 48 x 48), one workgroup, in LDS
__global__ __launch_bounds__(kBlk) void ml_top_kernel(PgoDev D, const MlDev* __restrict__ mlp)
{
    __shared__ double sA[48 * 97];      // [A | I], row stride 97 (odd: no bank pile-up on column walks)
    const MlDev& ml = *mlp;
    const MlLevel& L = ml.lv[ml.levels];
    const double lambda = D.scal[3];
    const int n = 6 * L.n, W = 2 * n, ld = 97;
    for (int i = threadIdx.x; i < n * W; i += kBlk) {
        const int r = i / W, c = i % W;
        sA[r * ld + c] = (c >= n) ? ((c - n == r) ? 1. : 0.) : 0.;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L.n * 36; i += kBlk) {
        const int A = i / 36, k = i % 36;
        sA[(6 * A + k / 6) * ld + 6 * A + k % 6] = L.G[i] + lambda * L.M[i];
    }
    for (int i = threadIdx.x; i < L.nslots * 36; i += kBlk) {
        const int s = i / 36, k = i % 36;
        sA[(6 * L.srow[s] + k / 6) * ld + 6 * L.col[s] + k % 6] = L.blk[i];
    }
    __syncthreads();
    // Gauss-Jordan without pivoting (SPD)
    for (int p = 0; p < n; p++) {
        const double piv = 1. / sA[p * ld + p];
        __syncthreads();
        for (int c = threadIdx.x; c < W; c += kBlk) sA[p * ld + c] *= piv;
        __syncthreads();
        for (int i = threadIdx.x; i < n * W; i += kBlk) {
            const int r = i / W, c = i % W;
            if (r != p && c != p) sA[r * ld + c] -= sA[r * ld + p] * sA[p * ld + c];
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += kBlk) if (r != p) sA[r * ld + p] = 0.;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n * n; i += kBlk) ml.top_inv[i] = sA[(i / n) * ld + n + i % n];
}

// ------------------------------------------------------------------------------------------------
// PCG-iteration kernels
// ------------------------------------------------------------------------------------------------
constexpr int kMlBlk = 384;              // 6 waves
constexpr int kRowsPerBlk = 64;          // 8 level-1 aggregates = 1 level-2 aggregate per workgroup
constexpr int kAggPerBlk = kRowsPerBlk / kMlFanout;
constexpr int kCoarseLdsDoubles = 7168;  // residuals of levels >= 2 staged in LDS by every workgroup (56 KB)

__device__ __forceinline__ double block_sum6(double v, double* s6)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s6[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s6[0] + s6[1]) + (s6[2] + s6[3])) + (s6[4] + s6[5]);
}
__device__ __forceinline__ double sum_partials6(const double* __restrict__ part, int count, double* s6)
{
    double v = 0.;
    for (int i = threadIdx.x; i < count; i += kMlBlk) v += part[i];
    return block_sum6(v, s6);
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

// init = 1: x = 0, r = b, p0 = p1 = 0, flags cleared.  init = 0: alpha = rz / p.Ap, x += alpha p, r -= alpha Ap.
// Then z = D0^-1 r (block-Jacobi part), r1 = P1^T r for the 8 level-1 aggregates of this workgroup and
// r2 = P2^T r1 for its level-2 aggregate.
__global__ __launch_bounds__(kMlBlk) void ml_update_kernel(PgoDev D, const MlDev* __restrict__ mlp,
                                                          const double* __restrict__ p, double* __restrict__ p0,
                                                          double* __restrict__ p1, int n_part, int init)
{
    __shared__ double s6[6];
    __shared__ double sv[kMlBlk];
    __shared__ double sw[kMlBlk];
    __shared__ double sr1[kAggPerBlk * 6];
    const MlDev& ml = *mlp;
    if (!init && D.flags[0]) return;
    double alpha = 0., rz = 0.;
    bool bad = false;
    if (!init) {
        const double pAp = sum_partials6(D.part_a, n_part, s6);
        rz = D.scal[0];
        bad = !(pAp > 0.);
        alpha = bad ? 0. : rz / pAp;
    }
    const int tid = threadIdx.x;
    const int a = blockIdx.x * kRowsPerBlk + tid / 6, r = tid % 6;
    const bool act = a < D.nb;
    double rv = 0.;
    if (act) {
        const size_t i = (size_t)a * 6 + r;
        if (init) { rv = D.b[i]; D.x[i] = 0.; p0[i] = 0.; p1[i] = 0.; }
        else { D.x[i] += alpha * p[i]; rv = D.r[i] - alpha * D.ap[i]; }
        D.r[i] = rv;
    }
    sv[tid] = rv;
    __syncthreads();
    double w = 0.;
    if (act) {
        const int g0 = tid - r;
        const double* __restrict__ m = D.minv + (size_t)a * 36 + r * 6;
        double zz = 0.;
#pragma unroll
        for (int c = 0; c < 6; c++) zz += m[c] * sv[g0 + c];
        D.z[(size_t)a * 6 + r] = zz;
        // (P1_a^T r_a)[r]:  u = R r_t ; r < 3: u[r] ; r >= 3: (d x u)[r-3] + 1/2 (R r_q)[r-3]
        const double* __restrict__ g = ml.lv[0].geo + (size_t)a * 12;
        const double t0 = sv[g0], t1 = sv[g0 + 1], t2 = sv[g0 + 2];
        const double u0 = g[0] * t0 + g[3] * t1 + g[6] * t2;      // R = (R^T)^T
        const double u1 = g[1] * t0 + g[4] * t1 + g[7] * t2;
        const double u2 = g[2] * t0 + g[5] * t1 + g[8] * t2;
        if (r < 3) w = (r == 0) ? u0 : (r == 1) ? u1 : u2;
        else {
            const double q0 = sv[g0 + 3], q1 = sv[g0 + 4], q2 = sv[g0 + 5];
            const int k = r - 3;
            const double rq = 0.5 * (g[k] * q0 + g[3 + k] * q1 + g[6 + k] * q2);
            const double dx = g[9], dy = g[10], dz = g[11];
            const double cr = (k == 0) ? dy * u2 - dz * u1 : (k == 1) ? dz * u0 - dx * u2 : dx * u1 - dy * u0;
            w = cr + rq;
        }
    }
    sw[tid] = w;
    __syncthreads();
    if (tid < kAggPerBlk * 6) {
        const int la = tid / 6, k = tid % 6;                  // local aggregate, component
        const int A = blockIdx.x * kAggPerBlk + la;
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[(la * kMlFanout + j) * 6 + k];      // inactive rows hold 0
        sr1[tid] = s;
        if (A < ml.lv[1].n) ml.lv[1].r[(size_t)A * 6 + k] = s;
    }
    __syncthreads();
    if (ml.levels >= 2 && tid < 6) {
        const int A2 = blockIdx.x;
        const int n1 = ml.lv[1].n;
        double s = 0.;
        for (int j = 0; j < kAggPerBlk; j++) {
            const int c = A2 * kMlFanout + j;
            if (c < n1) s += restrict_comp(ml.lv[1].geo + (size_t)c * 3, sr1 + j * 6, tid);
        }
        ml.lv[2].r[(size_t)A2 * 6 + tid] = s;
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (init) { D.flags[0] = 0; D.flags[1] = 0; D.flags[2] = 0; D.scal[2] = 1.; }
        else {
            D.scal[2] = rz;
            D.flags[1] += 1;
            if (bad) { D.flags[0] = 1; D.flags[2] = 1; }
        }
    }
}

// Coarse correction + z + r.z in one launch.  Every workgroup stages the residuals of the gather level
// g = min(2, L) in LDS, restricts them up to the top level, applies the top inverse for its own ancestor,
// walks back down its own ancestor chain and finishes z for its 64 rows:
//   z += P1 y1,  y1 = D1^-1 r1 + P2 y2,  y2 = D2^-1 r2 + P3 y3, ...   (partials of r.z -> part_b)
__global__ __launch_bounds__(kMlBlk) void ml_finish_kernel(PgoDev D, const MlDev* __restrict__ mlp)
{
    __shared__ double s6[6];
    __shared__ double sy[kAggPerBlk * 6];
    __shared__ double syc[6];                       // correction of the own level-g ancestor
    __shared__ double sres[kCoarseLdsDoubles];
    const MlDev& ml = *mlp;
    if (D.flags[0]) return;
    const int tid = threadIdx.x;
    const int Lt = ml.levels;
    const int g = Lt >= 2 ? 2 : 1;
    // ---- stage r_g (all of it) and restrict up to the top level, all in LDS
    int off[kMlMaxLevels + 2];
    off[g] = 0;
    for (int l = g; l <= Lt; l++) off[l + 1] = off[l] + 6 * ml.lv[l].n;
    for (int t = tid; t < 6 * ml.lv[g].n; t += kMlBlk) sres[t] = ml.lv[g].r[t];
    __syncthreads();
    for (int l = g + 1; l <= Lt; l++) {
        const MlLevel& C = ml.lv[l - 1];
        const int nP = ml.lv[l].n;
        for (int t = tid; t < nP * 6; t += kMlBlk) {
            const int A = t / 6, k = t % 6;
            const int c0 = A * kMlFanout, c1 = (c0 + kMlFanout < C.n) ? c0 + kMlFanout : C.n;
            double s = 0.;
            for (int c = c0; c < c1; c++) s += restrict_comp(C.geo + (size_t)c * 3, sres + off[l - 1] + c * 6, k);
            sres[off[l] + t] = s;
        }
        __syncthreads();
    }
    const int ntop = 6 * ml.lv[Lt].n;
    const double* rtop = sres + off[Lt];
    if (Lt == 1) {
        // the top level is level 1 itself: y1 of the own aggregates straight from the dense inverse
        if (tid < kAggPerBlk * 6) {
            const int A = blockIdx.x * kAggPerBlk + tid / 6, k = tid % 6;
            double s = 0.;
            if (A < ml.lv[1].n) for (int c = 0; c < ntop; c++) s += ml.top_inv[(size_t)(6 * A + k) * ntop + c] * rtop[c];
            sy[tid] = s;
        }
        __syncthreads();
    } else {
        // own ancestor chain: level 2 aggregate = blockIdx.x
        int anc[kMlMaxLevels + 1];
        anc[2] = blockIdx.x;
        for (int l = 3; l <= Lt; l++) anc[l] = anc[l - 1] / kMlFanout;
        if (tid < 6) {
            double s = 0.;
            for (int c = 0; c < ntop; c++) s += ml.top_inv[(size_t)(6 * anc[Lt] + tid) * ntop + c] * rtop[c];
            syc[tid] = s;
        }
        __syncthreads();
        for (int l = Lt - 1; l >= 2; l--) {
            double s = 0.;
            if (tid < 6) {
                const MlLevel& C = ml.lv[l];
                const int a = anc[l];
                const double* di = C.Dinv + (size_t)a * 36 + tid * 6;
                const double* rr = sres + off[l] + a * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) s += di[c] * rr[c];
                s += prolong_comp(C.geo + (size_t)a * 3, syc, tid);
            }
            __syncthreads();
            if (tid < 6) syc[tid] = s;
            __syncthreads();
        }
        // y1 of the 8 own level-1 aggregates
        if (tid < kAggPerBlk * 6) {
            const MlLevel& L1 = ml.lv[1];
            const int A = blockIdx.x * kAggPerBlk + tid / 6, k = tid % 6;
            double s = 0.;
            if (A < L1.n) {
                const double* di = L1.Dinv + (size_t)A * 36 + k * 6;
                const double* rr = L1.r + (size_t)A * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) s += di[c] * rr[c];
                s += prolong_comp(L1.geo + (size_t)A * 3, syc, k);
            }
            sy[tid] = s;
        }
        __syncthreads();
    }
    const int a = blockIdx.x * kRowsPerBlk + tid / 6, r = tid % 6;
    double acc = 0.;
    if (a < D.nb) {
        const double* y = sy + ((tid / 6) / kMlFanout) * 6;
        const double* __restrict__ gg = ml.lv[0].geo + (size_t)a * 12;
        double add;
        if (r < 3) {
            const double dx = gg[9], dy = gg[10], dz = gg[11];
            const double vx = y[0] + (y[4] * dz - y[5] * dy);      // v + w x d
            const double vy = y[1] + (y[5] * dx - y[3] * dz);
            const double vz = y[2] + (y[3] * dy - y[4] * dx);
            add = gg[r * 3] * vx + gg[r * 3 + 1] * vy + gg[r * 3 + 2] * vz;       // R^T (.)
        } else {
            const int k = r - 3;
            add = 0.5 * (gg[k * 3] * y[3] + gg[k * 3 + 1] * y[4] + gg[k * 3 + 2] * y[5]);
        }
        const size_t i = (size_t)a * 6 + r;
        const double zz = D.z[i] + add;
        D.z[i] = zz;
        acc = D.r[i] * zz;
    }
    const double tot = block_sum6(acc, s6);
    if (tid == 0) D.part_b[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------
// Two launches per PCG iteration with the full multilevel preconditioner.
//
//   ml_spmv : one workgroup = one level-1 aggregate (8 rows, one wave per row).  beta from the r.z partials,
//             p = z + beta p_old (own row + recomputed for neighbour columns), Ap, p.Ap partial, and
//             S1[A] = sum_rows P1^T (Ap)  — the level-1 restriction of Ap.
//   ml_cg   : one workgroup = one level-2 aggregate (64 rows).  alpha from the p.Ap partials; because
//             r_new = r - alpha Ap, the restricted residual of EVERY aggregate is r1_old - alpha S1 — no global
//             pass over r is needed: every workgroup restricts that to level 2.., applies the top inverse and
//             walks down its own ancestor chain in LDS; then x, r, z for its own 64 rows and the exact r1 of its
//             own aggregates (so the recursion never accumulates error).  r.z partial -> part_b.
// All global operands whose address does not depend on alpha/beta are loaded before the partial reduction so
// that one memory latency covers them (the kernels are latency-, not bandwidth-bound at these sizes).
// ------------------------------------------------------------------------------------------------
constexpr int kSpmvBlk = 512;            // 8 waves = 8 rows = one level-1 aggregate

__global__ __launch_bounds__(kSpmvBlk) void ml_spmv_kernel(PgoDev D, MlHot H,
                                                          const double* __restrict__ p_old, double* __restrict__ p_new,
                                                          int n_part, double tol2)
{
    __shared__ double s8[8];
    __shared__ double sw[8 * 6];
    if (D.flags[0]) return;
    STAMP_DECL
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane / 6, r = lane % 6;
    const bool lact = lane < 60;
    const int a = blockIdx.x * kMlFanout + wv;
    const bool ract = a < D.nb;
    // ---- prefetch (independent of beta); the r.z partials first: they gate everything else
    double v = 0.;
    for (int i = threadIdx.x; i < n_part; i += kSpmvBlk) v += D.part_b[i];
    int s0 = 0, s1 = 0;
    if (ract) { s0 = D.row_ptr[a]; s1 = D.row_ptr[a + 1]; }
    double hrow[6] = {0, 0, 0, 0, 0, 0}, zo[6] = {0, 0, 0, 0, 0, 0}, po[6] = {0, 0, 0, 0, 0, 0}, geo[12];
    if (ract && g == 0) {
        const double* __restrict__ h = D.hdiag + (size_t)a * 36 + r * 6;
        const double* __restrict__ zv = D.z + (size_t)a * 6;
        const double* __restrict__ pv = p_old + (size_t)a * 6;
        const double* __restrict__ gg = H.geo0 + (size_t)a * 12;
#pragma unroll
        for (int c = 0; c < 6; c++) { hrow[c] = h[c]; zo[c] = zv[c]; po[c] = pv[c]; }
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
    // first two slots of this lane group (rows have ~10 slots: most rows need one pass)
    double2 b0[2], b1[2], b2[2], z0[2], z1[2], z2[2], o0[2], o1[2], o2[2];
    bool have[2] = {false, false};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = s0 + g + 10 * k;
        if (ract && lact && s < s1) {
            const int c = D.col[s];
            if (c >= 0) {
                have[k] = true;
                const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                const double2* __restrict__ zv = reinterpret_cast<const double2*>(D.z + (size_t)c * 6);
                const double2* __restrict__ pv = reinterpret_cast<const double2*>(p_old + (size_t)c * 6);
                b0[k] = bk[0]; b1[k] = bk[1]; b2[k] = bk[2];
                z0[k] = zv[0]; z1[k] = zv[1]; z2[k] = zv[2];
                o0[k] = pv[0]; o1[k] = pv[1]; o2[k] = pv[2];
            }
        }
    }
    const int it = D.flags[1];
    const double rz_prev = D.scal[2], thr_old = D.scal[1], lambda = D.scal[3];
    STAMP(16);     // 16: prefetch issue
    // ---- beta
    v = wave_sum(v);
    if (lane == 0) s8[wv] = v;
    __syncthreads();
    const double rz = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    const double beta = (it == 0) ? 0. : rz / rz_prev;
    const double thresh = (it == 0) ? tol2 * rz : thr_old;
    STAMP(16);     // 17: partial reduction (prefetch landed)
    // ---- row product
    double acc = 0., pr = 0.;
    if (ract && g == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double pc = zo[c] + beta * po[c];
            acc += hrow[c] * pc;
            if (c == r) pr = pc;
        }
        acc += lambda * pr;
        p_new[(size_t)a * 6 + r] = pr;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (have[k])
            acc += b0[k].x * (z0[k].x + beta * o0[k].x) + b0[k].y * (z0[k].y + beta * o0[k].y) + b1[k].x * (z1[k].x + beta * o1[k].x) +
                   b1[k].y * (z1[k].y + beta * o1[k].y) + b2[k].x * (z2[k].x + beta * o2[k].x) + b2[k].y * (z2[k].y + beta * o2[k].y);
    }
    if (ract && lact) {
        for (int s = s0 + g + 20; s < s1; s += 10) {
            const int c = D.col[s];
            if (c >= 0) {
                const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                const double2* __restrict__ zv = reinterpret_cast<const double2*>(D.z + (size_t)c * 6);
                const double2* __restrict__ pv = reinterpret_cast<const double2*>(p_old + (size_t)c * 6);
                const double2 c0 = bk[0], c1 = bk[1], c2 = bk[2], y0 = zv[0], y1 = zv[1], y2 = zv[2], q0 = pv[0], q1 = pv[1], q2 = pv[2];
                acc += c0.x * (y0.x + beta * q0.x) + c0.y * (y0.y + beta * q0.y) + c1.x * (y1.x + beta * q1.x) +
                       c1.y * (y1.y + beta * q1.y) + c2.x * (y2.x + beta * q2.x) + c2.y * (y2.y + beta * q2.y);
            }
        }
    }
    double t;
    t = __shfl_down(acc, 48); if (lane + 48 < 60) acc += t;
    t = __shfl_down(acc, 24); if (lane + 24 < 48) acc += t;
    t = __shfl_down(acc, 12); if (lane + 12 < 24) acc += t;
    t = __shfl_down(acc, 6);  if (lane + 6 < 12) acc += t;
    STAMP(16);     // 18: row product + fold
    // lanes 0..5 hold Ap[a][0..5]
    double dot = 0., w = 0.;
    if (ract && lane < 6) {
        D.ap[(size_t)a * 6 + r] = acc;
        dot = acc * pr;
    }
    // (P1_a^T Ap_a)[r] needs all six components of the row: broadcast within lanes 0..5
    const double t0 = __shfl(acc, 0), t1 = __shfl(acc, 1), t2 = __shfl(acc, 2);
    const double q0 = __shfl(acc, 3), q1 = __shfl(acc, 4), q2 = __shfl(acc, 5);
    if (ract && lane < 6) {
        const double u0 = geo[0] * t0 + geo[3] * t1 + geo[6] * t2;
        const double u1 = geo[1] * t0 + geo[4] * t1 + geo[7] * t2;
        const double u2 = geo[2] * t0 + geo[5] * t1 + geo[8] * t2;
        if (r < 3) w = (r == 0) ? u0 : (r == 1) ? u1 : u2;
        else {
            const int k = r - 3;
            const double rq = 0.5 * (geo[k] * q0 + geo[3 + k] * q1 + geo[6 + k] * q2);
            const double dx = geo[9], dy = geo[10], dz = geo[11];
            const double cr = (k == 0) ? dy * u2 - dz * u1 : (k == 1) ? dz * u0 - dx * u2 : dx * u1 - dy * u0;
            w = cr + rq;
        }
    }
    dot = wave_sum(dot);                       // lanes >= 6 hold 0
    __syncthreads();                           // s8 reuse
    if (lane < 6) sw[wv * 6 + lane] = w;
    if (lane == 0) s8[wv] = dot;
    __syncthreads();
    if (tid < 6) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < 8; k++) s += sw[k * 6 + tid];
        H.S1[(size_t)blockIdx.x * 6 + tid] = s;
    }
    if (tid == 0) {
        D.part_a[blockIdx.x] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        if (blockIdx.x == 0) {
            D.scal[0] = rz;
            if (it == 0) D.scal[1] = thresh;
            if (!(rz > thresh)) D.flags[0] = 1;
        }
    }
    STAMP(16);     // 19: S1 + partial stores
#ifdef UZL_STAMPS
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_stamps[47], 1ull);
#endif
}

// init = 1: first application (r = b already stored, exact r1 in r1_old): only the preconditioner part runs.
// r1_old / r1_new: the level-1 residual is double-buffered like p (other workgroups read r1_old while the owner
// writes the exact r1_new of its aggregates).
// Dynamic LDS (doubles): res[levels 2..L] | geo[levels 2..L-1] | top rows | own-chain Dinv+geo |
//                        level-1 chunk: r (6 x kL1Chunk) + geo (3 x kL1Chunk)          (ml_cg_lds_bytes)
// Level 1 is streamed through the chunk buffer (bounded LDS for any graph size); its first chunk is loaded into
// registers before the alpha reduction, so one memory latency covers every operand of the kernel.
constexpr int kStageU = 4;               // loads in flight per thread per staging batch
// batched global -> LDS copy: kStageU independent loads per thread are issued before the first store
__device__ __forceinline__ void stage_to_lds(const double* __restrict__ src, double* dst, int n)
{
    for (int base = 0; base < n; base += kStageU * kMlBlk) {
        double v[kStageU];
#pragma unroll
        for (int u = 0; u < kStageU; u++) { const int t = base + u * kMlBlk + (int)threadIdx.x; v[u] = (t < n) ? src[t] : 0.; }
#pragma unroll
        for (int u = 0; u < kStageU; u++) { const int t = base + u * kMlBlk + (int)threadIdx.x; if (t < n) dst[t] = v[u]; }
    }
}

constexpr int kL1Chunk = 1280;                       // level-1 aggregates per chunk (multiple of 8): 10240 vertices
constexpr int kL1RU = (6 * kL1Chunk + kMlBlk - 1) / kMlBlk;      // 20 residual values per thread
constexpr int kL1GU = (3 * kL1Chunk + kMlBlk - 1) / kMlBlk;      // 10 offsets per thread

__global__ __launch_bounds__(kMlBlk) void ml_cg_kernel(PgoDev D, MlHot H, const double* __restrict__ p,
                                                      const double* __restrict__ r1_old, double* __restrict__ r1_new,
                                                      int n_part, int init)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double s6[6];
    __shared__ double sv[kMlBlk];
    __shared__ double sw[kMlBlk];
    __shared__ double sr1[kAggPerBlk * 6];
    __shared__ double sy[kAggPerBlk * 6];
    __shared__ double syc[6];
    if (D.flags[0]) return;
    STAMP_DECL
    const int tid = threadIdx.x;
    const int Lt = H.levels;
    const int a = blockIdx.x * kRowsPerBlk + tid / 6, r = tid % 6;
    const bool act = a < D.nb;
    const int n1 = H.n[1];
    const int ntop = 6 * H.n[Lt];
    // ---- LDS carve-up
    int roff[kMlMaxLevels + 2], goff[kMlMaxLevels + 2], anc[kMlMaxLevels + 2];
    int o = 0;
    for (int l = 2; l <= Lt; l++) { roff[l] = o; o += 6 * H.n[l]; }
    for (int l = 2; l < Lt; l++) { goff[l] = o; o += 3 * H.n[l]; }
    const int top_off = o;
    const int n_top_rows = (Lt == 1) ? kAggPerBlk * 6 : 6;
    o += n_top_rows * ntop;
    const int chain_off = o;                                  // (L-2) x 39
    if (Lt > 2) o += (Lt - 2) * 39;
    const int c1r = o;                                        // level-1 chunk: residual estimate
    const int c1g = c1r + 6 * kL1Chunk;                       //                children offsets
    anc[2] = blockIdx.x;
    for (int l = 3; l <= Lt; l++) anc[l] = anc[l - 1] / kMlFanout;
    STAMP(0);      // 0: entry
    // ---- every global load whose address is known now, before any barrier
    double part = 0.;
    if (!init) for (int i = tid; i < n_part; i += kMlBlk) part += D.part_a[i];
    double xv = 0., rv0 = 0., apv = 0., pv = 0., mrow[6] = {0, 0, 0, 0, 0, 0}, geo[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const size_t i = (size_t)a * 6 + r;
        rv0 = D.r[i];
        if (!init) { xv = D.x[i]; apv = D.ap[i]; pv = p[i]; }
        const double* __restrict__ m = D.minv + (size_t)a * 36 + r * 6;
        const double* __restrict__ gg = H.geo0 + (size_t)a * 12;
#pragma unroll
        for (int c = 0; c < 6; c++) mrow[c] = m[c];
#pragma unroll
        for (int c = 0; c < 12; c++) geo[c] = gg[c];
    }
    double d1row[6] = {0, 0, 0, 0, 0, 0}, g1own[3] = {0, 0, 0};
    const int A1 = blockIdx.x * kAggPerBlk + tid / 6;
    if (Lt >= 2 && tid < kAggPerBlk * 6 && A1 < n1) {
        const double* di = H.Dinv[1] + (size_t)A1 * 36 + (tid % 6) * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) d1row[c] = di[c];
        const double* gq = H.geo[1] + (size_t)A1 * 3;
        g1own[0] = gq[0]; g1own[1] = gq[1]; g1own[2] = gq[2];
    }
    const double rz = init ? 0. : D.scal[0];
    // first level-1 chunk into registers
    double r1reg[kL1RU], s1reg[kL1RU], g1reg[kL1GU];
    const int ch_n = n1 < kL1Chunk ? n1 : kL1Chunk;
#pragma unroll
    for (int u = 0; u < kL1RU; u++) {
        const int t = u * kMlBlk + tid;
        r1reg[u] = (t < 6 * ch_n) ? r1_old[t] : 0.;
        s1reg[u] = (!init && t < 6 * ch_n) ? H.S1[t] : 0.;
    }
#pragma unroll
    for (int u = 0; u < kL1GU; u++) {
        const int t = u * kMlBlk + tid;
        g1reg[u] = (Lt >= 2 && t < 3 * ch_n) ? H.geo[1][t] : 0.;
    }
    // small arrays: upper-level offsets, top-inverse rows, own-chain blocks (a handful of values per thread)
    for (int l = 2; l < Lt; l++) stage_to_lds(H.geo[l], dyn + goff[l], 3 * H.n[l]);
    for (int t = tid; t < n_top_rows * ntop; t += kMlBlk) {
        const int rr = t / ntop, c = t % ntop;
        const int grow = (Lt == 1) ? (blockIdx.x * kAggPerBlk * 6 + rr) : (6 * anc[Lt] + rr);
        dyn[top_off + t] = (grow < ntop) ? H.top_inv[(size_t)grow * ntop + c] : 0.;
    }
    for (int l = 2; l < Lt; l++) {
        if (tid < 39)
            dyn[chain_off + (l - 2) * 39 + tid] = (tid < 36) ? H.Dinv[l][(size_t)anc[l] * 36 + tid] : H.geo[l][(size_t)anc[l] * 3 + (tid - 36)];
    }
    STAMP(0);      // 1: prefetch issue
    double alpha = 0.;
    bool bad = false;
    if (!init) {
        const double pAp = block_sum6(part, s6);                          // barrier: everything above has landed
        bad = !(pAp > 0.);
        alpha = bad ? 0. : rz / pAp;
    }
    STAMP(0);      // 2: partial reduction
    // ---- level 1 -> level 2 (or, when level 1 is the top level, straight into the top residual), chunk by chunk
    for (int cb = 0; cb < n1; cb += kL1Chunk) {
        const int cn = (n1 - cb < kL1Chunk) ? n1 - cb : kL1Chunk;
        if (cb > 0) {          // later chunks (graphs > 10k free vertices): loaded here, latency exposed
#pragma unroll
            for (int u = 0; u < kL1RU; u++) {
                const int t = u * kMlBlk + tid;
                r1reg[u] = (t < 6 * cn) ? r1_old[(size_t)6 * cb + t] : 0.;
                s1reg[u] = (!init && t < 6 * cn) ? H.S1[(size_t)6 * cb + t] : 0.;
            }
#pragma unroll
            for (int u = 0; u < kL1GU; u++) {
                const int t = u * kMlBlk + tid;
                g1reg[u] = (Lt >= 2 && t < 3 * cn) ? H.geo[1][(size_t)3 * cb + t] : 0.;
            }
            __syncthreads();   // previous chunk fully consumed
        }
        double* dst_r = (Lt == 1) ? (dyn + top_off + n_top_rows * ntop) : (dyn + c1r);   // L == 1: r1 IS the top residual
#pragma unroll
        for (int u = 0; u < kL1RU; u++) {
            const int t = u * kMlBlk + tid;
            if (t < 6 * cn) dst_r[t] = r1reg[u] - alpha * s1reg[u];
        }
#pragma unroll
        for (int u = 0; u < kL1GU; u++) {
            const int t = u * kMlBlk + tid;
            if (Lt >= 2 && t < 3 * cn) dyn[c1g + t] = g1reg[u];
        }
        __syncthreads();
        if (Lt >= 2) {
            const int p0 = cb / kMlFanout, nPc = (cn + kMlFanout - 1) / kMlFanout;
            const int tasks = nPc * 6 * kMlFanout;
            for (int t0 = 0; t0 < tasks; t0 += kMlBlk) {
                const int t = t0 + tid;
                const int j = t & 7, ak = t >> 3;
                const int A = ak / 6, k = ak % 6;
                const int c = A * kMlFanout + j;
                double sacc = 0.;
                if (t < tasks && c < cn) sacc = restrict_comp(dyn + c1g + c * 3, dyn + c1r + c * 6, k);
                sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4);
                if (t < tasks && j == 0) dyn[roff[2] + p0 * 6 + ak] = sacc;
            }
        }
    }
    __syncthreads();
    STAMP(0);      // 3: level-1 fold + restriction
    // ---- restrict up to the top level: 8 lanes per (parent, component), one child each, xor-shuffle fold
    for (int l = 3; l <= Lt; l++) {
        const int nC = H.n[l - 1], nP = H.n[l];
        const int tasks = nP * 6 * kMlFanout;
        for (int t0 = 0; t0 < tasks; t0 += kMlBlk) {
            const int t = t0 + tid;
            const int j = t & 7, ak = t >> 3;
            const int A = ak / 6, k = ak % 6;
            const int c = A * kMlFanout + j;
            double sacc = 0.;
            if (t < tasks && c < nC) sacc = restrict_comp(dyn + goff[l - 1] + c * 3, dyn + roff[l - 1] + c * 6, k);
            sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4);
            if (t < tasks && j == 0) dyn[roff[l] + ak] = sacc;
        }
        __syncthreads();
    }
    const double* rtop = (Lt == 1) ? (dyn + top_off + n_top_rows * ntop) : (dyn + roff[Lt]);
    if (Lt == 1) {
        if (tid < kAggPerBlk * 6) {
            double sacc = 0.;
            for (int c = 0; c < ntop; c++) sacc += dyn[top_off + tid * ntop + c] * rtop[c];
            sy[tid] = (A1 < n1) ? sacc : 0.;
        }
    } else {
        {   // 8 lanes per top row
            const int row = tid >> 3, j = tid & 7;
            double sacc = 0.;
            if (row < 6) for (int c = j; c < ntop; c += 8) sacc += dyn[top_off + row * ntop + c] * rtop[c];
            sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4);
            if (row < 6 && j == 0) syc[row] = sacc;
        }
        __syncthreads();
        for (int l = Lt - 1; l >= 2; l--) {
            double sacc = 0.;
            if (tid < 6) {
                const double* ch = dyn + chain_off + (l - 2) * 39;
                const double* rr = dyn + roff[l] + anc[l] * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) sacc += ch[tid * 6 + c] * rr[c];
                sacc += prolong_comp(ch + 36, syc, tid);
            }
            __syncthreads();
            if (tid < 6) syc[tid] = sacc;
            __syncthreads();
        }
    }
    STAMP(0);      // 4: restrict + top + down chain
    // ---- own rows: x, r, block-Jacobi part, exact r1 of the own aggregates
    double rv = rv0;
    if (act && !init) {
        const size_t i = (size_t)a * 6 + r;
        rv = rv0 - alpha * apv;
        D.x[i] = xv + alpha * pv;
        D.r[i] = rv;
    }
    sv[tid] = act ? rv : 0.;
    __syncthreads();
    double zz = 0., w = 0.;
    if (act) {
        const int g0 = tid - r;
#pragma unroll
        for (int c = 0; c < 6; c++) zz += mrow[c] * sv[g0 + c];
        const double t0 = sv[g0], t1 = sv[g0 + 1], t2 = sv[g0 + 2];
        const double u0 = geo[0] * t0 + geo[3] * t1 + geo[6] * t2;
        const double u1 = geo[1] * t0 + geo[4] * t1 + geo[7] * t2;
        const double u2 = geo[2] * t0 + geo[5] * t1 + geo[8] * t2;
        if (r < 3) w = (r == 0) ? u0 : (r == 1) ? u1 : u2;
        else {
            const double q0 = sv[g0 + 3], q1 = sv[g0 + 4], q2 = sv[g0 + 5];
            const int k = r - 3;
            const double rq = 0.5 * (geo[k] * q0 + geo[3 + k] * q1 + geo[6 + k] * q2);
            const double dx = geo[9], dy = geo[10], dz = geo[11];
            const double cr = (k == 0) ? dy * u2 - dz * u1 : (k == 1) ? dz * u0 - dx * u2 : dx * u1 - dy * u0;
            w = cr + rq;
        }
    }
    sw[tid] = w;
    __syncthreads();
    STAMP(0);      // 5: x, r, zJ, w
    if (tid < kAggPerBlk * 6) {
        const int la = tid / 6, k = tid % 6;
        double s = 0.;
#pragma unroll
        for (int j = 0; j < kMlFanout; j++) s += sw[(la * kMlFanout + j) * 6 + k];
        sr1[tid] = s;
        if (A1 < n1) r1_new[(size_t)A1 * 6 + k] = s;                       // exact, for the next iteration's recursion
    }
    __syncthreads();
    STAMP(0);      // 6: exact r1
    if (Lt >= 2 && tid < kAggPerBlk * 6) {
        // y1 = D1^-1 r1 + P2 y2
        const int la = tid / 6, k = tid % 6;
        double s = 0.;
        if (A1 < n1) {
#pragma unroll
            for (int c = 0; c < 6; c++) s += d1row[c] * sr1[la * 6 + c];
            s += prolong_comp(g1own, syc, k);
        }
        sy[tid] = s;
    }
    __syncthreads();
    double acc = 0.;
    if (act) {
        const double* y = sy + ((tid / 6) / kMlFanout) * 6;
        double add;
        if (r < 3) {
            const double dx = geo[9], dy = geo[10], dz = geo[11];
            const double vx = y[0] + (y[4] * dz - y[5] * dy);
            const double vy = y[1] + (y[5] * dx - y[3] * dz);
            const double vz = y[2] + (y[3] * dy - y[4] * dx);
            add = geo[r * 3] * vx + geo[r * 3 + 1] * vy + geo[r * 3 + 2] * vz;
        } else {
            const int k = r - 3;
            add = 0.5 * (geo[k * 3] * y[3] + geo[k * 3 + 1] * y[4] + geo[k * 3 + 2] * y[5]);
        }
        zz += add;
        D.z[(size_t)a * 6 + r] = zz;
        acc = rv * zz;
    }
    STAMP(0);      // 7: y1, z
    const double tot = block_sum6(acc, s6);
    if (tid == 0) {
        D.part_b[blockIdx.x] = tot;
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void k_ml_geometry(const PgoDev& D, const MlDev* ml, const double* pose, int l, int n_l, hipStream_t s)
{
    hipLaunchKernelGGL(ml_geometry_kernel, dim3((n_l + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D, ml, pose, l);
}
void k_ml_transform(const PgoDev& D, const MlDev* ml, int f, int work, hipStream_t s)
{
    if (work > 0) hipLaunchKernelGGL(ml_transform_kernel, dim3((work + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D, ml, f);
}
void k_ml_reduce(const MlDev* ml, int l, int blocks36, hipStream_t s)
{
    const long work = (long)blocks36 * 36;
    if (work > 0) hipLaunchKernelGGL(ml_reduce_kernel, dim3((unsigned)((work + kBlk - 1) / kBlk)), dim3(kBlk), 0, s, ml, l);
}
void k_ml_invert(const PgoDev& D, const MlDev* ml, int total_aggs, hipStream_t s)
{
    if (total_aggs > 0) hipLaunchKernelGGL(ml_invert_kernel, dim3((total_aggs + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D, ml);
    hipLaunchKernelGGL(ml_top_kernel, dim3(1), dim3(kBlk), 0, s, D, ml);
}
int g_ml_rows(int nb) { return (nb + kRowsPerBlk - 1) / kRowsPerBlk; }
// the fused finish kernel stages the residuals of levels >= min(2, L) in LDS
size_t ml_cg_lds_bytes(const int* n, int levels);
bool ml_fits_lds(const int* n_per_level, int levels)
{
    return ml_cg_lds_bytes(n_per_level, levels) <= 140 * 1024 && (n_per_level[0] + kMlFanout - 1) / kMlFanout <= kMaxPartials;
}
void k_ml_update(const PgoDev& D, const MlDev* ml, const double* p, double* p0, double* p1, int n_part, int init, hipStream_t s)
{
    hipLaunchKernelGGL(ml_update_kernel, dim3(g_ml_rows(D.nb)), dim3(kMlBlk), 0, s, D, ml, p, p0, p1, n_part, init);
}
void k_ml_finish(const PgoDev& D, const MlDev* ml, hipStream_t s)
{
    hipLaunchKernelGGL(ml_finish_kernel, dim3(g_ml_rows(D.nb)), dim3(kMlBlk), 0, s, D, ml);
}
int g_ml_spmv(int nb) { return (nb + kMlFanout - 1) / kMlFanout; }
void k_ml_spmv(const PgoDev& D, const MlHot& ml, const double* p_old, double* p_new, int n_part, double tol2, hipStream_t s)
{
    hipLaunchKernelGGL(ml_spmv_kernel, dim3(g_ml_spmv(D.nb)), dim3(kSpmvBlk), 0, s, D, ml, p_old, p_new, n_part, tol2);
}
// dynamic LDS of ml_cg_kernel for a hierarchy (n_per_level[0..levels])
size_t ml_cg_lds_bytes(const int* n, int levels)
{
    size_t d = 0;
    for (int l = 2; l <= levels; l++) d += 6 * (size_t)n[l];
    for (int l = 2; l < levels; l++) d += 3 * (size_t)n[l];
    const size_t ntop = 6 * (size_t)n[levels];
    d += ((levels == 1) ? (size_t)kAggPerBlk * 6 : 6) * ntop;
    if (levels > 2) d += (size_t)(levels - 2) * 39;
    d += (levels == 1) ? ntop : (size_t)9 * kL1Chunk;            // level-1 chunk buffer (L == 1: r1 = top residual)
    return d * 8;
}
hipError_t k_ml_cg(const PgoDev& D, const MlHot& ml, const double* p, const double* r1_old, double* r1_new, int n_part,
                   int init, size_t lds, hipStream_t s)
{
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ml_cg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL(ml_cg_kernel, dim3(g_ml_rows(D.nb)), dim3(kMlBlk), lds, s, D, ml, p, r1_old, r1_new, n_part, init);
    return hipSuccess;
}

}  // namespace uzl

#ifdef UZL_STAMPS
extern "C" int uzl_debug_read_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(uzl::g_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return -3;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(uzl::g_stamps), z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
#endif
