// pgo_kernels.hip — gfx950 kernels of the pose-graph half (G1, G3-G7, G9, G10 of SURVEY §8a).
//
// The reference hands the graph to g2o (graph_optimization/src/g2o_optimizer.cpp:137-149):
// EdgeSE3 error/Jacobians, Huber kernel, block normal equations, Levenberg-Marquardt with a sparse
// direct solve.  Here the same mathematics runs as flat f64 kernels over SoA edge arrays and a
// block-CSR of 6x6 blocks; the linear solve is a preconditioned conjugate gradient (design choice of
// this back end; tolerance tight enough to track the direct solve).
//
// Poses live as (t, unit quaternion): the error e = toVectorMQT(Z^-1 Xi^-1 Xj)
// (graph_slam_common/thirdparty/src/isometry3d_mappings.cpp:94-99) is then pure quaternion algebra
// with no matrix->quaternion branches in the hot loop.
#include <hip/hip_ext.h>
#include <mutex>

#include "pgo_device.hpp"

namespace uzl {

// ------------------------------------------------------------------------------------------------
// G1  graph flattening on the device
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlk) void prepare_nodes_kernel(const uzl_node* __restrict__ nodes, int n,
                                                             int xy_only, double* __restrict__ pose)
{
    const int v = blockIdx.x * kBlk + threadIdx.x;
    if (v >= n) return;
    Pose P = pose_from_T(nodes[v].pose);                           // addVertex :160-188
    if (xy_only) P = project_xy(P);                                // :164-170
    store_pose(pose, v, P);
}

__global__ __launch_bounds__(kBlk) void prepare_flat_nodes_kernel(const double* __restrict__ poses12, int n,
                                                                  double* __restrict__ pose)
{
    const int v = blockIdx.x * kBlk + threadIdx.x;
    if (v >= n) return;
    store_pose(pose, v, pose_from_T(poses12 + (size_t)v * 12));
}

__device__ __forceinline__ void store_edge(const Pose& Z, const double* __restrict__ info_in, int k, int e,
                                           double* __restrict__ zinv, double* __restrict__ info)
{
    const Pose A = pose_inv(Z);
    zinv[0 * (size_t)e + k] = A.t.x; zinv[1 * (size_t)e + k] = A.t.y; zinv[2 * (size_t)e + k] = A.t.z;
    zinv[3 * (size_t)e + k] = A.q.w; zinv[4 * (size_t)e + k] = A.q.x; zinv[5 * (size_t)e + k] = A.q.y;
    zinv[6 * (size_t)e + k] = A.q.z;
#pragma unroll
    for (int i = 0; i < 36; i++) info[(size_t)i * e + k] = info_in[i];
}

// src[k] = index of the input edge behind system edge k; odom[k] = 1 for TYPE_2D_WHEEL_ODOMETRY
__global__ __launch_bounds__(kBlk) void prepare_edges_kernel(const uzl_edge* __restrict__ edges,
                                                             const int32_t* __restrict__ src, int e,
                                                             const double* __restrict__ sensors, int n_sensors,
                                                             int xy_only, int odom_params, double* __restrict__ zinv,
                                                             double* __restrict__ info)
{
    const int k = blockIdx.x * kBlk + threadIdx.x;
    if (k >= e) return;
    const uzl_edge* ed = edges + src[k];
    Pose Z = pose_from_T(ed->transform);
    const Pose Df = pose_from_T(ed->displacement_from);
    const Pose Dt = pose_from_T(ed->displacement_to);
    if (ed->type == UZL_EDGE_TYPE_2D_WHEEL_ODOMETRY) {
        if (odom_params) Z = odom_round_trip(Z, fabs(ed->diff_time));   // :209-227
        Z = pose_mul(pose_mul(Df, Z), pose_inv(Dt));               // addOdometryEdge :229
    } else {                                                       // addFeatureEdge :281
        Pose M = Df;
        if (ed->sensor_from >= 0 && ed->sensor_from < n_sensors) M = pose_mul(M, pose_from_T(sensors + 12 * (size_t)ed->sensor_from));
        M = pose_mul(M, Z);
        if (ed->sensor_to >= 0 && ed->sensor_to < n_sensors) M = pose_mul(M, pose_inv(pose_from_T(sensors + 12 * (size_t)ed->sensor_to)));
        Z = pose_mul(M, pose_inv(Dt));
    }
    if (xy_only) Z = project_xy(Z);                                // :231-237, :282-288
    store_edge(Z, ed->information, k, e, zinv, info);
}

__global__ __launch_bounds__(kBlk) void prepare_flat_edges_kernel(const double* __restrict__ meas12,
                                                                  const double* __restrict__ info36, int e,
                                                                  double* __restrict__ zinv, double* __restrict__ info)
{
    const int k = blockIdx.x * kBlk + threadIdx.x;
    if (k >= e) return;
    store_edge(pose_from_T(meas12 + (size_t)k * 12), info36 + (size_t)k * 36, k, e, zinv, info);
}

// ------------------------------------------------------------------------------------------------
// G3  EdgeSE3::computeError [EXT]:  e = toVectorMQT(Z^-1 * Xi^-1 * Xj)
// ------------------------------------------------------------------------------------------------
struct EdgeGeom {
    V3 te, tb;
    Q4 qa, qb, qe;
    double s;
};
__device__ __forceinline__ EdgeGeom edge_geom_of(const PgoDev& D, const Pose& Xi, const Pose& Xj, int k)
{
    const size_t e = (size_t)D.e;
    const V3 ta{D.zinv[0 * e + k], D.zinv[1 * e + k], D.zinv[2 * e + k]};
    EdgeGeom G;
    G.qa = Q4{D.zinv[3 * e + k], D.zinv[4 * e + k], D.zinv[5 * e + k], D.zinv[6 * e + k]};
    const V3 d{Xj.t.x - Xi.t.x, Xj.t.y - Xi.t.y, Xj.t.z - Xi.t.z};
    G.tb = mulTv(qrot(Xi.q), d);
    G.qb = qmul(qconj(Xi.q), Xj.q);
    const V3 rt = mulv(qrot(G.qa), G.tb);
    G.te = V3{rt.x + ta.x, rt.y + ta.y, rt.z + ta.z};
    Q4 qab = qnormalize(qmul(G.qa, G.qb));                          // toCompactQuaternion normalises (:77-82)
    G.s = (qab.w < 0.) ? -1. : 1.;                                  // ... and flips to w >= 0 (:38-44)
    G.qe = Q4{G.s * qab.w, G.s * qab.x, G.s * qab.y, G.s * qab.z};
    return G;
}
__device__ __forceinline__ EdgeGeom edge_geom(const PgoDev& D, const double* __restrict__ pose, int k)
{
    return edge_geom_of(D, load_pose(pose, D.ei[k]), load_pose(pose, D.ej[k]), k);
}

// G9  VertexSE3::oplusImpl [EXT]: X <- X * fromVectorMQT(d)   (isometry3d_mappings.cpp:84-91,117-122).  Written out with contraction
// off: ONE sequence of roundings wherever it is inlined - the evaluation of a trial in the device-resident loop recomputes the trial
// poses of an edge's endpoints (eval_lm_kernel) instead of waiting for oplus to have stored them, and must get the stored bits.
__device__ __forceinline__ Pose retract_pose(const Pose& P0, const double* __restrict__ d)
{
#pragma clang fp contract(off)
    Pose P = P0;
    const Q4 q = P0.q;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    P.t.x = P0.t.x + (((1 - (tyy + tzz)) * d[0] + (txy - twz) * d[1]) + (txz + twy) * d[2]);
    P.t.y = P0.t.y + (((txy + twz) * d[0] + (1 - (txx + tzz)) * d[1]) + (tyz - twx) * d[2]);
    P.t.z = P0.t.z + (((txz - twy) * d[0] + (tyz + twx) * d[1]) + (1 - (txx + tyy)) * d[2]);
    const double w2 = 1. - ((d[3] * d[3] + d[4] * d[4]) + d[5] * d[5]);
    if (w2 >= 0.) {                                                 // identity rotation if w2 < 0
        const double bw = sqrt(w2), bx = d[3], by = d[4], bz = d[5];
        const double aw = ((q.w * bw - q.x * bx) - q.y * by) - q.z * bz;
        const double ax = ((q.w * bx + q.x * bw) + q.y * bz) - q.z * by;
        const double ay = ((q.w * by - q.x * bz) + q.y * bw) + q.z * bx;
        const double az = ((q.w * bz + q.x * by) - q.y * bx) + q.z * bw;
        const double n = 1.0 / sqrt(((aw * aw + ax * ax) + ay * ay) + az * az);
        P.q = Q4{aw * n, ax * n, ay * n, az * n};
    }
    return P;
}
// chi = e^T Omega e, reading Omega from the SoA array
__device__ __forceinline__ double edge_chi(const PgoDev& D, int k, const double* ev)
{
    const size_t e = (size_t)D.e;
    double chi = 0.;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        double s = 0.;
#pragma unroll
        for (int c = 0; c < 6; c++) s += D.info[(size_t)(r * 6 + c) * e + k] * ev[c];
        chi += ev[r] * s;
    }
    return chi;
}

// G5 RobustKernelHuber::robustify [EXT] (rho0, rho1)
__device__ __forceinline__ void huber(double e2, double delta, double& rho0, double& rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.; }
    else { const double sq = sqrt(e2); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}

// activeRobustChi2 over `pose`; block partials -> part_a
// kRetract: `pose` is the estimate the trial starts from, and the trial poses are made here from D.x (retract_pose: the bits oplus stores)
template <bool kRetract = false>
__device__ __forceinline__ void chi2_kernel_body(PgoDev D, const double* __restrict__ pose, double delta, int blk = blockIdx.x, int nblk = gridDim.x)
{
    __shared__ double s4[4];
    double acc = 0.;
    for (int k = D.e_begin + blk * kBlk + threadIdx.x; k < D.e_end; k += nblk * kBlk) {
        const int vi = D.ei[k], vj = D.ej[k];
        Pose Xi = load_pose(pose, vi), Xj = load_pose(pose, vj);
        if (kRetract) {
            const int a = D.v2b[vi], b = D.v2b[vj];
            if (a >= 0) Xi = retract_pose(Xi, D.x + (size_t)a * 6);
            if (b >= 0) Xj = retract_pose(Xj, D.x + (size_t)b * 6);
        }
        const EdgeGeom G = edge_geom_of(D, Xi, Xj, k);
        const double ev[6] = {G.te.x, G.te.y, G.te.z, G.qe.x, G.qe.y, G.qe.z};
        const double chi = edge_chi(D, k, ev);
        double r0 = chi, r1 = 1.;
        if (D.robust[k]) huber(chi, delta, r0, r1);
        acc += r0;
    }
    const double tot = block_sum(acc, s4);
    if (threadIdx.x == 0) D.part_a[blk] = tot;
}
__global__ __launch_bounds__(kBlk) void chi2_kernel(PgoDev D, const double* __restrict__ pose, double delta)
{
    chi2_kernel_body(D, pose, delta);
}

// G10 storeImpl: ||e||_2 per system edge (g2o_optimizer.cpp:124-131)
__global__ __launch_bounds__(kBlk) void edge_error_kernel(PgoDev D, const double* __restrict__ pose,
                                                          double* __restrict__ err)
{
    const int k = blockIdx.x * kBlk + threadIdx.x;
    if (k >= D.e) return;
    const EdgeGeom G = edge_geom(D, pose, k);
    err[k] = sqrt(G.te.x * G.te.x + G.te.y * G.te.y + G.te.z * G.te.z + G.qe.x * G.qe.x + G.qe.y * G.qe.y + G.qe.z * G.qe.z);
}

__global__ __launch_bounds__(kBlk) void poses_out_kernel(const double* __restrict__ pose, int n,
                                                         double* __restrict__ out12)
{
    const int v = blockIdx.x * kBlk + threadIdx.x;
    if (v >= n) return;
    double T[12];
    T_from_pose(load_pose(pose, v), T);
#pragma unroll
    for (int i = 0; i < 12; i++) out12[(size_t)v * 12 + i] = T[i];
}

// ------------------------------------------------------------------------------------------------
// G4 + G6  linearizeOplus + constructQuadraticForm [EXT], one lane per edge.
//   Ji = [[-Ra, 2 Ra [tb]x], [0, -s((wb I - [vb]x)(wa I + [va]x) - vb va^T)]]
//   Jj = [[ Re, 0         ], [0,  we I + [ve]x                              ]]
// (derivation in DESIGN.md, 'Jacobians'; checked against central differences in the tests)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void skew(const double x, const double y, const double z, double* S)
{
    S[0] = 0; S[1] = -z; S[2] = y;
    S[3] = z; S[4] = 0; S[5] = -x;
    S[6] = -y; S[7] = x; S[8] = 0;
}
__device__ __forceinline__ void mat3mul(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

// ------------------------------------------------------------------------------------------------
// The sparse Hessian build (north star): ONE kernel where rounds 1-3 had linearize_kernel (a lane per edge writing two 624-byte slot
// records: H_ac block, the edge's share of H_aa, its share of -b) and assemble_kernel (reading all of them back to sum H_aa and b per
// row).  Here a workgroup owns 42 consecutive block rows and walks their slots - contiguous, sorted by edge - 256 at a time:
//   phase 1: a lane per SLOT (edge, side) reads the slot's record (slot-major: coalesced), recomputes the edge's error and Jacobians
//            (3.5 kflop, done on both sides of an edge: free against the round trip through HBM it replaces) and leaves the slot's
//            H_ac block and its share of H_aa | -b in LDS;
//   phase 2: the workgroup writes the chunk's blocks - one contiguous piece of D.blk - 16 bytes a lane, lane after lane; lane (row, r)
//            adds the shares of its row's slots IN SLOT ORDER (assemble_kernel's order: bit-reproducible, no atomics), and after the
//            last chunk writes H_aa | b once per row and the workgroup's largest diagonal entry (computeLambdaInit).
// The shares never reach HBM: 2 x 336 B per edge less written and read back; the per-slot input (two poses, Z^-1, Omega: 464 B) is read
// by both of an edge's slots.
// Workgroups behind the row blocks compute the chi2 partials (chi2_kernel's lanes, one per edge: computeActiveErrors).
// Sharded solve: a slot whose edge another rank linearises contributes nothing here (its block stays zero).
// ------------------------------------------------------------------------------------------------
constexpr int kLaRows = kBlk / 6;                     // <= 42 rows per pass of a workgroup: lane (row, r) owns row r of H_aa and b[r]
constexpr int kLaShare = 27;                          // doubles per slot in LDS: upper triangle of its share of H_aa (21) | share of -b (6)
constexpr int kLaStage = 37;                          // doubles per slot in LDS: its H_ac block (36) | pad (an odd stride: the lanes' writes spread over the banks)
constexpr int kSlotRec = 44;                          // doubles per slot record: Z^-1 (7) | Omega (36) | pad
// (r, c) -> index in the packed upper triangle, r <= c
__device__ __forceinline__ constexpr int tri6(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }

// The inputs of a slot's edge, SLOT-MAJOR in 16-byte pieces (values 2i, 2i + 1 of slot s at srec[(i * nslots + s) * 2]): the lanes of
// the Hessian build walk slots, and a wave's 64 consecutive slots then read every piece as one contiguous kilobyte.  (Rounds 4-5 kept
// one 352-byte record per EDGE and each lane fetched its own - 64 cache lines per wave instruction; the lane-per-edge kernels read the
// edge-major SoA arrays coalesced and are not concerned.)
__global__ __launch_bounds__(kBlk) void slot_records_kernel(const double* __restrict__ zinv, const double* __restrict__ info, int e,
                                                            const int32_t* __restrict__ slot_edge, int nslots, double* __restrict__ srec)
{
    const int s = blockIdx.x * kBlk + threadIdx.x;
    if (s >= nslots) return;
    const int k = slot_edge[s] >> 1;
    double v[kSlotRec];
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = zinv[(size_t)i * e + k];
#pragma unroll
    for (int i = 0; i < 36; i++) v[7 + i] = info[(size_t)i * e + k];
    v[43] = 0.;
    double2* __restrict__ o = reinterpret_cast<double2*>(srec);
#pragma unroll
    for (int i = 0; i < kSlotRec / 2; i++) o[(size_t)i * nslots + s] = make_double2(v[2 * i], v[2 * i + 1]);
}
void k_slot_records(const double* zinv, const double* info, int e, const int32_t* slot_edge, int nslots, double* srec, hipStream_t s)
{
    if (nslots > 0) hipLaunchKernelGGL(slot_records_kernel, dim3((nslots + kBlk - 1) / kBlk), dim3(kBlk), 0, s, zinv, info, e, slot_edge, nslots, srec);
}
// edge_geom on a slot record (same arithmetic, same numbers)
__device__ __forceinline__ EdgeGeom edge_geom_rec(const double* __restrict__ pose, int vi, int vj, const double* __restrict__ rc)
{
    const Pose Xi = load_pose(pose, vi);
    const Pose Xj = load_pose(pose, vj);
    const V3 ta{rc[0], rc[1], rc[2]};
    EdgeGeom G;
    G.qa = Q4{rc[3], rc[4], rc[5], rc[6]};
    const V3 d{Xj.t.x - Xi.t.x, Xj.t.y - Xi.t.y, Xj.t.z - Xi.t.z};
    G.tb = mulTv(qrot(Xi.q), d);
    G.qb = qmul(qconj(Xi.q), Xj.q);
    const V3 rt = mulv(qrot(G.qa), G.tb);
    G.te = V3{rt.x + ta.x, rt.y + ta.y, rt.z + ta.z};
    Q4 qab = qnormalize(qmul(G.qa, G.qb));
    G.s = (qab.w < 0.) ? -1. : 1.;
    G.qe = Q4{G.s * qab.w, G.s * qab.x, G.s * qab.y, G.s * qab.z};
    return G;
}

// column c of W_j = Omega' Jj (Jj = diag(B11, B22)) and of W_i = Omega' Ji (Ji = [[A11, A12], [0, A22]])
template <int C>
__device__ __forceinline__ void wj_col(const double* Om, const double* B11, const double* B22, double* w)
{
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
        w[rr] = C < 3 ? Om[rr * 6 + 0] * B11[0 * 3 + C] + Om[rr * 6 + 1] * B11[1 * 3 + C] + Om[rr * 6 + 2] * B11[2 * 3 + C]
                      : Om[rr * 6 + 3] * B22[0 * 3 + (C % 3)] + Om[rr * 6 + 4] * B22[1 * 3 + (C % 3)] + Om[rr * 6 + 5] * B22[2 * 3 + (C % 3)];
}
template <int C>
__device__ __forceinline__ void wi_col(const double* Om, const double* A11, const double* A12, const double* A22, double* w)
{
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
        w[rr] = C < 3 ? Om[rr * 6 + 0] * A11[0 * 3 + C] + Om[rr * 6 + 1] * A11[1 * 3 + C] + Om[rr * 6 + 2] * A11[2 * 3 + C]
                      : Om[rr * 6 + 0] * A12[0 * 3 + (C % 3)] + Om[rr * 6 + 1] * A12[1 * 3 + (C % 3)] + Om[rr * 6 + 2] * A12[2 * 3 + (C % 3)] +
                        Om[rr * 6 + 3] * A22[0 * 3 + (C % 3)] + Om[rr * 6 + 4] * A22[1 * 3 + (C % 3)] + Om[rr * 6 + 5] * A22[2 * 3 + (C % 3)];
}
// column C of the blocks, and of the shares of H_aa (upper triangle: rows <= C)
template <int C>
__device__ __forceinline__ void hij_col(const double* Om, const double* A11, const double* A12, const double* A22, const double* B11, const double* B22,
                                        bool write, int side, double* __restrict__ blk_out)
{
    double w[6], hc[6];
    wj_col<C>(Om, B11, B22, w);
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
        hc[rr] = A11[0 * 3 + rr] * w[0] + A11[1 * 3 + rr] * w[1] + A11[2 * 3 + rr] * w[2];
        hc[3 + rr] = A12[0 * 3 + rr] * w[0] + A12[1 * 3 + rr] * w[1] + A12[2 * 3 + rr] * w[2] +
                     A22[0 * 3 + rr] * w[3] + A22[1 * 3 + rr] * w[4] + A22[2 * 3 + rr] * w[5];
    }
#pragma unroll
    for (int rr = 0; rr < 6; rr++) {
        const double v = write ? hc[rr] : 0.;         // (a slot whose neighbour is fixed keeps a zero block)
        if (side == 0) blk_out[rr * 6 + C] = v;       // H_ij, row-major
        else blk_out[C * 6 + rr] = v;                 // its transpose: a column of H_ij is a row of the stored block
    }
}
template <int C>
__device__ __forceinline__ void share_col(const double* Om, const double* A11, const double* A12, const double* A22, const double* B11, const double* B22,
                                          int side, double* __restrict__ mine)
{
    double w[6];
    if (side == 1) {                                   // row j: Jj^T Wj
        wj_col<C>(Om, B11, B22, w);
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            if (rr <= C) mine[tri6(rr, C)] = B11[0 * 3 + rr] * w[0] + B11[1 * 3 + rr] * w[1] + B11[2 * 3 + rr] * w[2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            if (3 + rr <= C) mine[tri6(3 + rr, C)] = B22[0 * 3 + rr] * w[3] + B22[1 * 3 + rr] * w[4] + B22[2 * 3 + rr] * w[5];
    } else {                                           // row i: Ji^T Wi
        wi_col<C>(Om, A11, A12, A22, w);
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            if (rr <= C) mine[tri6(rr, C)] = A11[0 * 3 + rr] * w[0] + A11[1 * 3 + rr] * w[1] + A11[2 * 3 + rr] * w[2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
            if (3 + rr <= C) mine[tri6(3 + rr, C)] = A12[0 * 3 + rr] * w[0] + A12[1 * 3 + rr] * w[1] + A12[2 * 3 + rr] * w[2] +
                                                     A22[0 * 3 + rr] * w[3] + A22[1 * 3 + rr] * w[4] + A22[2 * 3 + rr] * w[5];
    }
}

// row block `rb` of D.rb_ptr (host-side partition: consecutive rows with <= 256 slots and <= 42 rows wherever the graph allows, so that a
// workgroup makes ONE pass: a chunk of slots, a group of rows; hub rows and very large graphs loop).
// The lanes leave their H_ac blocks in LDS and the workgroup writes them out after the barrier - the chunk's blocks are one contiguous
// piece of D.blk - 16 bytes a lane, lane after lane: written by their own lanes (18 x 16 bytes each, 288 bytes apart) the stores were
// half of the kernel's time (ablations: DESIGN_APPENDIX.md, "Hessian build").  131 KB of LDS: one workgroup per CU.  Measured and not kept: blocks and
// shares in turn through ONE 76-KB piece with the lane's state held in registers under 256, so that two workgroups share a CU
// (10k / 50k 34.4 -> 34.3 us, config 2 13.0 -> 17.1: two more barriers and a few spills on a path that is a latency chain).
__device__ __forceinline__ void hessian_rows_body(PgoDev D, const double* __restrict__ pose, double delta, int rb)
{
    __shared__ double s4[4];
    __shared__ double sh[kBlk * kLaShare];             // 55 KB: the chunk's shares of H_aa | -b
    __shared__ double st[kBlk * kLaStage];             // 76 KB: the chunk's H_ac blocks on their way out
    const int tid = threadIdx.x;
    double dmax = 0.;
    const int rows_begin = D.rb_ptr[rb], rows_end = D.rb_ptr[rb + 1];
    for (int row0 = rows_begin; row0 < rows_end; row0 += kLaRows) {
        const int row1 = (row0 + kLaRows < rows_end) ? row0 + kLaRows : rows_end;
        const int s_begin = D.row_ptr[row0], s_end = D.row_ptr[row1];
        const int a = row0 + tid / 6, r = tid % 6;
        const bool rowlane = tid < kLaRows * 6 && a < row1;
        const int ra0 = rowlane ? D.row_ptr[a] : 0, ra1 = rowlane ? D.row_ptr[a + 1] : 0;
        double h[6] = {0., 0., 0., 0., 0., 0.};
        double g = 0.;
        for (int base = s_begin; base < s_end; base += kBlk) {
            const int s = base + tid;
            double* __restrict__ mine = sh + (size_t)tid * kLaShare;
            double* __restrict__ blk_out = st + (size_t)tid * kLaStage;
            bool live = false;
            int side = 0;
            int4 sm = make_int4(0, 0, 0, 0);
            if (s < s_end) {
                sm = D.smeta[s];
                const int k = sm.x >> 1;
                side = sm.x & 1;
                live = k >= D.e_begin && k < D.e_end;
            }
            double Om[36], Oe[6], A11[9], A12[9], A22[9], B11[9], B22[9];      // Omega' (row-major, robustified), Omega' e, Ji, Jj
            if (!live) {
                if (s < s_end) {                       // (sharded solve: an edge of another rank - block and shares are that rank's)
#pragma unroll
                    for (int i = 0; i < 36; i++) blk_out[i] = 0.;
#pragma unroll
                    for (int i = 0; i < kLaShare; i++) mine[i] = 0.;
                }
            } else {
                // the slot's record: 22 x 16 bytes, each piece contiguous over the wave's slots
                double rc[kSlotRec];
                {
                    const double2* __restrict__ src = reinterpret_cast<const double2*>(D.srec) + s;
#pragma unroll
                    for (int i = 0; i < kSlotRec / 2; i++) { const double2 v = src[(size_t)i * D.nslots]; rc[2 * i] = v.x; rc[2 * i + 1] = v.y; }
                }
                const EdgeGeom G = edge_geom_rec(pose, sm.y, sm.z, rc);
                const double ev[6] = {G.te.x, G.te.y, G.te.z, G.qe.x, G.qe.y, G.qe.z};
#pragma unroll
                for (int i = 0; i < 36; i++) Om[i] = rc[7 + i];
                double chi = 0.;
#pragma unroll
                for (int rr = 0; rr < 6; rr++) {
                    double sacc = 0.;
#pragma unroll
                    for (int c = 0; c < 6; c++) sacc += Om[rr * 6 + c] * ev[c];
                    Oe[rr] = sacc;
                    chi += ev[rr] * sacc;
                }
                double r0 = chi, r1 = 1.;
                if (sm.w >> 30) huber(chi, delta, r0, r1);
#pragma unroll
                for (int i = 0; i < 36; i++) Om[i] *= r1;
#pragma unroll
                for (int i = 0; i < 6; i++) Oe[i] *= r1;
                // ---- Jacobian blocks
                const M33 Ra = qrot(G.qa);
                const M33 Re = qrot(G.qe);
                {
                    double S[9], T[9];
                    skew(G.tb.x, G.tb.y, G.tb.z, S);
                    mat3mul(Ra.m, S, T);
#pragma unroll
                    for (int i = 0; i < 9; i++) { A11[i] = -Ra.m[i]; A12[i] = 2. * T[i]; B11[i] = Re.m[i]; }
                    double Sa[9], Sb[9], L[9], R[9], P[9];
                    skew(G.qa.x, G.qa.y, G.qa.z, Sa);
                    skew(G.qb.x, G.qb.y, G.qb.z, Sb);
#pragma unroll
                    for (int i = 0; i < 9; i++) {
                        const double id = (i % 4 == 0) ? 1. : 0.;
                        L[i] = id * G.qb.w - Sb[i];
                        R[i] = id * G.qa.w + Sa[i];
                    }
                    mat3mul(L, R, P);
                    const double vb[3] = {G.qb.x, G.qb.y, G.qb.z}, va[3] = {G.qa.x, G.qa.y, G.qa.z};
#pragma unroll
                    for (int rr = 0; rr < 3; rr++)
#pragma unroll
                        for (int c = 0; c < 3; c++) A22[rr * 3 + c] = -G.s * (P[rr * 3 + c] - vb[rr] * va[c]);
                    double Se[9];
                    skew(G.qe.x, G.qe.y, G.qe.z, Se);
#pragma unroll
                    for (int i = 0; i < 9; i++) B22[i] = ((i % 4 == 0) ? G.qe.w : 0.) + Se[i];
                }
                // ---- H_ij = Ji^T W_j, W_j = Omega' Jj: ONE code path for both sides of the edge, so that the block row i stores and the
                //      transposed block row j stores are the same numbers
                const bool write = (sm.w & 0x3fffffff) != 0;                    // the edge's other endpoint has a row (it is not fixed)
                hij_col<0>(Om, A11, A12, A22, B11, B22, write, side, blk_out); hij_col<1>(Om, A11, A12, A22, B11, B22, write, side, blk_out);
                hij_col<2>(Om, A11, A12, A22, B11, B22, write, side, blk_out); hij_col<3>(Om, A11, A12, A22, B11, B22, write, side, blk_out);
                hij_col<4>(Om, A11, A12, A22, B11, B22, write, side, blk_out); hij_col<5>(Om, A11, A12, A22, B11, B22, write, side, blk_out);
                // ---- the slot's share of H_aa (upper triangle: the sum is then symmetric to the last bit) and of -b
                share_col<0>(Om, A11, A12, A22, B11, B22, side, mine); share_col<1>(Om, A11, A12, A22, B11, B22, side, mine);
                share_col<2>(Om, A11, A12, A22, B11, B22, side, mine); share_col<3>(Om, A11, A12, A22, B11, B22, side, mine);
                share_col<4>(Om, A11, A12, A22, B11, B22, side, mine); share_col<5>(Om, A11, A12, A22, B11, B22, side, mine);
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    if (side == 1) {                   // Jj^T Omega' e
                        mine[21 + rr] = B11[0 * 3 + rr] * Oe[0] + B11[1 * 3 + rr] * Oe[1] + B11[2 * 3 + rr] * Oe[2];
                        mine[24 + rr] = B22[0 * 3 + rr] * Oe[3] + B22[1 * 3 + rr] * Oe[4] + B22[2 * 3 + rr] * Oe[5];
                    } else {                           // Ji^T Omega' e
                        mine[21 + rr] = A11[0 * 3 + rr] * Oe[0] + A11[1 * 3 + rr] * Oe[1] + A11[2 * 3 + rr] * Oe[2];
                        mine[24 + rr] = A12[0 * 3 + rr] * Oe[0] + A12[1 * 3 + rr] * Oe[1] + A12[2 * 3 + rr] * Oe[2] +
                                        A22[0 * 3 + rr] * Oe[3] + A22[1 * 3 + rr] * Oe[4] + A22[2 * 3 + rr] * Oe[5];
                    }
                }
            }
            __syncthreads();
            {
                const int nq = (s_end - base < kBlk ? s_end - base : kBlk) * 18;
                double2* __restrict__ out = reinterpret_cast<double2*>(D.blk + (size_t)base * 36);
                int q = tid / 18, part = tid - q * 18;
                for (int gq = tid; gq < nq; gq += kBlk) {
                    const double* __restrict__ src = st + (size_t)q * kLaStage + 2 * part;
                    out[gq] = make_double2(src[0], src[1]);
                    q += kBlk / 18; part += kBlk % 18;
                    if (part >= 18) { part -= 18; q++; }
                }
            }
            if (rowlane) {                             // this chunk's slots of the lane's row, in slot order
                const int q0 = ra0 > base ? ra0 : base, q1 = ra1 < base + kBlk ? ra1 : base + kBlk;
                for (int q = q0; q < q1; q++) {
                    const double* __restrict__ dc = sh + (size_t)(q - base) * kLaShare;
#pragma unroll
                    for (int c = 0; c < 6; c++) h[c] += dc[r <= c ? tri6(r, c) : tri6(c, r)];
                    g += dc[21 + r];
                }
            }
            __syncthreads();
        }
        if (rowlane) {
            double* __restrict__ out = D.hdiag + (size_t)a * 36 + r * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) out[c] = h[c];
            D.b[(size_t)a * 6 + r] = -g;
            dmax = fmax(dmax, fabs(h[r]));
        }
    }
    const double m = block_max(dmax, s4);
    if (tid == 0) D.part_c[rb] = m;
}
// grid = g_rows row workgroups + g_edges chi2 workgroups
__global__ __launch_bounds__(kBlk) void hessian_kernel(PgoDev D, const double* __restrict__ pose, double delta, int g_rows, int g_edges)
{
    if ((int)blockIdx.x < g_rows) hessian_rows_body(D, pose, delta, blockIdx.x);
    else chi2_kernel_body(D, pose, delta, (int)blockIdx.x - g_rows, g_edges);
}

// max |H_jj| over the assembled diagonal blocks -> part_c (sharded solve: after the all-reduce of hdiag)
__global__ __launch_bounds__(kBlk) void diagmax_kernel(PgoDev D)
{
    __shared__ double s4[4];
    double m = 0.;
    for (int i = blockIdx.x * kBlk + threadIdx.x; i < D.nb * 6; i += gridDim.x * kBlk) m = fmax(m, fabs(D.hdiag[(size_t)(i / 6) * 36 + (i % 6) * 7]));
    const double t = block_max(m, s4);
    if (threadIdx.x == 0) D.part_c[blockIdx.x] = t;
}

// one block: final reductions of the LM bookkeeping scalars.
//   what = 0: scal[4] = sum(part_a[0..na))            (chi2)
//   what = 1: ... and scal[5] = sum(part_b[0..nb_))    (chi2 + computeScale)
//   what = 2: ... and scal[6] = max(part_c[0..nc))     (chi2 + max diagonal for computeLambdaInit)
__device__ __forceinline__ void finalize_kernel_body(PgoDev D, int na, int nb_, int nc, int what)
{
    __shared__ double s4[4];
    const double chi = sum_partials(D.part_a, na, s4);
    if (threadIdx.x == 0) D.scal[4] = chi;
    if (what == 1) {
        const double sc = sum_partials(D.part_b, nb_, s4);
        if (threadIdx.x == 0) D.scal[5] = sc;
    }
    if (what == 2) {
        double v = 0.;
        for (int i = threadIdx.x; i < nc; i += kBlk) v = fmax(v, D.part_c[i]);
        const double m = block_max(v, s4);
        if (threadIdx.x == 0) D.scal[6] = m;
    }
}
__global__ __launch_bounds__(kBlk) void finalize_kernel(PgoDev D, int na, int nb_, int nc, int what)
{
    finalize_kernel_body(D, na, nb_, nc, what);
}

// ------------------------------------------------------------------------------------------------
// G8  linear solve: block-Jacobi PCG on (H + lambda I) dx = b
// ------------------------------------------------------------------------------------------------
// M_a^-1 = (H_aa + lambda I)^-1 through a 6x6 Cholesky, one lane per row block
__global__ __launch_bounds__(kBlk) void precond_kernel(PgoDev D)
{
    const int a = blockIdx.x * kBlk + threadIdx.x;
    if (a >= D.nb) return;
    const double lambda = D.scal[3];
    double A[36];
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = D.hdiag[(size_t)a * 36 + i] + ((i % 7 == 0) ? lambda : 0.);
    double out[36];
    spd_inverse6(A, out);
    double* __restrict__ dst = D.minv + (size_t)a * 36;
#pragma unroll
    for (int i = 0; i < 36; i++) dst[i] = out[i];
}

// PCG with two launches per iteration and no grid barrier.  Scalars never cross a launch as "the value
// computed by one block": every block re-reduces the previous launch's block partials in the same order.
//   pcg_init   : x = 0, r = b, z = M^-1 r, p0 = p1 = 0; partials of r.z -> part_b; flags cleared
//   pcg_spmv   : rz = sum(part_b); beta = rz / rz_prev (0 in iteration 0); p_new = z + beta p_old for the own
//                row AND, recomputed on the fly, for every neighbour column (so p never needs a launch of its
//                own); Ap = (H + lambda I) p_new; partials of p.Ap -> part_a; block 0: rz -> scal[0], stop test
//   pcg_update : alpha = rz / sum(part_a); x += alpha p; r -= alpha Ap; z = M^-1 r; partials of r.z -> part_b;
//                block 0: rz_prev = rz, ++iteration
// p is double-buffered (p_old read, p_new written) because neighbours read p_old while rows write p_new.
__global__ __launch_bounds__(kBlk) void pcg_init_kernel(PgoDev D, double* __restrict__ p0, double* __restrict__ p1)
{
    __shared__ double s4[4];
    __shared__ double sv[kBlk];
    const int per = kBlk / 6;                        // 42 row blocks per block
    double acc = 0.;
    for (int base = blockIdx.x * per; base < D.nb; base += gridDim.x * per) {
        const int a = base + (int)threadIdx.x / 6, r = (int)threadIdx.x % 6;
        const bool act = (int)threadIdx.x < per * 6 && a < D.nb;
        double rv = 0.;
        if (act) { rv = D.b[(size_t)a * 6 + r]; }
        __syncthreads();
        sv[threadIdx.x] = rv;
        __syncthreads();
        if (act) {
            const double* __restrict__ m = D.minv + (size_t)a * 36 + r * 6;
            const int g0 = (int)threadIdx.x - r;
            double zz = 0.;
#pragma unroll
            for (int c = 0; c < 6; c++) zz += m[c] * sv[g0 + c];
            const size_t i = (size_t)a * 6 + r;
            D.x[i] = 0.; D.xs[i] = 0.; D.r[i] = rv; D.z[i] = zz; p0[i] = 0.; p1[i] = 0.;
            acc += rv * zz;
        }
    }
    const double tot = block_sum(acc, s4);
    if (threadIdx.x == 0) {
        D.part_b[blockIdx.x] = tot;
        if (blockIdx.x == 0) { D.flags[0] = 0; D.flags[1] = 0; D.flags[2] = 0; D.scal[2] = 1.; }
    }
}

// One wave per row block: 10 slot-groups of 6 lanes walk the row's contiguous slots (lane (g, r) owns row r
// of slot s0+g+10k), then a shuffle tree folds the groups.
__global__ __launch_bounds__(kBlk) void pcg_spmv_kernel(PgoDev D, const double* __restrict__ p_old,
                                                        double* __restrict__ p_new, int n_part, double tol2)
{
    __shared__ double s4[4];
    if (D.flags[0]) return;
    const int it = D.flags[1];
    const double rz = sum_partials(D.part_b, n_part, s4);
    const double beta = (it == 0) ? 0. : rz / D.scal[2];
    const double thresh = (it == 0) ? (tol2 * D.scal[8]) * rz : D.scal[1];      // scal[8]: the LM iteration's tightening of pcg_tol^2 (uzl_pgo.hip)
    const double lambda = D.scal[3];
    const int lane = threadIdx.x & 63, g = lane / 6, r = lane % 6;
    const bool act = lane < 60;
    const int wave = blockIdx.x * (kBlk / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (kBlk / 64);
    double dot = 0.;
    for (int a = wave; a < D.nb; a += nwaves) {
        const int s0 = D.row_ptr[a], s1 = D.row_ptr[a + 1];
        double acc = 0., pr = 0.;
        if (g == 0) {
            const double* __restrict__ h = D.hdiag + (size_t)a * 36 + r * 6;
            const double* __restrict__ zv = D.z + (size_t)a * 6;
            const double* __restrict__ po = p_old + (size_t)a * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double pc = zv[c] + beta * po[c];
                acc += h[c] * pc;
                if (c == r) pr = pc;
            }
            acc += lambda * pr;
            p_new[(size_t)a * 6 + r] = pr;
        }
        if (act) {
            for (int s = s0 + g; s < s1; s += 10) {
                const int c = D.col[s];
                if (c >= 0) {
                    const double2* __restrict__ bk = reinterpret_cast<const double2*>(D.blk + (size_t)s * 36 + r * 6);
                    const double2* __restrict__ zv = reinterpret_cast<const double2*>(D.z + (size_t)c * 6);
                    const double2* __restrict__ po = reinterpret_cast<const double2*>(p_old + (size_t)c * 6);
                    const double2 b0 = bk[0], b1 = bk[1], b2 = bk[2];
                    const double2 z0 = zv[0], z1 = zv[1], z2 = zv[2];
                    const double2 o0 = po[0], o1 = po[1], o2 = po[2];
                    acc += b0.x * (z0.x + beta * o0.x) + b0.y * (z0.y + beta * o0.y) + b1.x * (z1.x + beta * o1.x) +
                           b1.y * (z1.y + beta * o1.y) + b2.x * (z2.x + beta * o2.x) + b2.y * (z2.y + beta * o2.y);
                }
            }
        }
        // fold the 10 groups: offsets 48, 24, 12, 6 lanes
        double v;
        v = __shfl_down(acc, 48); if (lane + 48 < 60) acc += v;
        v = __shfl_down(acc, 24); if (lane + 24 < 48) acc += v;
        v = __shfl_down(acc, 12); if (lane + 12 < 24) acc += v;
        v = __shfl_down(acc, 6);  if (lane + 6 < 12) acc += v;
        if (lane < 6) {
            D.ap[(size_t)a * 6 + r] = acc;
            dot += acc * pr;
        }
    }
    const double tot = block_sum(dot, s4);
    if (threadIdx.x == 0) {
        D.part_a[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            D.scal[0] = rz;
            if (it == 0) { D.scal[1] = thresh; D.scal[11] = rz; }
            if (!(rz > thresh) ) D.flags[0] = 1;       // converged (or rz == 0 / NaN): x from the last update is final
            if (!(rz >= 0.)) D.flags[2] = 1;           // negative / NaN r.M^-1 r: breakdown
        }
    }
}

// alpha = rz / p.Ap; x += alpha p; r -= alpha Ap; z = Minv r; partials of r.z -> part_b
__global__ __launch_bounds__(kBlk) void pcg_update_kernel(PgoDev D, const double* __restrict__ p, int n_part)
{
    __shared__ double s4[4];
    __shared__ double sv[kBlk];
    if (D.flags[0]) return;
    const double pAp = sum_partials(D.part_a, n_part, s4);
    const double rz = D.scal[0];
    const bool bad = !(pAp > 0.);
    const double alpha = bad ? 0. : rz / pAp;
    const int per = kBlk / 6;
    double acc = 0.;
    for (int base = blockIdx.x * per; base < D.nb; base += gridDim.x * per) {
        const int a = base + (int)threadIdx.x / 6, r = (int)threadIdx.x % 6;
        const bool act = (int)threadIdx.x < per * 6 && a < D.nb;
        double rv = 0.;
        if (act) {
            const size_t i = (size_t)a * 6 + r;
            D.x[i] += alpha * p[i];
            rv = D.r[i] - alpha * D.ap[i];
            D.r[i] = rv;
        }
        __syncthreads();
        sv[threadIdx.x] = rv;
        __syncthreads();
        if (act) {
            const double* __restrict__ m = D.minv + (size_t)a * 36 + r * 6;
            const int g0 = (int)threadIdx.x - r;
            double zz = 0.;
#pragma unroll
            for (int c = 0; c < 6; c++) zz += m[c] * sv[g0 + c];
            D.z[(size_t)a * 6 + r] = zz;
            acc += rv * zz;
        }
    }
    const double tot = block_sum(acc, s4);
    if (threadIdx.x == 0) {
        D.part_b[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            D.scal[2] = rz;                          // rz_prev for the next pcg_spmv
            D.flags[1] += 1;
            if (bad) { D.flags[0] = 1; D.flags[2] = 1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G9  VertexSE3::oplusImpl [EXT] over the vertices (retract_pose), plus the partials of computeScale = sum dx (lambda dx + b) -> part_b
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void oplus_kernel_body(PgoDev D, const double* __restrict__ pose_in,
                                                     double* __restrict__ pose_out, int blk = blockIdx.x, int nblk = gridDim.x)
{
    __shared__ double s4[4];
    const double lambda = D.scal[3];
    double acc = 0.;
    for (int v = blk * kBlk + threadIdx.x; v < D.n; v += nblk * kBlk) {
        Pose P = load_pose(pose_in, v);
        const int a = D.v2b[v];
        if (a >= 0) {
            const double* __restrict__ dx = D.x + (size_t)a * 6;
            const double* __restrict__ bb = D.b + (size_t)a * 6;
            const double d[6] = {dx[0], dx[1], dx[2], dx[3], dx[4], dx[5]};
#pragma unroll
            for (int i = 0; i < 6; i++) acc += d[i] * (lambda * d[i] + bb[i]);
            P = retract_pose(P, d);
        }
        store_pose(pose_out, v, P);
    }
    const double tot = block_sum(acc, s4);
    if (threadIdx.x == 0) D.part_b[blk] = tot;
}
__global__ __launch_bounds__(kBlk) void oplus_kernel(PgoDev D, const double* __restrict__ pose_in, double* __restrict__ pose_out)
{
    oplus_kernel_body(D, pose_in, pose_out);
}

static inline int grid_for(int items, int per_block, int cap)
{
    int g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return g > cap ? cap : g;
}

__device__ __forceinline__ void residual_guard_kernel_body(PgoDev D);
// ------------------------------------------------------------------------------------------------
// slot twins of the device-resident LM loop (pgo_types.hpp: LmSlot / LmDev): graph = blockIdx.z, arguments from its slot, and every
// kernel predicates itself on the graph's phase - a pass is a fixed launch sequence (uzl_pgo_lm.hip)
// ------------------------------------------------------------------------------------------------
// grid = the largest graph's row workgroups (g_rows_launch) + chi2 workgroups
__global__ __launch_bounds__(kBlk) void hessian_lm_kernel(const LmSlot* __restrict__ slots, int g_rows_launch)
{
    const LmSlot& S = slots[blockIdx.z];
    const LmDev* lm = S.lm;
    if (lm->phase != kLmLin) return;
    const int b = blockIdx.x;
    if (b < g_rows_launch) { if (b < S.g_asm) hessian_rows_body(S.D, S.pose[lm->cur], lm->delta, b); }
    // chi2 of the linearisation point (computeActiveErrors): the loop needs it in its first iteration only - later ones start from an
    // accepted trial, whose chi2 the evaluation of that trial computed from the same poses with the same kernel body (lm_head_kernel
    // takes chi_cur from the partials at it == 0 and carries it afterwards).  Skipping the workgroups saves their second pass over every
    // edge's measurement and information matrix (17 MB of the kernel's 92 MB of traffic at 10k / 50k).
    else if (lm->it == 0 && b - g_rows_launch < S.g_edges) chi2_kernel_body(S.D, S.pose[lm->cur], lm->delta, b - g_rows_launch, S.g_edges);
}
// the evaluation of a trial runs once its solve has ended without a breakdown (lm_tail_kernel sorts the rest out)
__device__ __forceinline__ bool lm_evaluates(const LmDev* lm) { return lm->phase == kLmSolve && lm->flags[0] != 0 && lm->flags[2] == 0; }
// retraction and chi2 of the trial in ONE launch: the workgroups behind the vertex workgroups evaluate the edges at trial poses they make
// themselves (one launch less per trial: ~4 us; an edge lane's two retractions are ~100 flop beside its error's 500)
__global__ __launch_bounds__(kBlk) void eval_lm_kernel(const LmSlot* __restrict__ slots, int g_oplus_launch)
{
    const LmSlot& S = slots[blockIdx.z];
    const LmDev* lm = S.lm;
    if (!lm_evaluates(lm)) return;
    const int b = blockIdx.x;
    if (b < g_oplus_launch) { if (b < S.g_oplus) oplus_kernel_body(S.D, S.pose[lm->cur], S.pose[lm->cur ^ 1], b, S.g_oplus); }
    else if (b - g_oplus_launch < S.g_edges) chi2_kernel_body<true>(S.D, S.pose[lm->cur], lm->delta, b - g_oplus_launch, S.g_edges);
}
hipError_t kl_linearize(const LmSlot* sl, int nslots, int g_edges, int g_asm, hipStream_t s)
{
    hipLaunchKernelGGL(hessian_lm_kernel, dim3(g_asm + g_edges, 1, nslots), dim3(kBlk), 0, s, sl, g_asm);
    return hipSuccess;
}
void kl_eval(const LmSlot* sl, int nslots, int g_edges, int g_oplus, hipStream_t s)
{
    hipLaunchKernelGGL(eval_lm_kernel, dim3(g_oplus + g_edges, 1, nslots), dim3(kBlk), 0, s, sl, g_oplus);
}

int g_edges_for(int e) { return grid_for(e, kBlk, kMaxPartials); }

int g_oplus_for(int n) { return grid_for(n, kBlk, kMaxPartials); }

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

void k_prepare_nodes(const uzl_node* nodes, int n, int xy, double* pose, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(prepare_nodes_kernel, dim3((n + kBlk - 1) / kBlk), dim3(kBlk), 0, s, nodes, n, xy, pose);
}
void k_prepare_flat_nodes(const double* poses12, int n, double* pose, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(prepare_flat_nodes_kernel, dim3((n + kBlk - 1) / kBlk), dim3(kBlk), 0, s, poses12, n, pose);
}
void k_prepare_edges(const uzl_edge* edges, const int32_t* src, int e, const double* sensors, int ns, int xy, int odom_params,
                     double* zinv, double* info, hipStream_t s)
{
    if (e > 0) hipLaunchKernelGGL(prepare_edges_kernel, dim3((e + kBlk - 1) / kBlk), dim3(kBlk), 0, s, edges, src, e, sensors, ns, xy, odom_params, zinv, info);
}
void k_prepare_flat_edges(const double* meas12, const double* info36, int e, double* zinv, double* info, hipStream_t s)
{
    if (e > 0) hipLaunchKernelGGL(prepare_flat_edges_kernel, dim3((e + kBlk - 1) / kBlk), dim3(kBlk), 0, s, meas12, info36, e, zinv, info);
}
int k_chi2(const PgoDev& D, const double* pose, double delta, hipStream_t s)
{
    const int g = grid_for(D.e_end - D.e_begin, kBlk, kMaxPartials);
    hipLaunchKernelGGL(chi2_kernel, dim3(g), dim3(kBlk), 0, s, D, pose, delta);
    return g;
}
// chi2 of the trial `pose` (+) D.x: eval_lm_kernel's edge workgroups as a launch of their own (the host-driven loop; same body, same bits)
__global__ __launch_bounds__(kBlk) void chi2_trial_kernel(PgoDev D, const double* __restrict__ pose, double delta)
{
    chi2_kernel_body<true>(D, pose, delta);
}
int k_chi2_trial(const PgoDev& D, const double* pose, double delta, hipStream_t s)
{
    const int g = grid_for(D.e_end - D.e_begin, kBlk, kMaxPartials);
    hipLaunchKernelGGL(chi2_trial_kernel, dim3(g), dim3(kBlk), 0, s, D, pose, delta);
    return g;
}
// the Hessian build (G3-G6): *g_edges chi2 partials in part_a, *g_rows diagonal maxima in part_c
// (with_chi2 = false: the caller carries chi2 over from the accepted trial - the chi2 workgroups are not launched, part_a keeps that trial's partials)
hipError_t k_hessian(const PgoDev& D, const double* pose, double delta, int* g_edges, int* g_rows, hipStream_t s, bool with_chi2, hipEvent_t ev_a, hipEvent_t ev_b)
{
    *g_edges = grid_for(D.e_end - D.e_begin, kBlk, kMaxPartials);
    *g_rows = D.n_rb;
    const dim3 grid(*g_rows + (with_chi2 ? *g_edges : 0));
    if (ev_a) hipExtLaunchKernelGGL(hessian_kernel, grid, dim3(kBlk), 0, s, ev_a, ev_b, 0, D, pose, delta, *g_rows, *g_edges);      // profiling: the dispatch's own timestamps
    else hipLaunchKernelGGL(hessian_kernel, grid, dim3(kBlk), 0, s, D, pose, delta, *g_rows, *g_edges);
    return hipSuccess;
}
int k_diagmax(const PgoDev& D, hipStream_t s)
{
    const int g = grid_for(D.nb * 6, kBlk, kMaxPartials);
    hipLaunchKernelGGL(diagmax_kernel, dim3(g), dim3(kBlk), 0, s, D);
    return g;
}
// hands scal[0..8) and flags[0..4) to the host through pinned coherent memory (PgoHostScal); seq last
__global__ __launch_bounds__(64) void publish_kernel(const double* __restrict__ scal, const int32_t* __restrict__ flags,
                                                    PgoHostScal* __restrict__ out, uint32_t seq)
{
    // (system-scope stores into the uncached host block, acknowledged before the sequence word goes: as lm_tail_kernel, no fence - a
    //  system-scope release also writes the L2's dirty lines back, and the host reads nothing of them)
    const int t = threadIdx.x;
    if (t < 8) __hip_atomic_store(&out->scal[t], scal[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (t < 12) __hip_atomic_store(&out->flags[t - 8], flags[t - 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    publish_wait_own_stores();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&out->seq, seq, UZL_PUBLISH_SEQ_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void set_scalar_kernel(double* __restrict__ dst, double v) { *dst = v; }
__global__ void set_scalar2_kernel(double* __restrict__ dst_a, double va, double* __restrict__ dst_b, double vb) { *dst_a = va; *dst_b = vb; }
// lambda, the floor factor on pcg_tol^2 and the step accuracy of this trial's solve
__global__ void set_trial_kernel(double* __restrict__ scal, double lambda, double tol_f2, double eps_t, double eps_r)
{
    scal[3] = lambda; scal[8] = tol_f2; scal[12] = eps_t; scal[13] = eps_r;
}

// ------------------------------------------------------------------------------------------------
// The stop test of the block-Jacobi path (progress_decide, pgo_device.hpp): a launch of its own between the iterations.
// Small systems: one workgroup does it all.  Larger ones: a grid of kProgressChunk-element blocks leaves two partials each in part_c
// (free between assemble and the next linearisation) and a one-workgroup launch behind it folds them and decides - no hand-off
// between workgroups inside a launch (per-XCD L2s are not coherent with each other), nothing depends on block timing.
// ------------------------------------------------------------------------------------------------
constexpr int kProgressChunk = 2048, kProgressMaxBlocks = kMaxPartials / 2;
__host__ __device__ inline int progress_blocks(int nb) { const int g = (nb * 6 + kProgressChunk - 1) / kProgressChunk; return g < 1 ? 1 : (g > kProgressMaxBlocks ? kProgressMaxBlocks : g); }
// stage 0: partials of block blockIdx.x (of nblk); with nblk == 1 also the decision
__device__ __forceinline__ void pcg_progress_kernel_body(PgoDev D)
{
    __shared__ double st[4], sr[4];
    if (D.flags[0]) return;
    const int n = D.nb * 6, nblk = progress_blocks(D.nb);
    const int per = (n + nblk - 1) / nblk, i0 = blockIdx.x * per, i1 = (i0 + per < n) ? i0 + per : n;
    double mt = 0., mr = 0.;
    for (int i = i0 + threadIdx.x; i < i1; i += kBlk) {
        const double x = D.x[i], d = fabs(x - D.xs[i]);
        D.xs[i] = x;
        if (i % 6 < 3) mt = fmax(mt, d); else mr = fmax(mr, d);
    }
    for (int o = 32; o; o >>= 1) { mt = fmax(mt, __shfl_xor(mt, o)); mr = fmax(mr, __shfl_xor(mr, o)); }
    if ((threadIdx.x & 63) == 0) { st[threadIdx.x >> 6] = mt; sr[threadIdx.x >> 6] = mr; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    mt = fmax(fmax(st[0], st[1]), fmax(st[2], st[3])); mr = fmax(fmax(sr[0], sr[1]), fmax(sr[2], sr[3]));
    if (nblk == 1) { progress_decide(D, mt, mr, D.scal[0]); return; }
    double* __restrict__ pc = D.part_c + 2 * blockIdx.x;
    pc[0] = mt; pc[1] = mr;
}
// stage 1 (nblk > 1): one wave folds the partials
__device__ __forceinline__ void pcg_progress_final_body(PgoDev D)
{
    if (D.flags[0]) return;
    const int nblk = progress_blocks(D.nb);
    if (nblk == 1) return;
    const int lane = threadIdx.x;
    double mt = 0., mr = 0.;
    for (int w = lane; w < nblk; w += 64) {
        const double* __restrict__ pc = D.part_c + 2 * w;
        mt = fmax(mt, pc[0]); mr = fmax(mr, pc[1]);
    }
    for (int o = 32; o; o >>= 1) { mt = fmax(mt, __shfl_xor(mt, o)); mr = fmax(mr, __shfl_xor(mr, o)); }
    if (lane == 0) progress_decide(D, mt, mr, D.scal[0]);
}
__global__ __launch_bounds__(kBlk) void pcg_progress_kernel(PgoDev D) { pcg_progress_kernel_body(D); }
__global__ __launch_bounds__(64) void pcg_progress_final_kernel(PgoDev D) { pcg_progress_final_body(D); }
void k_pcg_progress(const PgoDev& D, hipStream_t s)
{
    const int g = progress_blocks(D.nb);
    hipLaunchKernelGGL(pcg_progress_kernel, dim3(g), dim3(kBlk), 0, s, D);
    if (g > 1) hipLaunchKernelGGL(pcg_progress_final_kernel, dim3(1), dim3(64), 0, s, D);
}

// After PCG has set `done`: scal[7] = |r|^2 / |b|^2 with the recurrence residual r (= b - (H + lambda) x up to rounding for ANY
// step lengths and directions, so it is the true residual even when the preconditioner misbehaved).  The host refuses a
// "converged" solve whose residual has not come down (uzl_pgo.hip, pcg_solve).  One workgroup; a no-op until `done`.
__device__ __forceinline__ void residual_guard_kernel_body(PgoDev D)
{
    __shared__ double sr[16], sb[16];
    if (!D.flags[0]) return;
    const int n = D.nb * 6;
    double rr = 0., bb = 0.;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {       // (eight entries in flight per lane, as in lm_tail_kernel)
        double rv[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 1024; rv[u] = i < n ? D.r[i] : 0.; bv[u] = i < n ? D.b[i] : 0.; }
#pragma unroll
        for (int u = 0; u < 8; u++) { rr += rv[u] * rv[u]; bb += bv[u] * bv[u]; }
    }
    for (int o = 32; o; o >>= 1) { rr += __shfl_xor(rr, o); bb += __shfl_xor(bb, o); }
    if ((threadIdx.x & 63) == 0) { sr[threadIdx.x >> 6] = rr; sb[threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        rr = 0.; bb = 0.;
        for (int w = 0; w < 16; w++) { rr += sr[w]; bb += sb[w]; }
        D.scal[7] = bb > 0. ? rr / bb : 0.;
    }
}
__global__ __launch_bounds__(1024) void residual_guard_kernel(PgoDev D)
{
    residual_guard_kernel_body(D);
}
void k_residual_guard(const PgoDev& D, hipStream_t s) { hipLaunchKernelGGL(residual_guard_kernel, dim3(1), dim3(1024), 0, s, D); }

void k_publish(const PgoDev& D, PgoHostScal* out_dev, uint32_t seq, hipStream_t s)
{
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s, D.scal, D.flags, out_dev, seq);
}
void k_set_scalar(double* dst, double v, hipStream_t s)
{
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, dst, v);
}
void k_set_scalar2(double* dst_a, double va, double* dst_b, double vb, hipStream_t s)
{
    hipLaunchKernelGGL(set_scalar2_kernel, dim3(1), dim3(1), 0, s, dst_a, va, dst_b, vb);
}
void k_set_trial(double* scal, double lambda, double tol_f2, double eps_t, double eps_r, hipStream_t s)
{
    hipLaunchKernelGGL(set_trial_kernel, dim3(1), dim3(1), 0, s, scal, lambda, tol_f2, eps_t, eps_r);
}
void k_finalize(const PgoDev& D, int na, int nb_, int nc, int what, hipStream_t s)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlk), 0, s, D, na, nb_, nc, what);
}
void k_precond(const PgoDev& D, hipStream_t s)
{
    hipLaunchKernelGGL(precond_kernel, dim3((D.nb + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D);
}
int k_pcg_init(const PgoDev& D, double* p0, double* p1, hipStream_t s)
{
    const int g = grid_for(D.nb, kBlk / 6, kMaxPartials);
    hipLaunchKernelGGL(pcg_init_kernel, dim3(g), dim3(kBlk), 0, s, D, p0, p1);
    return g;
}
int k_pcg_spmv(const PgoDev& D, const double* p_old, double* p_new, int n_part, double tol2, hipStream_t s)
{
    const int g = grid_for(D.nb, kBlk / 64, kMaxPartials);
    hipLaunchKernelGGL(pcg_spmv_kernel, dim3(g), dim3(kBlk), 0, s, D, p_old, p_new, n_part, tol2);
    return g;
}
int k_pcg_update(const PgoDev& D, const double* p, int n_part, hipStream_t s)
{
    const int g = grid_for(D.nb, kBlk / 6, kMaxPartials);
    hipLaunchKernelGGL(pcg_update_kernel, dim3(g), dim3(kBlk), 0, s, D, p, n_part);
    return g;
}
// grid sizes are pure functions of nb: the host needs them before the first launch (partial counts)
int g_pcg_spmv(int nb) { return grid_for(nb, kBlk / 64, kMaxPartials); }
int g_pcg_update(int nb) { return grid_for(nb, kBlk / 6, kMaxPartials); }
int k_oplus(const PgoDev& D, const double* pose_in, double* pose_out, hipStream_t s)
{
    const int g = grid_for(D.n, kBlk, kMaxPartials);
    hipLaunchKernelGGL(oplus_kernel, dim3(g), dim3(kBlk), 0, s, D, pose_in, pose_out);
    return g;
}
void k_edge_error(const PgoDev& D, const double* pose, double* err, hipStream_t s)
{
    if (D.e > 0) hipLaunchKernelGGL(edge_error_kernel, dim3((D.e + kBlk - 1) / kBlk), dim3(kBlk), 0, s, D, pose, err);
}
void k_poses_out(const double* pose, int n, double* out12, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(poses_out_kernel, dim3((n + kBlk - 1) / kBlk), dim3(kBlk), 0, s, pose, n, out12);
}

}  // namespace uzl
