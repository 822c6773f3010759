// pgo_device.hpp — device-side helpers shared by the pose-graph kernel files (small fixed-size algebra,
// deterministic reductions).  Everything is __forceinline__ and register resident.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pgo_types.hpp"

namespace uzl {

// Hand-over of a snapshot to the host through pinned, coherent memory (lm_tail_kernel, publish_kernel): the fields go out as relaxed
// system-scope stores, every lane waits until ITS stores have been acknowledged, a workgroup barrier, then one lane stores the sequence
// word.  On gfx9-family targets (gfx90a / gfx942 / gfx950) vmcnt counts stores as well as loads, so `s_waitcnt vmcnt(0)` is that wait -
// without the system-scope release fence, which also writes back every dirty line of the L2 (the solve's whole working set: 11.0 ->
// 7.4 us per tail, and the host reads none of it).  This is outside the HIP memory model and only right where vmcnt counts stores
// (gfx10+ tracks them with vscnt): any other target takes the fence, and the caller's sequence store is then a release.
#if defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__)
#define UZL_PUBLISH_SEQ_ORDER __ATOMIC_RELAXED
__device__ __forceinline__ void publish_wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
#define UZL_PUBLISH_SEQ_ORDER __ATOMIC_RELEASE
__device__ __forceinline__ void publish_wait_own_stores() { __threadfence_system(); }
#endif



constexpr int kBlk = 256;

// ------------------------------------------------------------------------------------------------
// small fixed-size algebra (everything stays in registers; indices are compile-time constants)
// ------------------------------------------------------------------------------------------------
struct Q4 { double w, x, y, z; };
struct V3 { double x, y, z; };
struct M33 { double m[9]; };

__device__ __forceinline__ Q4 qmul(const Q4& a, const Q4& b)
{
    return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qconj(const Q4& a) { return Q4{a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ Q4 qnormalize(const Q4& a)
{
    const double n = 1.0 / sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
    return Q4{a.w * n, a.x * n, a.y * n, a.z * n};
}
// Eigen::Quaterniond::toRotationMatrix [EXT]
__device__ __forceinline__ M33 qrot(const Q4& q)
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M33 R;
    R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
    R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
    R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
    return R;
}
__device__ __forceinline__ V3 mulv(const M33& R, const V3& v)
{
    return V3{R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z,
              R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
              R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 mulTv(const M33& R, const V3& v)
{
    return V3{R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z,
              R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
              R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
// Eigen::Quaterniond(Matrix3d) [EXT]; m row-major
__device__ __forceinline__ Q4 quat_from_R(const double* m)
{
    Q4 q;
    double t = m[0] + m[4] + m[8];
    if (t > 0.) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {            // i = 0
        t = sqrt(m[0] - m[4] - m[8] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (m[7] - m[5]) * t; q.y = (m[3] + m[1]) * t; q.z = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {             // i = 1
        t = sqrt(m[4] - m[8] - m[0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (m[2] - m[6]) * t; q.z = (m[7] + m[5]) * t; q.x = (m[1] + m[3]) * t;
    } else {                                               // i = 2
        t = sqrt(m[8] - m[0] - m[4] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (m[3] - m[1]) * t; q.x = (m[2] + m[6]) * t; q.y = (m[5] + m[7]) * t;
    }
    return q;
}

struct Pose { V3 t; Q4 q; };
__device__ __forceinline__ Pose load_pose(const double* __restrict__ p, int v)
{
    const double2* q = reinterpret_cast<const double2*>(p + (size_t)v * 8);
    const double2 a = q[0], b = q[1], c = q[2], d = q[3];
    return Pose{V3{a.x, a.y, b.x}, Q4{b.y, c.x, c.y, d.x}};
}
__device__ __forceinline__ void store_pose(double* __restrict__ p, int v, const Pose& P)
{
    double2* q = reinterpret_cast<double2*>(p + (size_t)v * 8);
    q[0] = make_double2(P.t.x, P.t.y); q[1] = make_double2(P.t.z, P.q.w);
    q[2] = make_double2(P.q.x, P.q.y); q[3] = make_double2(P.q.z, 0.);
}
// 3x4 row-major [R|t] -> Pose (unit quaternion)
__device__ __forceinline__ Pose pose_from_T(const double* T)
{
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    return Pose{V3{T[3], T[7], T[11]}, qnormalize(quat_from_R(R))};
}
__device__ __forceinline__ void T_from_pose(const Pose& P, double* T)
{
    const M33 R = qrot(P.q);
    T[0] = R.m[0]; T[1] = R.m[1]; T[2] = R.m[2]; T[3] = P.t.x;
    T[4] = R.m[3]; T[5] = R.m[4]; T[6] = R.m[5]; T[7] = P.t.y;
    T[8] = R.m[6]; T[9] = R.m[7]; T[10] = R.m[8]; T[11] = P.t.z;
}
__device__ __forceinline__ Pose pose_mul(const Pose& A, const Pose& B)
{
    const V3 rb = mulv(qrot(A.q), B.t);
    return Pose{V3{rb.x + A.t.x, rb.y + A.t.y, rb.z + A.t.z}, qnormalize(qmul(A.q, B.q))};
}
__device__ __forceinline__ Pose pose_inv(const Pose& A)
{
    const V3 t = mulTv(qrot(A.q), A.t);
    return Pose{V3{-t.x, -t.y, -t.z}, qconj(A.q)};
}
// optimize_xy_only: zero roll, pitch, z through toEuler/fromEuler
// (g2o_optimizer.cpp:164-170, isometry3d_mappings.cpp:47-75)
__device__ __forceinline__ Pose project_xy(const Pose& P)
{
    // toEuler takes Quaterniond(R) un-normalised; P.q is the normalised quaternion of the same R
    const double q0 = P.q.w, q1 = P.q.x, q2 = P.q.y, q3 = P.q.z;
    const double yaw = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
    const double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5);
    return Pose{V3{P.t.x, P.t.y, 0.}, Q4{cy, 0., 0., sy}};
}

// use_odometry_parameters (g2o_optimizer.cpp:209-227): the odometry measurement's (x, y, yaw) goes through g2o's
// sclam2d OdomConvert [EXT] - motion -> differential-drive wheel velocities (wheel base 1) -> motion - with roll,
// pitch and z kept (toEuler / fromEuler of isometry3d_mappings.cpp:47-75).  Identity on exact circular arcs.
__device__ __forceinline__ Pose odom_round_trip(const Pose& P, double dt)
{
    const double q0 = P.q.w, q1 = P.q.x, q2 = P.q.y, q3 = P.q.z;
    const double roll = atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
    const double pitch = asin(2 * (q0 * q2 - q3 * q1));
    const double theta = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
    const double x = P.t.x, y = P.t.y;
    double vl, vr;
    if (fabs(theta) > 1e-7) {
        const double c = cos(theta), s = sin(theta);
        const double y2 = 10.;
        const double x4 = (c * 0. - s * y2) + x, y4 = (s * 0. + c * y2) + y;
        const double R = (y2 * (x * y4 - y * x4)) / (y2 * (x - x4));
        const double w = (fabs(dt) > 1e-7) ? theta / dt : 0.;
        vl = (2. * R * w - w) / 2.;
        vr = w + vl;
    } else {
        vl = vr = (fabs(dt) > 1e-7) ? hypot(x, y) / dt : 0.;
    }
    double nx, ny, nth;
    if (fabs(vr - vl) > 1e-7) {
        const double R = 0.5 * ((vl + vr) / (vr - vl));
        const double w = vr - vl;
        nth = w * dt;
        const double c = cos(nth), s = sin(nth);
        nx = (c * 0. - s * (-R)) + 0.;
        ny = (s * 0. + c * (-R)) + R;
    } else {
        nx = 0.5 * (vr + vl) * dt; ny = 0.; nth = 0.;
    }
    const double sy = sin(nth * 0.5), cy = cos(nth * 0.5);
    const double sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
    const double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
    return Pose{V3{nx, ny, P.t.z}, Q4{cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                                      cr * cp * sy - sr * sp * cy}};
}

// ------------------------------------------------------------------------------------------------
// reductions (deterministic: fixed tree shapes, no atomics)
// ------------------------------------------------------------------------------------------------
// Wave-wide sum / maximum, the same value in every lane.  Two implementations of ONE summation tree (lanes pairwise, quads, halves of
// a row, rows, then rows 0+1 and 2+3, then their sum - the xor butterfly with strides 1, 2, 4, 8, 16, 32), bit for bit the same:
//   LDS = false: DPP moves (quad swaps, row mirrors, row broadcasts) and a readlane of lane 63 - no LDS crossbar: six short dependent
//                steps instead of six ds_bpermute round trips.  For the latency-bound single-graph kernels (ten sums per launch).
//   LDS = true : the ds_bpermute butterfly.  For the batched kernels, whose many waves per SIMD are bound by vector-ALU issue: there the
//                crossbar is the free resource (16 batched config-2 graphs: 54.7 M edges/s against 50.2 M with the DPP form).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double old, double v)          // lanes the move does not write keep `old`
{
    const long long b = __double_as_longlong(v), o = __double_as_longlong(old);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double lane63_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Prefixes of the same butterfly for sums over aligned groups of 4 / 8 / 16 lanes (every lane of the group gets the sum): v + v[lane ^ 1],
// then + [lane ^ 2], + [lane ^ 4], + [lane ^ 8] - the last two through row mirrors, which reach the right partner because the values are
// uniform over 4 resp. 8 lanes by then.  Same bits as the __shfl_xor chain they replace; only valid in this order, starting at stride 1.
__device__ __forceinline__ double xsum4(double v) { v += dpp_mov_f64<0xb1, 0xf>(0., v); v += dpp_mov_f64<0x4e, 0xf>(0., v); return v; }
__device__ __forceinline__ double xsum8(double v) { v = xsum4(v); v += dpp_mov_f64<0x141, 0xf>(0., v); return v; }
__device__ __forceinline__ double xsum16(double v) { v = xsum8(v); v += dpp_mov_f64<0x140, 0xf>(0., v); return v; }
template <bool LDS = false>
__device__ __forceinline__ double wave_sum(double v)
{
    if (LDS) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    }
    v += dpp_mov_f64<0xb1, 0xf>(0., v);          // quad_perm [1,0,3,2]     lane ^ 1
    v += dpp_mov_f64<0x4e, 0xf>(0., v);          // quad_perm [2,3,0,1]     lane ^ 2
    v += dpp_mov_f64<0x141, 0xf>(0., v);         // row_half_mirror         the other quad of the half row (values are quad-uniform)
    v += dpp_mov_f64<0x140, 0xf>(0., v);         // row_mirror              the other half of the row
    v += dpp_mov_f64<0x142, 0xa>(0., v);         // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f64<0x143, 0xc>(0., v);         // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return lane63_f64(v);
}
template <bool LDS = false>
__device__ __forceinline__ double wave_max(double v)
{
    if (LDS) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v = fmax(v, __shfl_xor(v, o));
        return v;
    }
    v = fmax(v, dpp_mov_f64<0xb1, 0xf>(v, v));
    v = fmax(v, dpp_mov_f64<0x4e, 0xf>(v, v));
    v = fmax(v, dpp_mov_f64<0x141, 0xf>(v, v));
    v = fmax(v, dpp_mov_f64<0x140, 0xf>(v, v));
    v = fmax(v, dpp_mov_f64<0x142, 0xa>(v, v));
    v = fmax(v, dpp_mov_f64<0x143, 0xc>(v, v));
    return lane63_f64(v);
}
// all threads get the block total (blockDim = 256)
__device__ __forceinline__ double block_sum(double v, double* s4)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}
__device__ __forceinline__ double block_max(double v, double* s4)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
}
// every block re-reduces the (<= 1024) partials of the previous kernel: same order everywhere
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int count, double* s4)
{
    double v = 0.;
    for (int i = threadIdx.x; i < count; i += kBlk) v += part[i];
    return block_sum(v, s4);
}

// inverse of a symmetric positive definite 6x6 (row-major) through its Cholesky factor
__device__ __forceinline__ void spd_inverse6(double* A, double* out)
{
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        d = sqrt(fmax(d, 1e-300));
        A[j * 6 + j] = d;
        const double inv = 1. / d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * inv;
        }
    }
    double Li[36];
#pragma unroll
    for (int i = 0; i < 36; i++) Li[i] = 0.;
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int r = c; r < 6; r++) {
            double s = (r == c) ? 1. : 0.;
#pragma unroll
            for (int k = c; k < r; k++) s -= A[r * 6 + k] * Li[k * 6 + c];
            Li[r * 6 + c] = s / A[r * 6 + r];
        }
    }
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.;
#pragma unroll
            for (int k = (r > c ? r : c); k < 6; k++) s += Li[k * 6 + r] * Li[k * 6 + c];
            out[r * 6 + c] = s;
        }
}

// 1/sqrt(x) from the hardware estimate + two Newton steps (deterministic; accuracy ~1e-15 is ample for a smoother)
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
// inverse of an SPD 6x6 (row-major) through its Cholesky factor, reciprocal square roots only
__device__ __forceinline__ void spd_inverse6_rs(double* A, double* out)
{
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        inv[j] = rsqrt_nr(fmax(d, 1e-300));
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * inv[j];
        }
    }
    double Li[36];
#pragma unroll
    for (int i = 0; i < 36; i++) Li[i] = 0.;
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int r = c; r < 6; r++) {
            double s = (r == c) ? 1. : 0.;
#pragma unroll
            for (int k = c; k < r; k++) s -= A[r * 6 + k] * Li[k * 6 + c];
            Li[r * 6 + c] = s * inv[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.;
#pragma unroll
            for (int k = (r > c ? r : c); k < 6; k++) s += Li[k * 6 + r] * Li[k * 6 + c];
            out[r * 6 + c] = s;
        }
}

// ------------------------------------------------------------------------------------------------
// When to stop the PCG: an a-posteriori estimate of the error left in the LM step, in the units of the parity bar.
// Every kProgressEvery iterations the solve looks at how far x moved since its last look,
//     s = x_k - x_{k-d},   e_k = x* - x_k = sum_{j >= k} alpha_j p_j,   so   e_{k-d} = s + e_k ;
// with the error contracting by q per window (q^2 = ratio of r.M^-1 r over the window, the energy norm of the error when M ~ A),
// |e_k| ~ q / (1 - q) |s|.  The solve stops when kProgressSafety times that estimate - largest translation component [m] and largest
// rotation (quaternion vector, ~ half-angle) component over all vertices - is below what the host asks for (scal[12], scal[13]:
// a fraction of BASELINE's 1e-3 m / 1e-4 rad spread over the LM iterations).  A step of 1e-7 m is accepted after the first look; a
// step of metres is iterated until 1e-6 of it is settled.  The relative test on r.M^-1 r stays as a floor.
// Multilevel path: the look is part of the iteration - ml_cg leaves, per workgroup, the largest movement of its rows (in units of the
// accuracy asked for: max(|dx_t| / eps_t, |dx_r| / eps_r)) and |r|^2 of its rows in part_c on the iterations ml_spmv flags (flags[3]);
// block 0 of the next ml_spmv folds them and decides (progress_decide_ml): no launch of its own, so the look can be taken every 4
// iterations.  The solve also waits for the residual to come down (|r|^2 <= kProgressResidual |b|^2, |b|^2 in scal[14]: folded by the
// first ml_spmv from what the first ml_cg left): that is the bar residual_guard_kernel holds a finished solve to, and a converged LM
// iteration - whose whole step is below the accuracy asked for - would otherwise stop at the first look with |r| barely reduced.
// Block-Jacobi path (graphs too small or too large for a hierarchy): pcg_progress_kernel, between the iterations.
// mt / mr: largest |x - xs| over the translation / rotation components; rz: r.M^-1 r of the iteration that ended at the look.
// ------------------------------------------------------------------------------------------------
constexpr double kProgressSafety = 2., kProgressQMax = 0.95;
// The extrapolation |e_k| ~ q / (1 - q) |s| holds while CG contracts.  CG is not monotone: on a stiff system x can sit almost still for a
// window while far from the solution (a plateau: the Ritz values have not found the soft modes yet), r.M^-1 r barely moves, q ~ 1 - and a
// tiny |s| would pass for convergence.  A window that contracted by less than this is not trusted: the solve goes on until contraction
// resumes (or the relative floor ends it).  With the multilevel operator q is 0.2 - 0.7 on the BASELINE graphs: no look is lost there.
constexpr double kProgressQTrust = 0.9;
// (below residual_guard_kernel's 0.25 - kResidualGuard, uzl_pgo.hip - by more than the two kernels' different summation orders can move
//  the ratio: a solve the look lets go at 0.2499 must not read 0.2501 there)
constexpr double kProgressResidual = 0.2;
__device__ __forceinline__ void progress_decide(PgoDev D, double mt, double mr, double rz)
{
    const double rz_prev = D.scal[11];
    double q = (rz_prev > 0. && rz >= 0.) ? sqrt(rz / rz_prev) : kProgressQMax;
    const bool trusted = q <= kProgressQTrust;
    q = fmin(q, kProgressQMax);
    const double gain = kProgressSafety * q / (1. - q);
    const double et = gain * mt, er = gain * mr;
    D.scal[11] = rz; D.scal[14] = et; D.scal[15] = er;
    if (trusted && et <= D.scal[12] && er <= D.scal[13] && rz >= 0.) D.flags[0] = 1;
}
// m: largest movement in units of the accuracy asked for; rr: |r|^2
__device__ __forceinline__ void progress_decide_ml(PgoDev D, double m, double rr, double rz)
{
    const double rz_prev = D.scal[11], m_prev = D.scal[15];
    double q = (rz_prev > 0. && rz >= 0.) ? sqrt(rz / rz_prev) : kProgressQMax;
    // ... and the contraction of the ITERATE itself, from the second look on: the error is the tail of the series of movements, and the
    // soft modes of a large chain-like system - where the pose error lives - contract more slowly than the energy norm of the residual
    // says (config 5: 2.4e-4 m off a tightly solved run at some intervals with the residual's q alone; tests/diag/c5_tolerance.py)
    if (m_prev > 0.) q = fmax(q, m / m_prev);
    const bool trusted = q <= kProgressQTrust;
    q = fmin(q, kProgressQMax);
    const double est = (kProgressSafety * q / (1. - q)) * m;
    D.scal[11] = rz; D.scal[15] = m;
    if (trusted && est <= 1. && rr <= kProgressResidual * D.scal[14] && rz >= 0.) D.flags[0] = 1;
}
// 1 / (accuracy asked for), for the movement of component `comp` of a row (0..2 translation [m], 3..5 rotation [q_xyz])
__device__ __forceinline__ double progress_unit(const double* __restrict__ scal, int comp)
{
    const double eps = scal[comp < 3 ? 12 : 13];
    return eps > 0. ? 1. / eps : 1e300;
}

}  // namespace uzl
