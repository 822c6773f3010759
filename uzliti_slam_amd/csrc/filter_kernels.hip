// filter_kernels.hip — device side of the edge filter (TransformationFilter::calcValidEdges,
// transformation_estimation/src/transformation_filter.cpp:249-276) for gfx950.
//
// Per evaluated cluster edge the reference composes five Isometry3d (world-frame "from" end pose through the
// measured transform) and two ("to" end pose) and keeps the translations as the RANSAC point pair (:253-262).
// One lane per edge; 7 x 96 B in, 48 B out; fixed operation order and -ffp-contract=off so that the points - and
// with them every RANSAC vote - equal the CPU checker's bit for bit.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "filter_types.hpp"

namespace uzl {

namespace {

struct Iso { double m[12]; };

__device__ __forceinline__ Iso iso_load(const double* __restrict__ p)
{
    Iso r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.m[i] = p[i];
    return r;
}

__device__ __forceinline__ Iso iso_mul(const Iso& A, const Iso& B)
{
    Iso o;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++)
            o.m[r * 4 + c] = (A.m[r * 4 + 0] * B.m[0 * 4 + c] + A.m[r * 4 + 1] * B.m[1 * 4 + c]) + A.m[r * 4 + 2] * B.m[2 * 4 + c];
        o.m[r * 4 + 3] = ((A.m[r * 4 + 0] * B.m[3] + A.m[r * 4 + 1] * B.m[7]) + A.m[r * 4 + 2] * B.m[11]) + A.m[r * 4 + 3];
    }
    return o;
}

__device__ __forceinline__ Iso iso_inv(const Iso& A)
{
    Iso o;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) o.m[r * 4 + c] = A.m[c * 4 + r];
        o.m[r * 4 + 3] = -((A.m[0 * 4 + r] * A.m[3] + A.m[1 * 4 + r] * A.m[7]) + A.m[2 * 4 + r] * A.m[11]);
    }
    return o;
}

}  // namespace

// P.col(k) = (pos_from * displacement_from * S[sensor_from] * transform * S[sensor_to]^-1).translation()
// Q.col(k) = (pos_to * displacement_to).translation()                                   (:253-262)
__global__ __launch_bounds__(kFilterBlock) void filter_points_kernel(const FilterEdgeDev* __restrict__ edges, int n,
                                                                     const double* __restrict__ sensors, int n_sensors,
                                                                     double* __restrict__ P, double* __restrict__ Q)
{
    const int k = blockIdx.x * kFilterBlock + threadIdx.x;
    if (k >= n) return;
    const FilterEdgeDev* e = edges + k;
    Iso ident;
#pragma unroll
    for (int i = 0; i < 12; i++) ident.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    const Iso Sf = (e->sensor_from >= 0 && e->sensor_from < n_sensors) ? iso_load(sensors + 12 * (size_t)e->sensor_from) : ident;
    const Iso St = (e->sensor_to >= 0 && e->sensor_to < n_sensors) ? iso_load(sensors + 12 * (size_t)e->sensor_to) : ident;
    Iso a = iso_mul(iso_load(e->pos_from), iso_load(e->disp_from));
    a = iso_mul(a, Sf);
    a = iso_mul(a, iso_load(e->transform));
    a = iso_mul(a, iso_inv(St));
    P[3 * (size_t)k + 0] = a.m[3]; P[3 * (size_t)k + 1] = a.m[7]; P[3 * (size_t)k + 2] = a.m[11];
    const Iso b = iso_mul(iso_load(e->pos_to), iso_load(e->disp_to));
    Q[3 * (size_t)k + 0] = b.m[3]; Q[3 * (size_t)k + 1] = b.m[7]; Q[3 * (size_t)k + 2] = b.m[11];
}

// consensus3D(P, Q, T, max_error, set) with the transform estimateSVD returned (:275-276).  It equals the RANSAC
// kernel's own final mask when the estimate succeeded; when it failed (best < 3) T is the identity and the
// reference still counts the pairs closer than max_error - reproduced here.  One lane per column.
__global__ __launch_bounds__(kFilterBlock) void filter_consensus_kernel(const double* __restrict__ P, const double* __restrict__ Q,
                                                                        const int32_t* __restrict__ col_cluster, int n,
                                                                        const uzl_edge_result* __restrict__ results,
                                                                        double max_error, uint8_t* __restrict__ set)
{
    const int k = blockIdx.x * kFilterBlock + threadIdx.x;
    if (k >= n) return;
    const double* T = results[col_cluster[k]].T;
    const double px = P[3 * (size_t)k], py = P[3 * (size_t)k + 1], pz = P[3 * (size_t)k + 2];
    const double x = fma(T[0], px, fma(T[1], py, fma(T[2], pz, T[3])));              // same fused recipe as the estimator's votes
    const double y = fma(T[4], px, fma(T[5], py, fma(T[6], pz, T[7])));
    const double z = fma(T[8], px, fma(T[9], py, fma(T[10], pz, T[11])));
    const double dx = x - Q[3 * (size_t)k], dy = y - Q[3 * (size_t)k + 1], dz = z - Q[3 * (size_t)k + 2];
    set[k] = sqrt(fma(dx, dx, fma(dy, dy, dz * dz))) < max_error ? 1 : 0;
}

void launch_filter_points(const FilterEdgeDev* edges, int n, const double* sensors, int n_sensors, double* P, double* Q, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(filter_points_kernel, dim3((n + kFilterBlock - 1) / kFilterBlock), dim3(kFilterBlock), 0, s, edges, n, sensors, n_sensors, P, Q);
}

void launch_filter_consensus(const double* P, const double* Q, const int32_t* col_cluster, int n, const uzl_edge_result* results,
                             double max_error, uint8_t* set, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(filter_consensus_kernel, dim3((n + kFilterBlock - 1) / kFilterBlock), dim3(kFilterBlock), 0, s, P, Q, col_cluster, n, results, max_error, set);
}

}  // namespace uzl
