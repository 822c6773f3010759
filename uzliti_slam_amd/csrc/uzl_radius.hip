// uzl_radius.hip — distance loop-closure candidate producer (kernel + host + C ABI uzl_radius_*).
//
// Mirrors SlamGraph::getNodesWithinRadius (graph_slam_common/src/slam_graph.cpp:266-278) and the filters of its caller
// (graph_slam/src/graph_slam_node.cpp:272-289) for a batch of query nodes.  HBM layout: positions SoA (x[], y[], z[]),
// rotations [n][9], stamps [n].  Kernel: one 256-lane workgroup per query streams the position arrays (coalesced,
// 24 B per node: the bound is HBM/L2 bandwidth, N x Q x 24 B), evaluates the rotation test only for the few hits, and
// appends hits in node order with a ballot + prefix count, so the output order equals the reference's std::map order.
// Two passes (count, then write at the exclusive prefix offsets) keep the jobs of all queries contiguous.
// -ffp-contract=off: the distance / angle tests are bit-identical to the CPU checker.
#include "uzl_common.hpp"
#include "uzl_streams.hpp"

#include <algorithm>
#include <new>

namespace uzl {

constexpr int kRadBlk = 256;

struct RadiusArgs {
    int32_t n, nq;
    const double* x; const double* y; const double* z;
    const double* rot;                 // [n][9]
    const int64_t* stamp;              // [n]
    const int32_t* queries;            // [nq]
    double radius, new_edge_time, max_rot_deg;
    int32_t* count;                    // [nq]
    const int64_t* offset;             // [nq] exclusive prefix (write pass)
    int32_t* out_from; int32_t* out_to;
    int64_t cap;
};

namespace {

__device__ __forceinline__ double rad_angle_of(const double* m)
{
    double q0, q1, q2, q3;
    double t = m[0] + m[4] + m[8];
    if (t > 0.) {
        t = sqrt(t + 1.0);
        q0 = 0.5 * t;
        t = 0.5 / t;
        q1 = (m[7] - m[5]) * t; q2 = (m[2] - m[6]) * t; q3 = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        double qv[3];
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q0 = (m[k * 3 + j] - m[j * 3 + k]) * t;
        qv[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        qv[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q1 = qv[0]; q2 = qv[1]; q3 = qv[2];
    }
    const double n2 = (q1 * q1 + q2 * q2) + q3 * q3;
    if (n2 < 1e-12 * 1e-12) return 0.;
    double w = q0;
    if (w < -1.) w = -1.;
    if (w > 1.) w = 1.;
    return 2. * acos(w);
}

}  // namespace

template <bool WRITE>
__global__ __launch_bounds__(kRadBlk) void radius_kernel(RadiusArgs a)
{
    __shared__ int s_wave[kRadBlk / 64];
    __shared__ int s_base;
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = a.queries[j];
    if (tid == 0) s_base = 0;
    __syncthreads();
    if (q < 0 || q >= a.n) { if (!WRITE && tid == 0) a.count[j] = 0; return; }
    const double qx = a.x[q], qy = a.y[q], qz = a.z[q];
    const int64_t qs = a.stamp[q];
    double Rq[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Rq[k] = a.rot[(size_t)q * 9 + k];
    const int64_t off = WRITE ? a.offset[j] : 0;
    for (int c0 = 0; c0 < a.n; c0 += kRadBlk) {
        const int c = c0 + tid;
        bool hit = false;
        if (c < a.n && c != q) {
            const double dx = a.x[c] - qx, dy = a.y[c] - qy, dz = a.z[c] - qz;
            if (sqrt((dx * dx + dy * dy) + dz * dz) < a.radius) {                           // slam_graph.cpp:272
                const double dts = fabs((double)(qs - a.stamp[c]) * 1e-9);                  // graph_slam_node.cpp:277
                if (dts > a.new_edge_time) {
                    const double* C = a.rot + (size_t)c * 9;
                    double Rd[9];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int k = 0; k < 3; k++) Rd[r * 3 + k] = (C[0 * 3 + r] * Rq[0 * 3 + k] + C[1 * 3 + r] * Rq[1 * 3 + k]) + C[2 * 3 + r] * Rq[2 * 3 + k];
                    const double diff_rotation = 180. * rad_angle_of(Rd) / M_PI;            // :279-280
                    hit = fabs(diff_rotation) < a.max_rot_deg;                              // :282
                }
            }
        }
        // ordered append: hits of this 256-node slab in node order
        const unsigned long long m = __ballot(hit);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int wave_off = 0, slab = 0;
#pragma unroll
        for (int w = 0; w < kRadBlk / 64; w++) { if (w < wv) wave_off += s_wave[w]; slab += s_wave[w]; }
        const int base = s_base;
        if (WRITE && hit) {
            const int64_t pos = off + base + wave_off + before;
            if (pos < a.cap) { a.out_from[pos] = c; a.out_to[pos] = q; }                    // estimateEdge(close_node, current_node)
        }
        __syncthreads();
        if (tid == 0) s_base = base + slab;
        __syncthreads();
    }
    if (!WRITE && tid == 0) a.count[j] = s_base;
}

}  // namespace uzl

using namespace uzl;

struct uzl_radius {
    std::mutex mu;
    std::string last_error;
    uzl_radius_cfg cfg;
    hipStream_t stream = nullptr;
    int32_t n = 0;
    DevBuf<double> d_x, d_y, d_z, d_rot;
    DevBuf<int64_t> d_stamp, d_off;
    DevBuf<int32_t> d_q, d_count, d_from, d_to;
    PinBuf<int32_t> h_count, h_from, h_to;
    PinBuf<int64_t> h_off;
};

namespace {
int fail(uzl_radius* h, int code, const char* msg) { h->last_error = msg; return code; }
}

#define UZL_GUARD_BEGIN(h)                       \
    if (!(h)) return UZL_ERR_BAD_ARG;            \
    std::lock_guard<std::mutex> lock_((h)->mu);  \
    try {
#define UZL_GUARD_END(h)                                                             \
    } catch (const ::uzl::HipError& e) { return ::uzl::report((h)->last_error, e); } \
    catch (const std::bad_alloc&) { (h)->last_error = "host out of memory"; return UZL_ERR_OOM; } \
    catch (...) { (h)->last_error = "unexpected exception"; return UZL_ERR_HIP; }

extern "C" {

void uzl_radius_cfg_default(uzl_radius_cfg* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->radius = 0.5; c->new_edge_time = 5.0; c->max_rotation_deg = 30.0; c->device = 0;
}

int uzl_radius_create(const uzl_radius_cfg* cfg, uzl_radius** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_radius_cfg c;
    if (cfg) c = *cfg; else uzl_radius_cfg_default(&c);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return UZL_ERR_NO_DEVICE;     // no CPU fallback
    if (c.device < 0 || c.device >= count) return UZL_ERR_NO_DEVICE;
    uzl_radius* h = new (std::nothrow) uzl_radius();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    if (hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return UZL_ERR_HIP; }
    stream_register(c.device, h->stream, false);
    *out = h;
    return UZL_OK;
}

void uzl_radius_destroy(uzl_radius* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_unregister(h->cfg.device, h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

const char* uzl_radius_last_error(uzl_radius* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_radius_set_nodes(uzl_radius* h, int32_t n, const double* poses, const int64_t* stamps)
{
    UZL_GUARD_BEGIN(h)
    if (n < 0 || (n > 0 && (!poses || !stamps))) return fail(h, UZL_ERR_BAD_ARG, "null arrays");
    UZL_HIP(hipSetDevice(h->cfg.device));
    std::vector<double> x((size_t)std::max(n, 1)), y(x.size()), z(x.size()), rot(x.size() * 9);
    for (int32_t i = 0; i < n; i++) {
        const double* T = poses + 12 * (size_t)i;
        x[i] = T[3]; y[i] = T[7]; z[i] = T[11];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rot[(size_t)i * 9 + r * 3 + c] = T[r * 4 + c];
    }
    h->d_x.reserve(x.size()); h->d_y.reserve(x.size()); h->d_z.reserve(x.size()); h->d_rot.reserve(rot.size()); h->d_stamp.reserve(x.size());
    hipStream_t s = h->stream;
    if (n) {
        UZL_HIP(hipMemcpyAsync(h->d_x.p, x.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_y.p, y.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_z.p, z.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_rot.p, rot.data(), (size_t)n * 72, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_stamp.p, stamps, (size_t)n * 8, hipMemcpyHostToDevice, s));
    }
    UZL_HIP(hipStreamSynchronize(s));
    h->n = n;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_radius_query(uzl_radius* h, int32_t nq, const int32_t* queries, int64_t cap, int32_t* out_from, int32_t* out_to,
                     int32_t* count_per_query, int64_t* n_jobs)
{
    UZL_GUARD_BEGIN(h)
    if (nq < 0 || cap < 0 || !n_jobs || (nq > 0 && !queries) || (cap > 0 && (!out_from || !out_to))) return fail(h, UZL_ERR_BAD_ARG, "bad arguments");
    *n_jobs = 0;
    if (nq == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;
    h->d_q.reserve((size_t)nq); h->d_count.reserve((size_t)nq); h->d_off.reserve((size_t)nq);
    h->h_count.reserve((size_t)nq); h->h_off.reserve((size_t)nq);
    UZL_HIP(hipMemcpyAsync(h->d_q.p, queries, (size_t)nq * 4, hipMemcpyHostToDevice, s));
    RadiusArgs a;
    memset(&a, 0, sizeof(a));
    a.n = h->n; a.nq = nq; a.x = h->d_x.p; a.y = h->d_y.p; a.z = h->d_z.p; a.rot = h->d_rot.p; a.stamp = h->d_stamp.p;
    a.queries = h->d_q.p; a.radius = h->cfg.radius; a.new_edge_time = h->cfg.new_edge_time; a.max_rot_deg = h->cfg.max_rotation_deg;
    a.count = h->d_count.p; a.cap = cap;
    hipLaunchKernelGGL(radius_kernel<false>, dim3(nq), dim3(kRadBlk), 0, s, a);
    UZL_HIP(hipMemcpyAsync(h->h_count.p, h->d_count.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipStreamSynchronize(s));
    int64_t total = 0;
    for (int32_t j = 0; j < nq; j++) { h->h_off.p[j] = total; total += h->h_count.p[j]; if (count_per_query) count_per_query[j] = h->h_count.p[j]; }
    *n_jobs = total;
    const int64_t w = std::min(total, cap);
    if (w > 0) {
        h->d_from.reserve((size_t)w); h->d_to.reserve((size_t)w); h->h_from.reserve((size_t)w); h->h_to.reserve((size_t)w);
        UZL_HIP(hipMemcpyAsync(h->d_off.p, h->h_off.p, (size_t)nq * 8, hipMemcpyHostToDevice, s));
        a.offset = h->d_off.p; a.out_from = h->d_from.p; a.out_to = h->d_to.p; a.cap = w;
        hipLaunchKernelGGL(radius_kernel<true>, dim3(nq), dim3(kRadBlk), 0, s, a);
        UZL_HIP(hipGetLastError());
        UZL_HIP(hipMemcpyAsync(h->h_from.p, h->d_from.p, (size_t)w * 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_to.p, h->d_to.p, (size_t)w * 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipStreamSynchronize(s));
        memcpy(out_from, h->h_from.p, (size_t)w * 4);
        memcpy(out_to, h->h_to.p, (size_t)w * 4);
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

}  // extern "C"
