// uzl_match.hip — host side of the edge-estimation half: frame store in HBM, batched job launch,
// C ABI (uzl_match_*, uzl_ransac_points).  Mirrors FeatureTransformationEstimator
// (transformation_estimation/src/feature_transformation_estimator.cpp) as a batched, device-resident
// service: frames are uploaded once, every node-pair job references them by id.
#include "uzl_common.hpp"
#include "uzl_streams.hpp"
#include <thread>
#include "match_types.hpp"
#include "match_internal.hpp"
#include "wire_types.hpp"

#include <algorithm>
#include <cmath>
#include <new>

namespace uzl {

void launch_knn2(const uint32_t* arena, const Combo* combos, int n_combos, int max_nq, uint2* knn,
                 bool has8, bool has16, bool has_generic, hipStream_t s);
size_t estimate_lds_bytes(int sort_cap, int iterations, int lds_points, bool in_lds);
hipError_t launch_estimate(const EstimateArgs& a, int n_jobs, bool in_lds, size_t lds_bytes, hipStream_t s);

constexpr size_t kLdsBudget = 152 * 1024;   // of the CU's 160 KiB

struct FrameRec {
    bool alive = false;
    uint64_t desc_off = 0, pos_off = 0, valid_off = 0;   // byte offsets into the arena
    uint64_t ext_off = 0, ext_size = 0;                   // the arena extent the frame occupies (handed back by remove_frame)
    int32_t n = 0, words = 0, feature_type = 0, sensor_frame = 0;
};
struct Extent { size_t off, size; };

}  // namespace uzl

using namespace uzl;

struct uzl_match {
    std::mutex mu;
    std::string last_error;
    uzl_match_cfg cfg;
    hipStream_t stream = nullptr;
    // frame arena: one HBM allocation, frames addressed by offset so it can grow
    DevBuf<uint8_t> arena;
    size_t arena_used = 0;               // high-water mark: [0, arena_used) is handed out or on the free list
    std::vector<Extent> free_list;       // freed extents, sorted by offset, neighbours merged (first fit)
    std::vector<Extent> deferred_free;   // frames removed while a batch that may read them is in flight: freed by its collect
    size_t live_bytes = 0;
    std::vector<FrameRec> frames;
    int32_t live_frames = 0;
    // batch state
    PinBuf<Combo> h_combos; DevBuf<Combo> d_combos;
    PinBuf<Job> h_jobs; DevBuf<Job> d_jobs;
    DevBuf<uint2> d_knn;
    DevBuf<uzl_edge_result> d_results; PinBuf<uzl_edge_result> h_results;
    DevBuf<int32_t> d_cq, d_ct, d_cd; DevBuf<uint8_t> d_mask;
    PinBuf<int32_t> h_cq, h_ct, h_cd; PinBuf<uint8_t> h_mask;
    DevBuf<double> d_pq_scratch, d_dist_scratch; DevBuf<uint8_t> d_mask_scratch;
    DevBuf<double> d_P, d_Q;
    // wire batches (uzl_match_add_frames_wire / frame_to_wire): raw Feature records, segment table, u/v, error flag
    DevBuf<uint32_t> d_wire_stage; DevBuf<WireSeg> d_wire_segs; DevBuf<int32_t> d_wire_uv; DevBuf<int32_t> d_wire_bad;
    // uzl_match_add_frame staging: the caller's three arrays are packed into pinned memory in the frame's arena layout and go up as ONE
    // asynchronous copy; the call returns without waiting (the inputs are no longer needed once packed).  Two halves: a half is reused
    // only after the copies issued from it have completed (event).
    PinBuf<uint8_t> h_up;
    // uzl_match_add_frames staging: two larger halves, packed by several host threads while the other half's copy is in flight
    PinBuf<uint8_t> h_bulk;
    hipEvent_t bulk_ev[2] = {nullptr, nullptr};
    bool bulk_pending[2] = {false, false};
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    bool up_pending[2] = {false, false};
    int up_half = 0;
    size_t up_used = 0;
    bool in_flight = false;
    int32_t fl_jobs = 0, fl_stride = 0, fl_max_corr = 0;
    bool fl_diag = false;
    KernelTimer timer;
};

namespace {

int fail(uzl_match* h, int code, const char* msg)
{
    h->last_error = msg;
    return code;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kUpHalf = 8u << 20;          // bytes per half of the add_frame staging buffer
constexpr size_t kBulkHalf = 32u << 20;       // bytes per half of the add_frames staging buffer (one DMA per half)

// ---- frame arena: first-fit free list over one growing HBM allocation.  The reference removes and merges nodes all the time
// (graph_slam_node.cpp:665-777); a store that only reclaims space when it is empty leaks HBM in a long-running node.
// grows the arena to hold arena_used; if that throws (out of memory) the extents just handed out go back - the caller's frames were not
// registered, so nothing refers to them
template <class Undo>
void arena_grow(uzl_match* h, Undo&& undo)
{
    try { h->arena.reserve(h->arena_used, /*keep=*/true, h->stream); }
    catch (...) { undo(); throw; }
}

size_t arena_alloc(uzl_match* h, size_t size)
{
    size = align_up(size, 256);
    for (size_t i = 0; i < h->free_list.size(); i++) {
        Extent& e = h->free_list[i];
        if (e.size >= size) {
            const size_t off = e.off;
            e.off += size; e.size -= size;
            if (e.size == 0) h->free_list.erase(h->free_list.begin() + (std::ptrdiff_t)i);
            return off;
        }
    }
    const size_t off = h->arena_used;
    h->arena_used += size;
    return off;
}
void arena_free_now(uzl_match* h, size_t off, size_t size)
{
    if (size == 0) return;
    size = align_up(size, 256);
    auto it = std::lower_bound(h->free_list.begin(), h->free_list.end(), off, [](const Extent& e, size_t o) { return e.off < o; });
    it = h->free_list.insert(it, Extent{off, size});
    if (it + 1 != h->free_list.end() && it->off + it->size == (it + 1)->off) { it->size += (it + 1)->size; h->free_list.erase(it + 1); }
    if (it != h->free_list.begin() && (it - 1)->off + (it - 1)->size == it->off) { (it - 1)->size += it->size; it = h->free_list.erase(it) - 1; }
    if (it->off + it->size == h->arena_used) { h->arena_used = it->off; h->free_list.erase(it); }      // the top of the arena comes down
}
void arena_free(uzl_match* h, size_t off, size_t size)
{
    if (h->in_flight) h->deferred_free.push_back(Extent{off, size});     // the batch in flight may still read the frame
    else arena_free_now(h, off, size);
}
// [desc | pad | pos | pad | valid | tail] of one frame, relative to its extent; returns the extent's size
size_t frame_layout(size_t n, size_t bytes_per_desc, FrameRec& r, size_t base)
{
    size_t off = base;
    r.desc_off = off; off = align_up(off + n * bytes_per_desc, 16);
    r.pos_off = off; off = align_up(off + n * 24, 16);
    r.valid_off = off; off += n;
    off = align_up(off + 64, 256);      // tail padding: clamped lanes may read one row past nothing, never past the extent
    r.ext_off = base; r.ext_size = off - base;
    return off - base;
}

int next_pow2(int v)
{
    int p = 4;
    while (p < v) p <<= 1;
    return p;
}

// Builds combos/jobs, uploads them and enqueues knn2 + estimate + D2H on the handle's stream.
int do_launch(uzl_match* h, int32_t n_jobs, const uzl_pair_job* jobs, const int32_t* frame_ids,
              int32_t n_frame_ids, int32_t max_corr)
{
    if (h->in_flight) return fail(h, UZL_ERR_BUSY, "a batch is already in flight");
    if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !frame_ids))) return fail(h, UZL_ERR_BAD_ARG, "null jobs/frame_ids");
    if (h->cfg.ransac_iteration < 1 || h->cfg.ransac_iteration > kMaxIterations)
        return fail(h, UZL_ERR_BAD_ARG, "ransac_iteration out of range [1,4096]");
    UZL_HIP(hipSetDevice(h->cfg.device));
    h->timer.reset();
    h->fl_jobs = n_jobs;
    h->fl_diag = max_corr > 0;
    h->fl_max_corr = max_corr;
    if (n_jobs == 0) { h->in_flight = true; h->fl_stride = 0; return UZL_OK; }

    // ---- combos: eligible (from, to) FeatureData pairs in the reference's loop order (:40-49)
    std::vector<Combo> combos;
    h->h_jobs.reserve((size_t)n_jobs);
    int max_nq = 0;
    int64_t total_q = 0;
    bool has8 = false, has16 = false, hasg = false;
    for (int32_t j = 0; j < n_jobs; j++) {
        const uzl_pair_job& pj = jobs[j];
        if (pj.from_begin < 0 || pj.to_begin < 0 || pj.from_count < 0 || pj.to_count < 0 ||
            (int64_t)pj.from_begin + pj.from_count > n_frame_ids || (int64_t)pj.to_begin + pj.to_count > n_frame_ids)
            return fail(h, UZL_ERR_BAD_ARG, "job frame range outside frame_ids");
        Job dj;
        dj.job_id = pj.job_id;
        dj.combo_begin = (int32_t)combos.size();
        dj.pq_off = 0; dj.pq_count = 0;
        for (int32_t a = 0; a < pj.from_count; a++) {
            const int32_t fid = frame_ids[pj.from_begin + a];
            if (fid < 0 || fid >= (int32_t)h->frames.size() || !h->frames[fid].alive)
                return fail(h, UZL_ERR_NOT_FOUND, "unknown from-frame id");
            const FrameRec& ff = h->frames[fid];
            for (int32_t b = 0; b < pj.to_count; b++) {
                const int32_t tid = frame_ids[pj.to_begin + b];
                if (tid < 0 || tid >= (int32_t)h->frames.size() || !h->frames[tid].alive)
                    return fail(h, UZL_ERR_NOT_FOUND, "unknown to-frame id");
                const FrameRec& ft = h->frames[tid];
                if (!(ff.n >= 7 && ft.n >= 7 && ff.feature_type == ft.feature_type &&
                      ff.sensor_frame == ft.sensor_frame && ff.words == ft.words)) continue;   // :47-49
                Combo c;
                c.desc_from_off = ff.desc_off / 4; c.desc_to_off = ft.desc_off / 4;
                c.pos_from_off = ff.pos_off / 8; c.pos_to_off = ft.pos_off / 8;
                c.valid_from_off = ff.valid_off; c.valid_to_off = ft.valid_off;
                c.nt = ff.n; c.nq = ft.n; c.words = ff.words;
                c.knn_off = (int32_t)total_q;
                c.frame_from = fid; c.frame_to = tid;
                total_q += ft.n;
                if (total_q > INT32_MAX) return fail(h, UZL_ERR_BAD_ARG, "batch too large (2-NN buffer > 2^31 entries)");
                max_nq = std::max(max_nq, ft.n);
                if (c.words == 8) has8 = true; else if (c.words == 16) has16 = true; else hasg = true;
                combos.push_back(c);
            }
        }
        dj.combo_count = (int32_t)combos.size() - dj.combo_begin;
        h->h_jobs.p[j] = dj;
    }
    const int stride = std::max(max_nq, 1);
    h->fl_stride = stride;
    const size_t nc = combos.size();
    h->h_combos.reserve(std::max<size_t>(nc, 1));
    if (nc) memcpy(h->h_combos.p, combos.data(), nc * sizeof(Combo));
    h->d_combos.reserve(std::max<size_t>(nc, 1));
    h->d_jobs.reserve((size_t)n_jobs);
    h->d_knn.reserve((size_t)std::max<int64_t>(total_q, 1));
    h->d_results.reserve((size_t)n_jobs);
    h->h_results.reserve((size_t)n_jobs);
    hipStream_t s = h->stream;
    if (nc) UZL_HIP(hipMemcpyAsync(h->d_combos.p, h->h_combos.p, nc * sizeof(Combo), hipMemcpyHostToDevice, s));
    UZL_HIP(hipMemcpyAsync(h->d_jobs.p, h->h_jobs.p, (size_t)n_jobs * sizeof(Job), hipMemcpyHostToDevice, s));

    // ---- M1
    h->timer.begin("knn2", s);
    launch_knn2(reinterpret_cast<const uint32_t*>(h->arena.p), h->d_combos.p, (int)nc, max_nq, h->d_knn.p,
                has8, has16, hasg, s);
    h->timer.end(s);
    UZL_HIP(hipGetLastError());

    // ---- M2..M9
    EstimateArgs a;
    memset(&a, 0, sizeof(a));
    a.arena = h->arena.p; a.combos = h->d_combos.p; a.jobs = h->d_jobs.p; a.knn = h->d_knn.p;
    a.prm.thresh = h->cfg.ransac_threshold; a.prm.break_pct = h->cfg.ransac_break_percentage;
    a.prm.seed = h->cfg.seed; a.prm.iterations = h->cfg.ransac_iteration; a.prm.do_prosac = h->cfg.do_prosac ? 1 : 0;
    a.prm.max_corr = stride;
    // results go straight to the pinned host array the caller's copy is taken from (432 B per job written over PCIe by one lane;
    // hipHostMalloc memory is mapped into the device's address space): no device-to-host copy operation behind the kernel
    a.results = h->h_results.p;
    a.sort_cap = next_pow2(max_nq);
    { static const bool vv = diag_flag("UZL_VOTE_VALU"); a.vote_valu = vv ? 1 : 0; }
    if (h->fl_diag) {
        const size_t tot = (size_t)n_jobs * stride;
        h->d_cq.reserve(tot); h->d_ct.reserve(tot); h->d_cd.reserve(tot); h->d_mask.reserve(tot);
        h->h_cq.reserve(tot); h->h_ct.reserve(tot); h->h_cd.reserve(tot); h->h_mask.reserve(tot);
        a.corr_query = h->d_cq.p; a.corr_train = h->d_ct.p; a.corr_dist = h->d_cd.p; a.inlier_mask = h->d_mask.p;
    }
    const int lds_points = (stride + 1) & ~1;
    bool in_lds = estimate_lds_bytes(a.sort_cap, a.prm.iterations, lds_points, true) <= kLdsBudget;
    a.prm.lds_points = lds_points;
    if (!in_lds) {
        const size_t tot = (size_t)n_jobs * stride;
        h->d_pq_scratch.reserve(tot * 6); h->d_dist_scratch.reserve(tot); h->d_mask_scratch.reserve(tot);
        a.pq_scratch = h->d_pq_scratch.p; a.dist_scratch = h->d_dist_scratch.p; a.mask_scratch = h->d_mask_scratch.p;
    }
    const size_t lds = estimate_lds_bytes(a.sort_cap, a.prm.iterations, lds_points, in_lds);
    if (lds > kLdsBudget) return fail(h, UZL_ERR_BAD_ARG, "frame too large for the LDS sort (n > 16384)");
    h->timer.begin("estimate", s);
    UZL_HIP(launch_estimate(a, n_jobs, in_lds, lds, s));
    h->timer.end(s);

    if (h->fl_diag) {
        const size_t tot = (size_t)n_jobs * stride;
        UZL_HIP(hipMemcpyAsync(h->h_cq.p, h->d_cq.p, tot * 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_ct.p, h->d_ct.p, tot * 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_cd.p, h->d_cd.p, tot * 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_mask.p, h->d_mask.p, tot, hipMemcpyDeviceToHost, s));
    }
    h->in_flight = true;
    return UZL_OK;
}

int do_collect(uzl_match* h, uzl_edge_result* results, int32_t* corr_query, int32_t* corr_train,
               int32_t* corr_dist, uint8_t* inlier_mask)
{
    if (!h->in_flight) return fail(h, UZL_ERR_STATE, "no batch in flight");
    UZL_HIP(hipSetDevice(h->cfg.device));
    UZL_HIP(hipStreamSynchronize(h->stream));
    h->in_flight = false;
    for (const Extent& e : h->deferred_free) arena_free_now(h, e.off, e.size);
    h->deferred_free.clear();
    h->timer.resolve();
    const int32_t n = h->fl_jobs;
    if (n > 0 && !results) return fail(h, UZL_ERR_BAD_ARG, "results is null");
    if (n > 0) memcpy(results, h->h_results.p, (size_t)n * sizeof(uzl_edge_result));
    if (h->fl_diag) {
        const int32_t mc = h->fl_max_corr, st = h->fl_stride;
        for (int32_t j = 0; j < n; j++) {
            const int32_t m = std::min(results[j].n_corr, mc);
            const size_t src = (size_t)j * st, dst = (size_t)j * mc;
            if (corr_query) { memcpy(corr_query + dst, h->h_cq.p + src, (size_t)m * 4); for (int32_t k = m; k < mc; k++) corr_query[dst + k] = -1; }
            if (corr_train) { memcpy(corr_train + dst, h->h_ct.p + src, (size_t)m * 4); for (int32_t k = m; k < mc; k++) corr_train[dst + k] = -1; }
            if (corr_dist) { memcpy(corr_dist + dst, h->h_cd.p + src, (size_t)m * 4); for (int32_t k = m; k < mc; k++) corr_dist[dst + k] = -1; }
            if (inlier_mask) { memcpy(inlier_mask + dst, h->h_mask.p + src, (size_t)m); for (int32_t k = m; k < mc; k++) inlier_mask[dst + k] = 0; }
        }
    }
    return UZL_OK;
}

// RANSAC over point sets already resident on the device (3 x total column-major at dP/dQ); enqueues on the handle's
// stream and leaves results in h->d_results / h->d_mask (row stride *stride_out).  Used by uzl_ransac_points and by
// the edge filter (uzl_filter.hip), whose pose-chain kernel produces the points on the same stream.
int enqueue_ransac(uzl_match* h, int32_t n_problems, const int32_t* offsets, const double* dP, const double* dQ,
                   double max_error, int32_t iterations, double break_percentage, int32_t do_prosac,
                   const uint64_t* job_ids, int* stride_out)
{
    int max_m = 0;
    h->h_jobs.reserve((size_t)n_problems);
    for (int32_t b = 0; b < n_problems; b++) {
        const int32_t m = offsets[b + 1] - offsets[b];
        max_m = std::max(max_m, m);
        Job j;
        j.job_id = job_ids ? job_ids[b] : (uint64_t)b;
        j.combo_begin = 0; j.combo_count = 0; j.pq_off = offsets[b]; j.pq_count = m;
        h->h_jobs.p[b] = j;
    }
    const int stride = std::max(max_m, 1);
    hipStream_t s = h->stream;
    h->d_jobs.reserve((size_t)n_problems);
    h->d_results.reserve((size_t)n_problems);
    const size_t tot = (size_t)n_problems * stride;
    h->d_mask.reserve(tot);
    UZL_HIP(hipMemcpyAsync(h->d_jobs.p, h->h_jobs.p, (size_t)n_problems * sizeof(Job), hipMemcpyHostToDevice, s));
    EstimateArgs a;
    memset(&a, 0, sizeof(a));
    a.jobs = h->d_jobs.p;
    a.prm.thresh = max_error; a.prm.break_pct = break_percentage; a.prm.seed = h->cfg.seed;
    a.prm.iterations = iterations; a.prm.do_prosac = do_prosac ? 1 : 0; a.prm.max_corr = stride;
    a.results = h->d_results.p;
    a.P_in = dP; a.Q_in = dQ;
    a.inlier_mask = h->d_mask.p;
    a.sort_cap = 4;
    { static const bool vv = diag_flag("UZL_VOTE_VALU"); a.vote_valu = vv ? 1 : 0; }
    const int lds_points = (stride + 1) & ~1;
    a.prm.lds_points = lds_points;
    const bool in_lds = estimate_lds_bytes(a.sort_cap, iterations, lds_points, true) <= kLdsBudget;
    if (!in_lds) {
        h->d_pq_scratch.reserve(tot * 6); h->d_dist_scratch.reserve(tot); h->d_mask_scratch.reserve(tot);
        a.pq_scratch = h->d_pq_scratch.p; a.dist_scratch = h->d_dist_scratch.p; a.mask_scratch = h->d_mask_scratch.p;
    }
    const size_t lds = estimate_lds_bytes(a.sort_cap, iterations, lds_points, in_lds);
    h->timer.reset();
    h->timer.begin("ransac_points", s);
    UZL_HIP(launch_estimate(a, n_problems, in_lds, lds, s));
    h->timer.end(s);
    *stride_out = stride;
    return UZL_OK;
}

}  // namespace

#define UZL_GUARD_BEGIN(h)                       \
    if (!(h)) return UZL_ERR_BAD_ARG;            \
    std::lock_guard<std::mutex> lock_((h)->mu);  \
    try {
#define UZL_GUARD_END(h)                                                             \
    } catch (const ::uzl::HipError& e) { return ::uzl::report((h)->last_error, e); } \
    catch (const std::bad_alloc&) { (h)->last_error = "host out of memory"; return UZL_ERR_OOM; } \
    catch (...) { (h)->last_error = "unexpected exception"; return UZL_ERR_HIP; }

extern "C" {

int uzl_abi_version(void) { return UZL_ABI_VERSION; }

int uzl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return UZL_ERR_NO_DEVICE;
    return n;
}

const char* uzl_status_string(int status)
{
    switch (status) {
        case UZL_OK: return "ok";
        case UZL_ERR_BAD_ARG: return "bad argument";
        case UZL_ERR_NO_DEVICE: return "no HIP device";
        case UZL_ERR_HIP: return "HIP runtime error";
        case UZL_ERR_NOT_CONVERGED: return "PCG did not converge";
        case UZL_ERR_BUSY: return "busy";
        case UZL_ERR_OOM: return "out of memory";
        case UZL_ERR_NOT_FOUND: return "not found";
        case UZL_ERR_STATE: return "call order violated";
        case UZL_ERR_TRUNCATED: return "message truncated or output buffer too small";
        case UZL_ERR_UNSUPPORTED: return "unsupported format variant";
        default: return "unknown status";
    }
}

void uzl_match_cfg_default(uzl_match_cfg* cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->ransac_threshold = 0.2;           // cfg/FeatureLinkEstimation.cfg:9
    cfg->link_covariance = 0.01;           // :10
    cfg->ransac_iteration = 100;           // :11
    cfg->ransac_break_percentage = 0.6;    // :12
    cfg->use_epnp = 1;                     // :13
    cfg->do_prosac = 1;                    // estimateSVD default argument (feature_transformation_estimator.h:45)
    cfg->device = 0;
    cfg->seed = 0;
}

int uzl_match_create(const uzl_match_cfg* cfg, uzl_match** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_match_cfg c;
    if (cfg) c = *cfg; else uzl_match_cfg_default(&c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UZL_ERR_NO_DEVICE;
    if (c.device < 0 || c.device >= ndev) return UZL_ERR_NO_DEVICE;
    uzl_match* h = new (std::nothrow) uzl_match();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    try {
        UZL_HIP(hipSetDevice(c.device));
        UZL_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        stream_register(c.device, h->stream, true);               // (its launch sequences run beside a solve: config 5)
        h->arena.reserve((size_t)64 << 20);
    } catch (const HipError& e) {
        std::string msg;
        int code = report(msg, e);
        if (h->stream) { stream_unregister(c.device, h->stream); (void)hipStreamDestroy(h->stream); }
        delete h;
        return code;
    }
    *out = h;
    return UZL_OK;
}

void uzl_match_destroy(uzl_match* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_unregister(h->cfg.device, h->stream); (void)hipStreamDestroy(h->stream); }
    for (auto& e : h->up_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : h->bulk_ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

int uzl_match_set_config(uzl_match* h, const uzl_match_cfg* cfg)
{
    if (!h || !cfg) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (cfg->device != h->cfg.device) return fail(h, UZL_ERR_BAD_ARG, "device cannot change after create");
    if (cfg->ransac_iteration < 1 || cfg->ransac_iteration > kMaxIterations)
        return fail(h, UZL_ERR_BAD_ARG, "ransac_iteration out of range [1,4096]");
    h->cfg = *cfg;
    return UZL_OK;
}

const char* uzl_match_last_error(uzl_match* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_match_add_frame(uzl_match* h, const uzl_frame* f, int32_t* frame_id)
{
    if (!f || !frame_id) return UZL_ERR_BAD_ARG;
    UZL_GUARD_BEGIN(h)
    if (f->n < 0 || f->n > kMaxKeypoints) return fail(h, UZL_ERR_BAD_ARG, "frame.n out of range [0,16384]");
    if (f->bytes_per_desc <= 0 || f->bytes_per_desc % 4 != 0 || f->bytes_per_desc > 508)
        return fail(h, UZL_ERR_BAD_ARG, "bytes_per_desc must be a multiple of 4 in [4,508]");
    if (f->n > 0 && (!f->desc || !f->pos_xyz || !f->valid3d)) return fail(h, UZL_ERR_BAD_ARG, "null frame arrays");
    UZL_HIP(hipSetDevice(h->cfg.device));
    const size_t n = (size_t)f->n;
    const size_t desc_b = n * (size_t)f->bytes_per_desc, pos_b = n * 24, val_b = n;
    FrameRec r;
    const size_t need = frame_layout(n, (size_t)f->bytes_per_desc, r, 0);
    const size_t used_before = h->arena_used;
    const std::vector<Extent> free_before = h->free_list;
    const size_t base = arena_alloc(h, need);
    if (h->in_flight && h->arena_used > h->arena.cap) {               // (growing re-allocates: not under a batch that reads the arena)
        h->arena_used = used_before; h->free_list = free_before;
        return fail(h, UZL_ERR_BUSY, "arena must grow while a batch is in flight");
    }
    frame_layout(n, (size_t)f->bytes_per_desc, r, base);
    arena_grow(h, [&]() { h->arena_used = used_before; h->free_list = free_before; });
    if (n) {
        const size_t span = r.valid_off + val_b - r.desc_off;        // [desc | pad | pos | pad | valid] as it lies in the arena
        if (span <= kUpHalf) {
            if (!h->h_up.p) {
                h->h_up.reserve(2 * kUpHalf);
                for (auto& e : h->up_ev) UZL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            if (h->up_used + span > kUpHalf) {                       // this half is full: close it, move to the other one once it is free
                UZL_HIP(hipEventRecord(h->up_ev[h->up_half], h->stream));
                h->up_pending[h->up_half] = true;
                h->up_half ^= 1; h->up_used = 0;
                if (h->up_pending[h->up_half]) { UZL_HIP(hipEventSynchronize(h->up_ev[h->up_half])); h->up_pending[h->up_half] = false; }
            }
            uint8_t* st = h->h_up.p + (size_t)h->up_half * kUpHalf + h->up_used;
            memcpy(st, f->desc, desc_b);
            memcpy(st + (r.pos_off - r.desc_off), f->pos_xyz, pos_b);
            memcpy(st + (r.valid_off - r.desc_off), f->valid3d, val_b);
            UZL_HIP(hipMemcpyAsync(h->arena.p + r.desc_off, st, span, hipMemcpyHostToDevice, h->stream));
            h->up_used += align_up(span, 256);
        } else {                                                     // a frame larger than the staging half: straight from the caller's arrays
            UZL_HIP(hipMemcpyAsync(h->arena.p + r.desc_off, f->desc, desc_b, hipMemcpyHostToDevice, h->stream));
            UZL_HIP(hipMemcpyAsync(h->arena.p + r.pos_off, f->pos_xyz, pos_b, hipMemcpyHostToDevice, h->stream));
            UZL_HIP(hipMemcpyAsync(h->arena.p + r.valid_off, f->valid3d, val_b, hipMemcpyHostToDevice, h->stream));
            UZL_HIP(hipStreamSynchronize(h->stream));   // inputs are borrowed only for the duration of the call
        }
    }
    r.alive = true; r.n = f->n; r.words = f->bytes_per_desc / 4;
    r.feature_type = f->feature_type; r.sensor_frame = f->sensor_frame;
    h->frames.push_back(r);
    h->live_frames++; h->live_bytes += r.ext_size;
    *frame_id = (int32_t)h->frames.size() - 1;
    return UZL_OK;
    UZL_GUARD_END(h)
}

// n FeatureData at once (the adapter's batching worker holds that many; transformation_estimator.cpp:35-43 copies one node pair per
// call).  The frames get ONE contiguous arena extent; host threads pack them into pinned staging, half by half, and every half goes up
// as one DMA while the next one is being packed.  Returns without waiting for the last copy (the inputs are packed by then).
int uzl_match_add_frames(uzl_match* h, int32_t n_frames, const uzl_frame* f, int32_t* frame_ids)
{
    if (n_frames < 0 || (n_frames > 0 && (!f || !frame_ids))) return UZL_ERR_BAD_ARG;
    UZL_GUARD_BEGIN(h)
    if (n_frames == 0) return UZL_OK;
    for (int32_t k = 0; k < n_frames; k++) {
        if (f[k].n < 0 || f[k].n > kMaxKeypoints) return fail(h, UZL_ERR_BAD_ARG, "frame.n out of range [0,16384]");
        if (f[k].bytes_per_desc <= 0 || f[k].bytes_per_desc % 4 != 0 || f[k].bytes_per_desc > 508)
            return fail(h, UZL_ERR_BAD_ARG, "bytes_per_desc must be a multiple of 4 in [4,508]");
        if (f[k].n > 0 && (!f[k].desc || !f[k].pos_xyz || !f[k].valid3d)) return fail(h, UZL_ERR_BAD_ARG, "null frame arrays");
    }
    UZL_HIP(hipSetDevice(h->cfg.device));
    std::vector<FrameRec> recs((size_t)n_frames);
    size_t total = 0;
    for (int32_t k = 0; k < n_frames; k++) total += frame_layout((size_t)f[k].n, (size_t)f[k].bytes_per_desc, recs[k], total);
    const size_t used_before = h->arena_used;
    const std::vector<Extent> free_before = h->free_list;
    // one contiguous extent when a hole (or the top of the arena) takes the whole batch; otherwise frame by frame into the holes
    bool one_extent = h->free_list.empty();
    for (const Extent& e : h->free_list) one_extent = one_extent || e.size >= align_up(total, 256);
    if (one_extent) {
        const size_t base = arena_alloc(h, total);
        for (int32_t k = 0; k < n_frames; k++) { recs[k].desc_off += base; recs[k].pos_off += base; recs[k].valid_off += base; recs[k].ext_off += base; }
    } else {
        for (int32_t k = 0; k < n_frames; k++) {
            const size_t rel = recs[k].ext_off, base = arena_alloc(h, recs[k].ext_size);
            recs[k].desc_off += base - rel; recs[k].pos_off += base - rel; recs[k].valid_off += base - rel; recs[k].ext_off = base;
        }
    }
    if (h->in_flight && h->arena_used > h->arena.cap) {
        h->arena_used = used_before; h->free_list = free_before;
        return fail(h, UZL_ERR_BUSY, "arena must grow while a batch is in flight");
    }
    arena_grow(h, [&]() { h->arena_used = used_before; h->free_list = free_before; });
    for (int32_t k = 0; k < n_frames; k++) {
        FrameRec& r = recs[k];
        r.alive = true; r.n = f[k].n; r.words = f[k].bytes_per_desc / 4; r.feature_type = f[k].feature_type; r.sensor_frame = f[k].sensor_frame;
    }
    if (!h->h_bulk.p) {
        h->h_bulk.reserve(2 * kBulkHalf);
        for (auto& e : h->bulk_ev) UZL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    std::vector<size_t> soff((size_t)n_frames);
    int32_t k0 = 0;
    int half = 0;
    while (k0 < n_frames) {
        // frames [k0, k1): as many as fit one staging half (a frame's extent is at most ~8.7 MB), packed back to back
        int32_t k1 = k0;
        size_t bytes = 0;
        while (k1 < n_frames && bytes + recs[k1].ext_size <= kBulkHalf) { soff[k1] = bytes; bytes += recs[k1].ext_size; k1++; }
        if (h->bulk_pending[half]) { UZL_HIP(hipEventSynchronize(h->bulk_ev[half])); h->bulk_pending[half] = false; }
        uint8_t* st = h->h_bulk.p + (size_t)half * kBulkHalf;
        auto pack = [&](int32_t a, int32_t b) {
            for (int32_t k = a; k < b; k++) {
                const size_t n = (size_t)f[k].n;
                if (!n) continue;
                uint8_t* d = st + soff[k];
                memcpy(d + (recs[k].desc_off - recs[k].ext_off), f[k].desc, n * (size_t)f[k].bytes_per_desc);
                memcpy(d + (recs[k].pos_off - recs[k].ext_off), f[k].pos_xyz, n * 24);
                memcpy(d + (recs[k].valid_off - recs[k].ext_off), f[k].valid3d, n);
            }
        };
        const int32_t cnt = k1 - k0;
        const int nt = (int)std::min<size_t>({(size_t)8, (size_t)hw, (size_t)cnt, std::max<size_t>(1, bytes >> 20)});      // ~1 MB per thread at least
        if (nt <= 1) pack(k0, k1);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(pack, k0 + (int32_t)((int64_t)cnt * t / nt), k0 + (int32_t)((int64_t)cnt * (t + 1) / nt));
            pack(k0, k0 + cnt / nt);
            for (auto& t : th) t.join();
        }
        for (int32_t a = k0; a < k1;) {                              // one DMA per run of frames that are neighbours in the arena too
            int32_t b = a + 1;
            while (b < k1 && recs[b].ext_off == recs[b - 1].ext_off + recs[b - 1].ext_size) b++;
            const size_t run_bytes = soff[b - 1] + recs[b - 1].ext_size - soff[a];
            UZL_HIP(hipMemcpyAsync(h->arena.p + recs[a].ext_off, st + soff[a], run_bytes, hipMemcpyHostToDevice, h->stream));
            a = b;
        }
        UZL_HIP(hipEventRecord(h->bulk_ev[half], h->stream));
        h->bulk_pending[half] = true;
        half ^= 1;
        k0 = k1;
    }
    for (int32_t k = 0; k < n_frames; k++) {
        h->frames.push_back(recs[k]);
        h->live_frames++; h->live_bytes += recs[k].ext_size;
        frame_ids[k] = (int32_t)h->frames.size() - 1;
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_match_arena_bytes(uzl_match* h, uint64_t* live, uint64_t* high_water, uint64_t* capacity)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (live) *live = h->live_bytes;
    if (high_water) *high_water = h->arena_used;
    if (capacity) *capacity = h->arena.cap;
    return UZL_OK;
}

int uzl_match_remove_frame(uzl_match* h, int32_t frame_id)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (frame_id < 0 || frame_id >= (int32_t)h->frames.size() || !h->frames[frame_id].alive)
        return fail(h, UZL_ERR_NOT_FOUND, "unknown frame id");
    FrameRec& r = h->frames[frame_id];
    r.alive = false;
    h->live_frames--; h->live_bytes -= r.ext_size;
    arena_free(h, r.ext_off, r.ext_size);      // back to the free list (after the batch in flight, if there is one)
    return UZL_OK;
}

// FeatureData::fromMsg (sensor_data.cpp:123-167) for a batch of frames: raw Feature records go to HBM as they are and one
// launch unpacks all of them into the arena.
int uzl_match_add_frames_wire(uzl_match* h, int32_t n_frames, const uzl_wire_sensor* sensors, const int32_t* sensor_frame_keys,
                              int32_t* frame_ids, int32_t* uv)
{
    if (n_frames < 0 || (n_frames > 0 && (!sensors || !frame_ids))) return UZL_ERR_BAD_ARG;
    UZL_GUARD_BEGIN(h)
    if (n_frames == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    std::vector<WireSeg> segs((size_t)n_frames + 1);      // + sentinel
    std::vector<FrameRec> recs((size_t)n_frames);
    size_t off = 0;                                       // relative to the batch's extent, which is allocated below
    uint64_t src = 0;
    int64_t items = 0, feats = 0;
    for (int32_t k = 0; k < n_frames; k++) {
        const uzl_wire_sensor& f = sensors[k];
        if (f.n_features < 0 || f.n_features > kMaxKeypoints) return fail(h, UZL_ERR_BAD_ARG, "frame.n out of range [0,16384]");
        // binary descriptor types only (sensor_data.cpp:130-140); SURF / SIFT rows are floats and never reach the Hamming matcher
        if (f.descriptor_type < UZL_FEATURE_BRIEF || f.descriptor_type > UZL_FEATURE_FREAK)
            return fail(h, UZL_ERR_UNSUPPORTED, "descriptor_type is not a binary descriptor (BRIEF/ORB/BRISK/FREAK)");
        if (f.n_features > 0) {
            if (!f.uniform) return fail(h, UZL_ERR_UNSUPPORTED, "Feature records with differing descriptor lengths");
            if (f.desc_len <= 0 || f.desc_len % 4 != 0 || f.desc_len > 508)
                return fail(h, UZL_ERR_BAD_ARG, "descriptor length must be a multiple of 4 in [4,508]");
            if (!f.records.p || f.records.n != uzl_wire_features_size(f.n_features, f.desc_len))
                return fail(h, UZL_ERR_BAD_ARG, "records span does not hold n_features records of desc_len elements");
        }
        const size_t n = (size_t)f.n_features;
        const int32_t D = f.n_features > 0 ? f.desc_len : 32;
        FrameRec& r = recs[k];
        off += frame_layout(n, (size_t)D, r, off);
        r.alive = true; r.n = f.n_features; r.words = D / 4;
        r.feature_type = f.descriptor_type; r.sensor_frame = sensor_frame_keys ? sensor_frame_keys[k] : 0;
        WireSeg& g = segs[k];
        g.item_begin = items; g.feat_begin = feats; g.src_off = src;
        g.desc_off = r.desc_off; g.pos_off = r.pos_off; g.valid_off = r.valid_off;
        g.stride = (uint32_t)(41 + 4 * D); g.words = D / 4; g.n = f.n_features; g._pad = wire_kpb(g.stride);
        items += ((int64_t)n + g._pad - 1) / g._pad; feats += (int64_t)n;          // items: workgroups
        src = align_up(src + (n ? f.records.n : 0), 16);                          // every frame's records start 16-byte aligned
    }
    memset(&segs[(size_t)n_frames], 0, sizeof(WireSeg));
    segs[(size_t)n_frames].item_begin = items;
    const size_t used_before = h->arena_used;
    const std::vector<Extent> free_before = h->free_list;
    const size_t base = arena_alloc(h, off);
    auto undo_alloc = [&]() { h->arena_used = used_before; h->free_list = free_before; };
    if (h->in_flight && h->arena_used > h->arena.cap) { undo_alloc(); return fail(h, UZL_ERR_BUSY, "arena must grow while a batch is in flight"); }
    for (int32_t k = 0; k < n_frames; k++) {
        recs[k].desc_off += base; recs[k].pos_off += base; recs[k].valid_off += base; recs[k].ext_off += base;
        segs[k].desc_off += base; segs[k].pos_off += base; segs[k].valid_off += base;
    }
    arena_grow(h, undo_alloc);
    h->d_wire_stage.reserve((size_t)(src / 4) + 8);                              // + tail: the 16 bytes after the last record may be read
    h->d_wire_segs.reserve((size_t)n_frames + 1);
    h->d_wire_bad.reserve(1);
    if (uv) h->d_wire_uv.reserve((size_t)std::max<int64_t>(2 * feats, 1));
    for (int32_t k = 0; k < n_frames; k++)
        if (sensors[k].n_features > 0)
            UZL_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t*>(h->d_wire_stage.p) + segs[k].src_off, sensors[k].records.p, sensors[k].records.n,
                                   hipMemcpyHostToDevice, h->stream));
    UZL_HIP(hipMemcpyAsync(h->d_wire_segs.p, segs.data(), sizeof(WireSeg) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, h->stream));
    UZL_HIP(hipMemsetAsync(h->d_wire_bad.p, 0, 4, h->stream));
    if (!h->in_flight) h->timer.reset();       // the events of a batch still in flight are resolved by its collect
    h->timer.begin("wire_unpack", h->stream);
    launch_wire_unpack(h->d_wire_stage.p, h->arena.p, h->d_wire_segs.p, n_frames, items, uv ? h->d_wire_uv.p : nullptr, h->d_wire_bad.p, h->stream);
    h->timer.end(h->stream);
    UZL_HIP(hipGetLastError());
    int32_t bad = 0;
    UZL_HIP(hipMemcpyAsync(&bad, h->d_wire_bad.p, 4, hipMemcpyDeviceToHost, h->stream));
    if (uv && feats) UZL_HIP(hipMemcpyAsync(uv, h->d_wire_uv.p, (size_t)feats * 8, hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipStreamSynchronize(h->stream));                                    // inputs are borrowed only for the duration of the call
    h->timer.resolve();
    if (bad) { undo_alloc(); return fail(h, UZL_ERR_BAD_ARG, "a Feature record's descriptor count differs from desc_len"); }
    for (int32_t k = 0; k < n_frames; k++) {
        h->frames.push_back(recs[k]);
        h->live_frames++; h->live_bytes += recs[k].ext_size;
        frame_ids[k] = (int32_t)h->frames.size() - 1;
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

// FeatureData::toMsg (sensor_data.cpp:78-121): the Feature records of a resident frame
int uzl_match_frame_to_wire(uzl_match* h, int32_t frame_id, const int32_t* uv, uint8_t* records, uint64_t cap, uint64_t* written)
{
    UZL_GUARD_BEGIN(h)
    if (frame_id < 0 || frame_id >= (int32_t)h->frames.size() || !h->frames[frame_id].alive)
        return fail(h, UZL_ERR_NOT_FOUND, "unknown frame id");
    const FrameRec& r = h->frames[frame_id];
    const uint64_t bytes = uzl_wire_features_size(r.n, 4 * r.words);
    if (written) *written = bytes;
    if (bytes == 0) return UZL_OK;
    if (!records || cap < bytes) return fail(h, UZL_ERR_TRUNCATED, "records buffer too small");
    UZL_HIP(hipSetDevice(h->cfg.device));
    WireSeg g;
    memset(&g, 0, sizeof(g));
    g.desc_off = r.desc_off; g.pos_off = r.pos_off; g.valid_off = r.valid_off;
    g.stride = (uint32_t)(41 + 16 * r.words); g.words = r.words; g.n = r.n;
    h->d_wire_stage.reserve((size_t)(bytes / 4) + 4);
    if (uv) {
        h->d_wire_uv.reserve((size_t)r.n * 2);
        UZL_HIP(hipMemcpyAsync(h->d_wire_uv.p, uv, (size_t)r.n * 8, hipMemcpyHostToDevice, h->stream));
    }
    if (!h->in_flight) h->timer.reset();
    h->timer.begin("wire_pack", h->stream);
    launch_wire_pack(h->arena.p, g, uv ? h->d_wire_uv.p : nullptr, h->d_wire_stage.p, bytes, h->stream);
    h->timer.end(h->stream);
    UZL_HIP(hipGetLastError());
    UZL_HIP(hipMemcpyAsync(records, h->d_wire_stage.p, bytes, hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipStreamSynchronize(h->stream));
    h->timer.resolve();
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_match_get_frame(uzl_match* h, int32_t frame_id, uint8_t* desc, double* pos_xyz, uint8_t* valid3d, int32_t* n, int32_t* bytes_per_desc)
{
    UZL_GUARD_BEGIN(h)
    if (frame_id < 0 || frame_id >= (int32_t)h->frames.size() || !h->frames[frame_id].alive)
        return fail(h, UZL_ERR_NOT_FOUND, "unknown frame id");
    const FrameRec& r = h->frames[frame_id];
    if (n) *n = r.n;
    if (bytes_per_desc) *bytes_per_desc = 4 * r.words;
    if (r.n == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    if (desc) UZL_HIP(hipMemcpyAsync(desc, h->arena.p + r.desc_off, (size_t)r.n * 4 * r.words, hipMemcpyDeviceToHost, h->stream));
    if (pos_xyz) UZL_HIP(hipMemcpyAsync(pos_xyz, h->arena.p + r.pos_off, (size_t)r.n * 24, hipMemcpyDeviceToHost, h->stream));
    if (valid3d) UZL_HIP(hipMemcpyAsync(valid3d, h->arena.p + r.valid_off, (size_t)r.n, hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipStreamSynchronize(h->stream));
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_match_frame_count(uzl_match* h)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return h->live_frames;
}

int uzl_match_launch(uzl_match* h, int32_t n_jobs, const uzl_pair_job* jobs, const int32_t* frame_ids,
                     int32_t n_frame_ids, int32_t max_corr)
{
    UZL_GUARD_BEGIN(h)
    return do_launch(h, n_jobs, jobs, frame_ids, n_frame_ids, max_corr);
    UZL_GUARD_END(h)
}

int uzl_match_collect(uzl_match* h, uzl_edge_result* results, int32_t* corr_query, int32_t* corr_train,
                      int32_t* corr_dist, uint8_t* inlier_mask)
{
    UZL_GUARD_BEGIN(h)
    return do_collect(h, results, corr_query, corr_train, corr_dist, inlier_mask);
    UZL_GUARD_END(h)
}

int uzl_match_estimate(uzl_match* h, int32_t n_jobs, const uzl_pair_job* jobs, const int32_t* frame_ids,
                       int32_t n_frame_ids, uzl_edge_result* results, int32_t max_corr, int32_t* corr_query,
                       int32_t* corr_train, int32_t* corr_dist, uint8_t* inlier_mask)
{
    UZL_GUARD_BEGIN(h)
    const bool diag = corr_query || corr_train || corr_dist || inlier_mask;
    int rc = do_launch(h, n_jobs, jobs, frame_ids, n_frame_ids, diag ? max_corr : 0);
    if (rc != UZL_OK) return rc;
    return do_collect(h, results, corr_query, corr_train, corr_dist, inlier_mask);
    UZL_GUARD_END(h)
}

int uzl_match_knn2(uzl_match* h, int32_t frame_from, int32_t frame_to, int32_t* idx0, int32_t* dist0,
                   int32_t* idx1, int32_t* dist1)
{
    UZL_GUARD_BEGIN(h)
    if (h->in_flight) return fail(h, UZL_ERR_BUSY, "a batch is in flight");
    if (frame_from < 0 || frame_from >= (int32_t)h->frames.size() || !h->frames[frame_from].alive ||
        frame_to < 0 || frame_to >= (int32_t)h->frames.size() || !h->frames[frame_to].alive)
        return fail(h, UZL_ERR_NOT_FOUND, "unknown frame id");
    const FrameRec& ff = h->frames[frame_from];
    const FrameRec& ft = h->frames[frame_to];
    if (ff.words != ft.words) return fail(h, UZL_ERR_BAD_ARG, "descriptor widths differ");
    if (ft.n == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    Combo c;
    memset(&c, 0, sizeof(c));
    c.desc_from_off = ff.desc_off / 4; c.desc_to_off = ft.desc_off / 4;
    c.nt = ff.n; c.nq = ft.n; c.words = ff.words; c.knn_off = 0;
    h->h_combos.reserve(1); h->d_combos.reserve(1);
    h->h_combos.p[0] = c;
    h->d_knn.reserve((size_t)ft.n);
    UZL_HIP(hipMemcpyAsync(h->d_combos.p, h->h_combos.p, sizeof(Combo), hipMemcpyHostToDevice, h->stream));
    launch_knn2(reinterpret_cast<const uint32_t*>(h->arena.p), h->d_combos.p, 1, ft.n, h->d_knn.p,
                c.words == 8, c.words == 16, c.words != 8 && c.words != 16, h->stream);
    UZL_HIP(hipGetLastError());
    std::vector<uint2> keys((size_t)ft.n);
    UZL_HIP(hipMemcpyAsync(keys.data(), h->d_knn.p, (size_t)ft.n * sizeof(uint2), hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipStreamSynchronize(h->stream));
    for (int32_t q = 0; q < ft.n; q++) {
        const uint32_t a = keys[q].x, b = keys[q].y;
        if (idx0) idx0[q] = (a == 0xffffffffu) ? -1 : (int32_t)(a & kIdxMask);
        if (dist0) dist0[q] = (a == 0xffffffffu) ? -1 : (int32_t)(a >> kIdxBits);
        if (idx1) idx1[q] = (b == 0xffffffffu) ? -1 : (int32_t)(b & kIdxMask);
        if (dist1) dist1[q] = (b == 0xffffffffu) ? -1 : (int32_t)(b >> kIdxBits);
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_ransac_points(uzl_match* h, int32_t n_problems, const int32_t* offsets, const double* P,
                      const double* Q, double max_error, int32_t iterations, double break_percentage,
                      int32_t do_prosac, const uint64_t* job_ids, double* T, int32_t* consensus,
                      double* mse, int32_t* iterations_run, uint8_t* mask)
{
    UZL_GUARD_BEGIN(h)
    if (h->in_flight) return fail(h, UZL_ERR_BUSY, "a batch is in flight");
    if (n_problems < 0 || (n_problems > 0 && (!offsets || !P || !Q))) return fail(h, UZL_ERR_BAD_ARG, "null arrays");
    if (iterations < 1 || iterations > kMaxIterations) return fail(h, UZL_ERR_BAD_ARG, "iterations out of range [1,4096]");
    if (n_problems == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    const int64_t total = offsets[n_problems];
    if (offsets[0] < 0) return fail(h, UZL_ERR_BAD_ARG, "offsets[0] must be >= 0");
    for (int32_t b = 0; b < n_problems; b++)
        if (offsets[b + 1] < offsets[b]) return fail(h, UZL_ERR_BAD_ARG, "offsets must be non-decreasing");
    hipStream_t s = h->stream;
    h->d_P.reserve((size_t)std::max<int64_t>(total, 1) * 3);
    h->d_Q.reserve((size_t)std::max<int64_t>(total, 1) * 3);
    if (total > 0) {
        UZL_HIP(hipMemcpyAsync(h->d_P.p, P, (size_t)total * 24, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_Q.p, Q, (size_t)total * 24, hipMemcpyHostToDevice, s));
    }
    int stride = 0;
    const int rc = enqueue_ransac(h, n_problems, offsets, h->d_P.p, h->d_Q.p, max_error, iterations, break_percentage, do_prosac, job_ids, &stride);
    if (rc != UZL_OK) return rc;
    const size_t tot = (size_t)n_problems * stride;
    h->h_results.reserve((size_t)n_problems); h->h_mask.reserve(tot);
    UZL_HIP(hipMemcpyAsync(h->h_results.p, h->d_results.p, (size_t)n_problems * sizeof(uzl_edge_result), hipMemcpyDeviceToHost, s));
    UZL_HIP(hipMemcpyAsync(h->h_mask.p, h->d_mask.p, tot, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipStreamSynchronize(s));
    h->timer.resolve();
    for (int32_t b = 0; b < n_problems; b++) {
        const uzl_edge_result& r = h->h_results.p[b];
        if (T) memcpy(T + 12 * (size_t)b, r.T, sizeof(r.T));
        if (consensus) consensus[b] = r.consensus;
        if (mse) mse[b] = r.mse;
        if (iterations_run) iterations_run[b] = r.iterations_run;
        if (mask) memcpy(mask + offsets[b], h->h_mask.p + (size_t)b * stride, (size_t)(offsets[b + 1] - offsets[b]));
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_match_set_profiling(uzl_match* h, int32_t on)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->timer.on = on != 0;
    return UZL_OK;
}

int uzl_match_kernel_times(uzl_match* h, int32_t cap, const char** names, double* ms, int32_t* launches)
{
    if (!h || cap < 0 || (cap > 0 && (!names || !ms || !launches))) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return h->timer.report(cap, names, ms, launches);
}

}  // extern "C"

// ---- internal interface for uzl_filter.hip (match_internal.hpp) ----
namespace uzl {

hipStream_t match_stream(uzl_match* h) { return h->stream; }

int match_ransac_device(uzl_match* h, int32_t n_problems, const int32_t* offsets, const double* dP, const double* dQ,
                        double max_error, int32_t iterations, double break_percentage, int32_t do_prosac,
                        const uint64_t* job_ids, MatchDeviceResults* out)
{
    UZL_GUARD_BEGIN(h)
    if (h->in_flight) return fail(h, UZL_ERR_BUSY, "a batch is in flight");
    if (iterations < 1 || iterations > kMaxIterations) return fail(h, UZL_ERR_BAD_ARG, "iterations out of range [1,4096]");
    UZL_HIP(hipSetDevice(h->cfg.device));
    int stride = 0;
    const int rc = enqueue_ransac(h, n_problems, offsets, dP, dQ, max_error, iterations, break_percentage, do_prosac, job_ids, &stride);
    if (rc != UZL_OK) return rc;
    out->results = h->d_results.p; out->mask = h->d_mask.p; out->stride = stride; out->stream = h->stream;
    return UZL_OK;
    UZL_GUARD_END(h)
}

}  // namespace uzl
