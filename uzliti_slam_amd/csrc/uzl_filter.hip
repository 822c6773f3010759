// uzl_filter.hip — host side of the edge filter + its C ABI (uzl_filter_*).
//
// Mirrors TransformationFilter / EdgeCluster (transformation_estimation/src/transformation_filter.cpp:43-350,
// include/transformation_estimation/transformation_filter.h:28-106).  The cluster bookkeeping is the reference's
// sequential logic on the host (ids are 64-bit keys; the adapter keeps the strings); calcValidEdges() sends every
// changed cluster of one call to the GPU as ONE batch: pose chains -> 3-point RANSAC (the estimator's kernel, one
// workgroup per cluster) -> consensus, all on one stream, one synchronisation.
#include "uzl_common.hpp"
#include "filter_types.hpp"
#include "match_internal.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <new>
#include <unordered_map>

namespace uzl {
void launch_filter_points(const FilterEdgeDev* edges, int n, const double* sensors, int n_sensors, double* P, double* Q, hipStream_t s);
void launch_filter_consensus(const double* P, const double* Q, const int32_t* col_cluster, int n, const uzl_edge_result* results,
                             double max_error, uint8_t* set, hipStream_t s);
}

using namespace uzl;

namespace {

// EdgeData (transformation_filter.h:28-46)
struct EdgeRec {
    uint64_t key = 0;
    FilterEdgeDev geo;            // pos_from_/pos_to_ and the stored SlamEdge's transforms
    int64_t time_from = 0, time_to = 0;
    double score = 0.0;
    bool edge_valid = false;      // SlamEdge::valid_ of the stored copy
    bool valid = false;           // EdgeData::valid_
};

// EdgeCluster (transformation_filter.h:48-78).  edges_ is an unordered_map in the reference; here a vector in
// insertion order plus an index, so that the order the reference leaves open is defined.
struct Cluster {
    uint64_t uid = 0;
    int64_t from_start = 0, from_end = 0, to_start = 0, to_end = 0;
    bool changed = false;
    int consensus = 0;
    int evaluations = 0;
    std::vector<EdgeRec> edges;
    std::unordered_map<uint64_t, int> where;
    // last GPU evaluation (introspection)
    std::vector<double> lastP, lastQ;
    double lastT[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int last_ransac = 0;

    int size() const { return (int)edges.size(); }

    void put(const EdgeRec& r)                                      // edges_[edge.id_] = data; if (edge.valid_) consensus_++
    {
        auto it = where.find(r.key);
        if (it == where.end()) { where[r.key] = (int)edges.size(); edges.push_back(r); }
        else edges[it->second] = r;
        if (r.edge_valid) consensus++;
    }
    void add(const EdgeRec& r)                                      // :62-77
    {
        from_start = std::min(r.time_from, from_start); from_end = std::max(r.time_from, from_end);
        to_start = std::min(r.time_to, to_start); to_end = std::max(r.time_to, to_end);
        changed = true;
        put(r);
    }
    void remove(uint64_t key)                                       // :89-96
    {
        auto it = where.find(key);
        if (it == where.end()) return;
        const int at = it->second;
        if (edges[at].valid) consensus--;
        edges.erase(edges.begin() + at);
        where.erase(it);
        for (auto& w : where) if (w.second > at) w.second--;
    }
    void merge(const Cluster& o)                                    // :98-107
    {
        from_start = std::min(o.from_start, from_start); from_end = std::max(o.from_end, from_end);
        to_start = std::min(o.to_start, to_start); to_end = std::max(o.to_end, to_end);
        changed = true;
        consensus += o.consensus;
        for (const EdgeRec& r : o.edges)
            if (where.find(r.key) == where.end()) { where[r.key] = (int)edges.size(); edges.push_back(r); }
    }
    bool is_part(int64_t tf, int64_t tt, double max_dt) const      // :109-115
    {
        auto sec = [](int64_t a, int64_t b) { return (double)(a - b) * 1e-9; };
        return sec(tf, from_start) > -max_dt && sec(tf, from_end) < max_dt && sec(tt, to_start) > -max_dt && sec(tt, to_end) < max_dt;
    }
};

typedef std::shared_ptr<Cluster> ClusterPtr;

}  // namespace

struct uzl_filter {
    std::mutex mu;
    std::string last_error;
    uzl_filter_cfg cfg;
    uzl_match* fte = nullptr;                                       // TransformationFilter::fte (transformation_filter.h:105)
    std::vector<ClusterPtr> clusters;                               // clusters_
    std::map<uint64_t, std::vector<ClusterPtr>> edges;              // edges_ (ordered: allEdges()/validEdges() come out sorted)
    std::vector<double> sensors;
    int n_sensors = 0;
    bool sensors_dirty = true;
    uint64_t next_uid = 0;
    // device batch
    PinBuf<FilterEdgeDev> h_edges; DevBuf<FilterEdgeDev> d_edges;
    PinBuf<int32_t> h_col; DevBuf<int32_t> d_col;
    DevBuf<double> d_sensors, d_P, d_Q;
    DevBuf<uint8_t> d_set;
    PinBuf<double> h_P, h_Q; PinBuf<uint8_t> h_set; PinBuf<uzl_edge_result> h_res;
};

namespace {

int fail(uzl_filter* h, int code, const char* msg)
{
    h->last_error = msg;
    return code;
}

EdgeRec make_rec(const uzl_filter_edge& e, int64_t tf, int64_t tt)
{
    EdgeRec r;
    r.key = e.key;
    memcpy(r.geo.pos_from, e.pose_from, 96); memcpy(r.geo.pos_to, e.pose_to, 96);
    memcpy(r.geo.transform, e.transform, 96);
    memcpy(r.geo.disp_from, e.displacement_from, 96); memcpy(r.geo.disp_to, e.displacement_to, 96);
    r.geo.sensor_from = e.sensor_from; r.geo.sensor_to = e.sensor_to;
    r.time_from = tf; r.time_to = tt;
    r.score = e.matching_score;
    r.edge_valid = e.valid != 0;
    r.valid = e.valid != 0;
    return r;
}

// TransformationFilter::add (:138-207)
void add_one(uzl_filter* h, const uzl_filter_edge& e)
{
    auto known = h->edges.find(e.key);
    if (known != h->edges.end()) {                                  // :140-146 -> EdgeCluster::updateEdge (:79-87)
        for (auto& c : known->second) {
            auto it = c->where.find(e.key);
            if (it == c->where.end()) continue;
            EdgeRec& d = c->edges[it->second];
            const EdgeRec fresh = make_rec(e, d.time_from, d.time_to);
            d.geo = fresh.geo; d.score = fresh.score; d.edge_valid = fresh.edge_valid;     // valid_ and the stamps stay
        }
        return;
    }
    for (int32_t a = 0; a < e.n_stamps_from; a++) {
        for (int32_t b = 0; b < e.n_stamps_to; b++) {
            const int64_t tf = e.stamps_from_ns[a], tt = e.stamps_to_ns[b];
            std::vector<unsigned> matched;                          // :152-159
            for (unsigned i = 0; i < h->clusters.size(); i++)
                if (h->clusters[i]->size() < h->cfg.max_cluster_size && h->clusters[i]->is_part(tf, tt, h->cfg.max_dt)) matched.push_back(i);
            const EdgeRec rec = make_rec(e, tf, tt);
            if (matched.empty()) {                                  // :162-165
                ClusterPtr c = std::make_shared<Cluster>();
                c->uid = h->next_uid++;
                c->from_start = c->from_end = tf; c->to_start = c->to_end = tt;
                c->put(rec);
                h->clusters.push_back(c);
                h->edges[e.key].push_back(c);
            } else {
                ClusterPtr c0 = h->clusters[matched[0]];
                c0->add(rec);                                       // :168
                h->edges[e.key].push_back(c0);                      // :169
                for (int i = (int)matched.size() - 1; i >= 1; i--) {                        // :172-199
                    ClusterPtr ci = h->clusters[matched[i]];
                    if (c0->size() + ci->size() < h->cfg.max_cluster_size) {
                        for (const EdgeRec& m : ci->edges) {        // :175-182: the first listing of ci is repointed
                            auto lst = h->edges.find(m.key);
                            if (lst == h->edges.end()) continue;
                            for (auto& ec : lst->second) if (ec == ci) { ec = c0; break; }
                        }
                        c0->merge(*ci);
                        h->clusters.erase(h->clusters.begin() + matched[i]);
                    }
                }
            }
        }
    }
}

// TransformationFilter::remove (:209-220)
void remove_one(uzl_filter* h, uint64_t key)
{
    auto it = h->edges.find(key);
    if (it == h->edges.end()) return;
    for (auto& c : it->second) {
        c->remove(key);
        if (c->size() == 0) h->clusters.erase(std::remove(h->clusters.begin(), h->clusters.end(), c), h->clusters.end());
    }
    h->edges.erase(it);
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

void uzl_filter_cfg_default(uzl_filter_cfg* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->max_dt = 5.0; c->min_size = 8.0; c->max_cluster_size = 100; c->ransac_iterations = 200;
    c->max_error = 0.3; c->min_time_span = 2.0; c->max_edges = 5; c->device = 0; c->seed = 0;
}

int uzl_filter_create(const uzl_filter_cfg* cfg, uzl_filter** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_filter_cfg c;
    if (cfg) c = *cfg; else uzl_filter_cfg_default(&c);
    if (c.max_cluster_size < 1 || c.ransac_iterations < 1 || c.ransac_iterations > 4096 || c.max_edges < 1 || !(c.max_error > 0.0))
        return UZL_ERR_BAD_ARG;
    uzl_filter* h = new (std::nothrow) uzl_filter();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    uzl_match_cfg mc;
    uzl_match_cfg_default(&mc);
    mc.device = c.device; mc.seed = c.seed;
    const int rc = uzl_match_create(&mc, &h->fte);                  // no GPU -> the filter cannot exist either
    if (rc != UZL_OK) { delete h; return rc; }
    *out = h;
    return UZL_OK;
}

void uzl_filter_destroy(uzl_filter* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->fte) uzl_match_destroy(h->fte);
    delete h;
}

const char* uzl_filter_last_error(uzl_filter* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_filter_set_sensors(uzl_filter* h, int32_t n_sensors, const double* sensors)
{
    UZL_GUARD_BEGIN(h)
    if (n_sensors < 0 || (n_sensors > 0 && !sensors)) return fail(h, UZL_ERR_BAD_ARG, "bad sensor table");
    h->sensors.assign(sensors, sensors + 12 * (size_t)n_sensors);
    h->n_sensors = n_sensors;
    h->sensors_dirty = true;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_add(uzl_filter* h, int32_t n_edges, const uzl_filter_edge* edges)
{
    UZL_GUARD_BEGIN(h)
    if (n_edges < 0 || (n_edges > 0 && !edges)) return fail(h, UZL_ERR_BAD_ARG, "null edges");
    for (int32_t i = 0; i < n_edges; i++) {
        const uzl_filter_edge& e = edges[i];
        if (e.n_stamps_from < 0 || e.n_stamps_to < 0 || (e.n_stamps_from > 0 && !e.stamps_from_ns) || (e.n_stamps_to > 0 && !e.stamps_to_ns))
            return fail(h, UZL_ERR_BAD_ARG, "bad stamp arrays");
    }
    for (int32_t i = 0; i < n_edges; i++) add_one(h, edges[i]);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_remove(uzl_filter* h, int32_t n_keys, const uint64_t* keys)
{
    UZL_GUARD_BEGIN(h)
    if (n_keys < 0 || (n_keys > 0 && !keys)) return fail(h, UZL_ERR_BAD_ARG, "null keys");
    for (int32_t i = 0; i < n_keys; i++) remove_one(h, keys[i]);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_all_edges(uzl_filter* h, int32_t cap, uint64_t* keys, int32_t* n)     // :343-350
{
    UZL_GUARD_BEGIN(h)
    if (!n || cap < 0 || (cap > 0 && !keys)) return fail(h, UZL_ERR_BAD_ARG, "bad output");
    int32_t k = 0;
    for (const auto& e : h->edges) { if (k < cap) keys[k] = e.first; k++; }
    *n = k;
    return UZL_OK;
    UZL_GUARD_END(h)
}

// TransformationFilter::calcValidEdges (:222-291), batched over the clusters that pass the three gates
int uzl_filter_calc_valid_edges(uzl_filter* h, int32_t* n_evaluated)
{
    UZL_GUARD_BEGIN(h)
    if (n_evaluated) *n_evaluated = 0;
    std::vector<Cluster*> todo;
    std::vector<int32_t> offsets(1, 0);
    std::vector<uint64_t> job_ids;
    auto sec = [](int64_t a, int64_t b) { return (double)(a - b) * 1e-9; };
    for (auto& c : h->clusters) {
        if ((double)c->size() < h->cfg.min_size) continue;                                  // :233
        if (!c->changed) continue;                                                          // :236
        if (std::fabs(sec(c->from_start, c->from_end)) < h->cfg.min_time_span ||
            std::fabs(sec(c->to_start, c->to_end)) < h->cfg.min_time_span) continue;        // :240-244
        todo.push_back(c.get());                                                            // `changed` is cleared (:247) once the results are back:
                                                                                            // a failed or refused batch leaves the clusters due
        offsets.push_back(offsets.back() + c->size());
        job_ids.push_back((c->uid << 20) + (uint64_t)c->evaluations);
    }
    if (todo.empty()) return UZL_OK;
    const int n_cl = (int)todo.size(), total = offsets.back();
    UZL_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = match_stream(h->fte);
    h->h_edges.reserve((size_t)total); h->d_edges.reserve((size_t)total);
    h->h_col.reserve((size_t)total); h->d_col.reserve((size_t)total);
    h->d_P.reserve((size_t)total * 3); h->d_Q.reserve((size_t)total * 3); h->d_set.reserve((size_t)total);
    h->h_P.reserve((size_t)total * 3); h->h_Q.reserve((size_t)total * 3); h->h_set.reserve((size_t)total);
    h->h_res.reserve((size_t)n_cl);
    for (int b = 0; b < n_cl; b++)
        for (int k = 0; k < todo[b]->size(); k++) {
            h->h_edges.p[offsets[b] + k] = todo[b]->edges[k].geo;
            h->h_col.p[offsets[b] + k] = b;
        }
    if (h->sensors_dirty) {
        h->d_sensors.reserve(std::max<size_t>(h->sensors.size(), 12));
        if (!h->sensors.empty())
            UZL_HIP(hipMemcpyAsync(h->d_sensors.p, h->sensors.data(), h->sensors.size() * 8, hipMemcpyHostToDevice, s));
        UZL_HIP(hipStreamSynchronize(s));                         // the source is pageable host memory
        h->sensors_dirty = false;
    }
    UZL_HIP(hipMemcpyAsync(h->d_edges.p, h->h_edges.p, (size_t)total * sizeof(FilterEdgeDev), hipMemcpyHostToDevice, s));
    UZL_HIP(hipMemcpyAsync(h->d_col.p, h->h_col.p, (size_t)total * 4, hipMemcpyHostToDevice, s));
    launch_filter_points(h->d_edges.p, total, h->d_sensors.p, h->n_sensors, h->d_P.p, h->d_Q.p, s);
    MatchDeviceResults r;
    const int rc = match_ransac_device(h->fte, n_cl, offsets.data(), h->d_P.p, h->d_Q.p, h->cfg.max_error,
                                       h->cfg.ransac_iterations, 1.0, 0, job_ids.data(), &r);     // :270-273
    if (rc != UZL_OK) { h->last_error = uzl_match_last_error(h->fte); return rc; }
    launch_filter_consensus(h->d_P.p, h->d_Q.p, h->d_col.p, total, r.results, h->cfg.max_error, h->d_set.p, s);   // :275-276
    UZL_HIP(hipGetLastError());
    UZL_HIP(hipMemcpyAsync(h->h_set.p, h->d_set.p, (size_t)total, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipMemcpyAsync(h->h_res.p, r.results, (size_t)n_cl * sizeof(uzl_edge_result), hipMemcpyDeviceToHost, s));
    UZL_HIP(hipMemcpyAsync(h->h_P.p, h->d_P.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipMemcpyAsync(h->h_Q.p, h->d_Q.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
    UZL_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < n_cl; b++) {
        Cluster* c = todo[b];
        const int m = c->size();
        const uint8_t* set = h->h_set.p + offsets[b];
        int consensus = 0;
        for (int k = 0; k < m; k++) consensus += set[k];
        c->changed = false;                                                                 // :247
        c->evaluations++;
        c->lastP.assign(h->h_P.p + 3 * (size_t)offsets[b], h->h_P.p + 3 * (size_t)offsets[b + 1]);
        c->lastQ.assign(h->h_Q.p + 3 * (size_t)offsets[b], h->h_Q.p + 3 * (size_t)offsets[b + 1]);
        memcpy(c->lastT, h->h_res.p[b].T, sizeof(c->lastT));
        c->last_ransac = h->h_res.p[b].consensus;
        if ((double)consensus >= h->cfg.min_size && consensus >= c->consensus) {            // :279-284
            c->consensus = consensus;
            for (int k = 0; k < m; k++) c->edges[k].valid = set[k] != 0;
        }
    }
    if (n_evaluated) *n_evaluated = n_cl;
    return UZL_OK;
    UZL_GUARD_END(h)
}

// TransformationFilter::validEdges (:293-337)
int uzl_filter_valid_edges(uzl_filter* h, int32_t cap, uint64_t* keys, int32_t* n)
{
    UZL_GUARD_BEGIN(h)
    if (!n || cap < 0 || (cap > 0 && !keys)) return fail(h, UZL_ERR_BAD_ARG, "bad output");
    std::vector<uint64_t> ids;
    const int max_edges = h->cfg.max_edges;
    for (const auto& c : h->clusters) {
        std::vector<const EdgeRec*> v;                                                      // :299-303
        for (const EdgeRec& r : c->edges) if (r.valid) v.push_back(&r);
        if ((int)v.size() > 2 * max_edges) {                                                // :311
            std::stable_sort(v.begin(), v.end(), [](const EdgeRec* a, const EdgeRec* b) { return a->score > b->score; });   // :313 (and :321)
            for (int i = 0; i < max_edges; i++) ids.push_back(v[i]->key);                   // :316-318
            const double increment = (double)v.size() / (double)max_edges;                  // :324
            for (int i = 0; i < max_edges - 1; i++) ids.push_back(v[(size_t)std::floor(increment * i)]->key);   // :325-327
            ids.push_back(v.back()->key);                                                   // :328
        } else {
            for (const EdgeRec* r : v) ids.push_back(r->key);                               // :331-333
        }
    }
    std::sort(ids.begin(), ids.end());                                                      // std::set
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (size_t i = 0; i < ids.size() && (int32_t)i < cap; i++) keys[i] = ids[i];
    *n = (int32_t)ids.size();
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_cluster_count(uzl_filter* h)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return (int)h->clusters.size();
}

int uzl_filter_cluster_info(uzl_filter* h, int32_t index, uzl_cluster_info* o)
{
    UZL_GUARD_BEGIN(h)
    if (!o || index < 0 || index >= (int32_t)h->clusters.size()) return fail(h, UZL_ERR_BAD_ARG, "bad cluster index");
    const Cluster& c = *h->clusters[index];
    o->uid = c.uid; o->from_start_ns = c.from_start; o->from_end_ns = c.from_end; o->to_start_ns = c.to_start; o->to_end_ns = c.to_end;
    o->size = c.size(); o->consensus = c.consensus; o->changed = c.changed ? 1 : 0; o->evaluations = c.evaluations;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_cluster_edges(uzl_filter* h, int32_t index, int32_t cap, uint64_t* keys, uint8_t* valid)
{
    UZL_GUARD_BEGIN(h)
    if (index < 0 || index >= (int32_t)h->clusters.size()) return fail(h, UZL_ERR_BAD_ARG, "bad cluster index");
    const Cluster& c = *h->clusters[index];
    if (cap < c.size() || !keys || !valid) return fail(h, UZL_ERR_BAD_ARG, "output too small");
    for (int k = 0; k < c.size(); k++) { keys[k] = c.edges[k].key; valid[k] = c.edges[k].valid ? 1 : 0; }
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_filter_cluster_last_eval(uzl_filter* h, int32_t index, int32_t cap, double* P, double* Q, double* T, int32_t* ransac_consensus)
{
    UZL_GUARD_BEGIN(h)
    if (index < 0 || index >= (int32_t)h->clusters.size()) return fail(h, UZL_ERR_BAD_ARG, "bad cluster index");
    const Cluster& c = *h->clusters[index];
    const int m = (int)c.lastP.size() / 3;
    if (cap < m) return fail(h, UZL_ERR_BAD_ARG, "output too small");
    if (m > 0 && P) memcpy(P, c.lastP.data(), (size_t)m * 24);
    if (m > 0 && Q) memcpy(Q, c.lastQ.data(), (size_t)m * 24);
    if (T) memcpy(T, c.lastT, sizeof(c.lastT));
    if (ransac_consensus) *ransac_consensus = c.last_ransac;
    return m;
    UZL_GUARD_END(h)
}

}  // extern "C"
