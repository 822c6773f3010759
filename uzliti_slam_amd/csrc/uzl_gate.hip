// uzl_gate.hip — host side of the edge acceptance gate + its C ABI (uzl_gate_*).
//
// Mirrors GraphSlamNode::newEdgeCallback (graph_slam/src/graph_slam_node.cpp:779-829).  Index bookkeeping (isMerged,
// existsEdge(from, to, type), which edges are in the graph) is host logic; thresholds, the graph search and the
// plausibility test run on the GPU, one lane per candidate, all candidates of a call in one launch.  The callback's
// sequential semantics are replayed over the results in candidate order: a candidate that duplicates an edge accepted
// earlier in the same call is dropped, and when an accepted edge is valid (score >= min_accept_valid: it changes what
// astar can reach) the remaining candidates are searched again on the updated adjacency.
#include "uzl_common.hpp"
#include "uzl_streams.hpp"
#include "gate_types.hpp"

#include <algorithm>
#include <cfloat>
#include <new>
#include <set>
#include <unordered_set>
#include <tuple>

namespace uzl {
void launch_gate(const GateArgs& a, hipStream_t s);
void launch_gate_wave(const GateWaveArgs& a, hipStream_t s);
bool launch_gate_reg(const GateWaveArgs& a, int slots, hipStream_t s);
bool launch_gate_bound(const GateWaveArgs& a, hipStream_t s);
}
using namespace uzl;

struct uzl_gate {
    std::mutex mu;
    std::string last_error;
    uzl_gate_cfg cfg;
    hipStream_t stream = nullptr;
    int32_t n = 0;
    std::vector<double> poses;
    std::vector<uint8_t> merged;
    struct E { int32_t from, to, type, valid; };
    std::vector<E> edges;                 // the graph's edges: those of the last uzl_gate_set_graph that are in range, then the accepted candidates
    std::vector<E> given;                 // the edge list of the last uzl_gate_set_graph as given (a grown graph repeats it: set_graph)
    size_t n_base = 0;                    // edges[0 .. n_base) came with set_graph
    std::vector<int32_t> nb_tmp;
    std::unordered_set<uint64_t> pair_type;                         // (min, max, type) of every edge, packed (pair_key): existsEdge(from, to, type)
    bool adj_dirty = true, poses_dirty = true;
    std::vector<int32_t> adj_ptr, adj_nbr;
    DevBuf<double> d_poses, d_gs, d_dist, d_gclosed;
    DevBuf<int32_t> d_adj_ptr, d_adj_nbr, d_over;
    DevBuf<uzl_gate_edge> d_cand;
    DevBuf<uint8_t> d_run, d_st, d_pre, d_heur;
    DevBuf<GateHeapEnt> d_heap;
    std::vector<GateNodeRec> rec;
    DevBuf<GateNodeRec> d_rec;
    DevBuf<GateState> d_gst;
    DevBuf<uint8_t> d_redo;
    DevBuf<long long> d_dbg;         // diagnostic build (UZL_GATE_DBG=1): per-search counters of gate_reg_kernel
    std::vector<long long> dbg_last;
    bool dbg_on = false;
    int reg_slots = 2;               // open-list entries per lane gate_reg_kernel starts with (4 once a search of this handle outgrew 2)
    PinBuf<uint8_t> h_redo;
    bool lane_kernel_only = false;   // A/B (diagnostic build, UZL_GATE_LANE=1): every search through gate_kernel, as in round 1
    bool wave_kernel_only = false;   // A/B (UZL_GATE_WAVE=1): gate_wave_kernel (list in LDS, per-node state in HBM) instead of gate_reg_kernel
    int64_t n_wave = 0, n_lane = 0;  // searches run by either kernel (uzl_gate_search_counts)
    int64_t n_bound = 0;             // candidates the deciding search (gate_bound_kernel) settled without the reference's search
    PinBuf<uint8_t> h_pre, h_heur;
    PinBuf<double> h_dist;
    PinBuf<int32_t> h_over;
};

namespace {

int fail(uzl_gate* h, int code, const char* msg)
{
    h->last_error = msg;
    return code;
}

// node indices are below 2^28 (a graph of that size does not fit the search's scratch anyway), edge types below 2^8
inline uint64_t pair_key(int32_t from, int32_t to, int32_t type)
{
    return ((uint64_t)(uint32_t)std::min(from, to) << 36) | ((uint64_t)(uint32_t)std::max(from, to) << 8) | (uint64_t)(uint32_t)(type & 0xff);
}
void add_edge(uzl_gate* h, int32_t from, int32_t to, int32_t type, int32_t valid)
{
    h->edges.push_back({from, to, type, valid});
    h->pair_type.insert(pair_key(from, to, type));
    if (valid && type != UZL_EDGE_TYPE_2D_LASER) h->adj_dirty = true;
}

// getNeighbors(v, only_valid = true) for every v (slam_graph.cpp:558-578): valid, non-laser edges, both directions
void build_adjacency(uzl_gate* h)
{
    const int n = h->n;
    h->adj_ptr.assign((size_t)n + 1, 0);
    for (const auto& e : h->edges) {
        if (!e.valid || e.type == UZL_EDGE_TYPE_2D_LASER) continue;
        h->adj_ptr[e.from + 1]++;
        if (e.to != e.from) h->adj_ptr[e.to + 1]++;
    }
    for (int i = 0; i < n; i++) h->adj_ptr[i + 1] += h->adj_ptr[i];
    h->adj_nbr.assign((size_t)std::max(h->adj_ptr[n], 1), 0);
    std::vector<int32_t> fill(h->adj_ptr.begin(), h->adj_ptr.end() - 1);
    for (const auto& e : h->edges) {
        if (!e.valid || e.type == UZL_EDGE_TYPE_2D_LASER) continue;
        h->adj_nbr[fill[e.from]++] = e.to;
        if (e.to != e.from) h->adj_nbr[fill[e.to]++] = e.from;
    }
    // node records of the wave-per-candidate search: position, degree, first neighbours in adjacency order
    h->rec.assign(((size_t)std::max(n, 1) + kGateBlockNodes - 1) / kGateBlockNodes * kGateBlockNodes, GateNodeRec{});     // whole blocks: gate_reg_kernel's cache loads them as units
    for (int v = 0; v < n; v++) {
        GateNodeRec& r = h->rec[v];
        r.px = h->poses[12 * (size_t)v + 3]; r.py = h->poses[12 * (size_t)v + 7]; r.pz = h->poses[12 * (size_t)v + 11];
        r.deg = h->adj_ptr[v + 1] - h->adj_ptr[v]; r.adj = h->adj_ptr[v];
        for (int j = 0; j < kGateRecNbr; j++) r.nbr[j] = j < r.deg ? h->adj_nbr[h->adj_ptr[v] + j] : -1;
        {   // multi-edges: flagged, so that a search only looks for repeated neighbours where there are any
            const int32_t* nb = h->adj_nbr.data() + h->adj_ptr[v];
            const int d = h->adj_ptr[v + 1] - h->adj_ptr[v];
            bool multi = false;
            if (d <= 16) { for (int a = 1; a < d && !multi; a++) for (int b = 0; b < a; b++) if (nb[a] == nb[b]) { multi = true; break; } }
            else { h->nb_tmp.assign(nb, nb + d); std::sort(h->nb_tmp.begin(), h->nb_tmp.end()); multi = std::adjacent_find(h->nb_tmp.begin(), h->nb_tmp.end()) != h->nb_tmp.end(); }
            if (multi) r.deg |= kGateRecMulti;
        }
    }
    h->adj_dirty = false;
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

void uzl_gate_cfg_default(uzl_gate_cfg* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->min_matching_score = 20.0; c->max_edge_distance_T = 1.0; c->max_edge_distance_R = 20.0;
    c->scope_size_factor = 0.1; c->min_accept_valid = DBL_MAX; c->device = 0;
}

int uzl_gate_create(const uzl_gate_cfg* cfg, uzl_gate** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_gate_cfg c;
    if (cfg) c = *cfg; else uzl_gate_cfg_default(&c);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return UZL_ERR_NO_DEVICE;     // no CPU fallback
    if (c.device < 0 || c.device >= count) return UZL_ERR_NO_DEVICE;
    uzl_gate* h = new (std::nothrow) uzl_gate();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    h->lane_kernel_only = diag_flag("UZL_GATE_LANE");
    h->wave_kernel_only = diag_flag("UZL_GATE_WAVE");
    h->dbg_on = diag_flag("UZL_GATE_DBG");
    if (hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return UZL_ERR_HIP;
    }
    stream_register(c.device, h->stream, false);
    *out = h;
    return UZL_OK;
}

void uzl_gate_destroy(uzl_gate* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); stream_unregister(h->cfg.device, h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

const char* uzl_gate_last_error(uzl_gate* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_gate_set_graph(uzl_gate* h, int32_t n_nodes, const double* poses, const uint8_t* merged, int32_t n_edges,
                       const uzl_gate_edge* edges)
{
    UZL_GUARD_BEGIN(h)
    if (n_nodes < 0 || n_edges < 0 || (n_nodes > 0 && !poses) || (n_edges > 0 && !edges)) return fail(h, UZL_ERR_BAD_ARG, "null arrays");
    // An online session calls this once per re-optimisation interval with a graph that has only GROWN (graph_slam_node.cpp:779-829 sees the
    // SlamGraph of the moment; edge ids are time-ordered, :294): the edges of the last call come first again, at most their `valid` flags
    // differ, new edges follow.  Then only the tail is entered (the existsEdge index was ~1 ms of hash inserts per call at 20k nodes,
    // config 5) - the resulting state is the one the full rebuild below leaves: same edge list in the same order, so the same
    // adjacency order and the same searches.  Edges uzl_gate_check accepted since the last call are not part of the caller's graph
    // unless they come back in `edges`.
    bool grown = h->given.size() <= (size_t)n_edges && n_nodes >= h->n && !h->given.empty();
    for (size_t k = 0; grown && k < h->given.size(); k++) {
        const uzl_gate::E& g = h->given[k];
        grown = g.from == edges[k].from && g.to == edges[k].to && g.type == edges[k].type;
    }
    // (an edge that was out of range with fewer nodes and is in range now would have to be inserted in the middle of the list)
    for (size_t k = 0; grown && k < h->given.size(); k++) {
        const uzl_gate::E& g = h->given[k];
        const bool was_in = g.from >= 0 && g.to >= 0 && g.from < h->n && g.to < h->n, is_in = g.from >= 0 && g.to >= 0 && g.from < n_nodes && g.to < n_nodes;
        grown = was_in == is_in;
    }
    if (grown) {
        for (size_t k = h->n_base; k < h->edges.size(); k++) h->pair_type.erase(pair_key(h->edges[k].from, h->edges[k].to, h->edges[k].type));      // accepted candidates
        h->edges.resize(h->n_base);
        size_t own = 0;
        for (size_t k = 0; k < h->given.size(); k++) {
            uzl_gate::E& g = h->given[k];
            if (g.from < 0 || g.to < 0 || g.from >= h->n || g.to >= h->n) continue;
            const int32_t v = edges[k].valid ? 1 : 0;
            if (h->edges[own].valid != v) { h->edges[own].valid = v; g.valid = v; }
            own++;
        }
    } else {
        h->edges.clear(); h->pair_type.clear(); h->given.clear();
        h->edges.reserve((size_t)n_edges + 64); h->pair_type.reserve((size_t)n_edges * 2 + 64);
    }
    h->given.reserve((size_t)n_edges);
    for (int32_t k = (int32_t)h->given.size(); k < n_edges; k++) {
        const uzl_gate_edge& e = edges[k];
        h->given.push_back({e.from, e.to, e.type, e.valid ? 1 : 0});
        if (e.from < 0 || e.to < 0 || e.from >= n_nodes || e.to >= n_nodes) continue;
        add_edge(h, e.from, e.to, e.type, e.valid ? 1 : 0);
    }
    h->n_base = h->edges.size();
    h->n = n_nodes;
    h->poses.assign(poses, poses + 12 * (size_t)n_nodes);
    h->merged.assign((size_t)n_nodes, 0);
    if (merged) h->merged.assign(merged, merged + n_nodes);
    h->adj_dirty = true; h->poses_dirty = true;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_gate_check(uzl_gate* h, int32_t nc, const uzl_gate_edge* cand, uint8_t* accept, uint8_t* valid, double* astar_dist)
{
    UZL_GUARD_BEGIN(h)
    if (nc < 0 || (nc > 0 && (!cand || !accept))) return fail(h, UZL_ERR_BAD_ARG, "null arrays");
    for (int32_t k = 0; k < nc; k++) { accept[k] = 0; if (valid) valid[k] = 0; if (astar_dist) astar_dist[k] = -1.; }
    if (nc == 0) return UZL_OK;
    UZL_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;
    const int n = h->n;
    if (h->poses_dirty) {
        h->d_poses.reserve(std::max<size_t>(h->poses.size(), 12));
        if (n) UZL_HIP(hipMemcpyAsync(h->d_poses.p, h->poses.data(), h->poses.size() * 8, hipMemcpyHostToDevice, s));
        UZL_HIP(hipStreamSynchronize(s));
        h->poses_dirty = false;
    }
    h->d_cand.reserve((size_t)nc); h->d_run.reserve((size_t)nc);
    h->d_pre.reserve((size_t)nc); h->d_heur.reserve((size_t)nc); h->d_dist.reserve((size_t)nc); h->d_over.reserve(1);
    h->h_pre.reserve((size_t)nc); h->h_heur.reserve((size_t)nc); h->h_dist.reserve((size_t)nc); h->h_over.reserve(1);
    UZL_HIP(hipMemcpyAsync(h->d_cand.p, cand, sizeof(uzl_gate_edge) * (size_t)nc, hipMemcpyHostToDevice, s));
    std::vector<uint8_t> run((size_t)nc), srch((size_t)nc);
    int32_t first = 0;                                        // candidates before `first` are decided
    constexpr int kChunk = 256;                               // searches per launch (scratch = chunk x (9 n + 16 heap_cap) bytes)
    while (first < nc) {
        if (h->adj_dirty) {
            build_adjacency(h);
            h->d_adj_ptr.reserve(h->adj_ptr.size()); h->d_adj_nbr.reserve(h->adj_nbr.size());
            UZL_HIP(hipMemcpyAsync(h->d_adj_ptr.p, h->adj_ptr.data(), h->adj_ptr.size() * 4, hipMemcpyHostToDevice, s));
            UZL_HIP(hipMemcpyAsync(h->d_adj_nbr.p, h->adj_nbr.data(), h->adj_nbr.size() * 4, hipMemcpyHostToDevice, s));
            h->d_rec.reserve(h->rec.size());
            UZL_HIP(hipMemcpyAsync(h->d_rec.p, h->rec.data(), h->rec.size() * sizeof(GateNodeRec), hipMemcpyHostToDevice, s));
            UZL_HIP(hipStreamSynchronize(s));
        }
        const int32_t last = std::min(nc, first + kChunk);
        const int32_t m = last - first;
        // index checks against the graph as it is now (:784-791); duplicates inside the chunk are caught in the replay
        for (int32_t k = first; k < last; k++) {
            const uzl_gate_edge& c = cand[k];
            bool ok = c.from >= 0 && c.to >= 0 && c.from < n && c.to < n;
            if (ok) ok = !h->merged[c.from] && !h->merged[c.to];
            if (ok) ok = h->pair_type.count(pair_key(c.from, c.to, c.type)) == 0;
            run[k] = ok ? 1 : 0;
        }
        UZL_HIP(hipMemcpyAsync(h->d_run.p + first, run.data() + first, (size_t)m, hipMemcpyHostToDevice, s));
        // ---- one wave per candidate, open list in LDS
        h->d_redo.reserve((size_t)nc); h->h_redo.reserve((size_t)nc);
        const bool lds_path = !h->wave_kernel_only && gate_lds_bytes(n) <= kGateLdsMax;
        if (lds_path) h->d_gclosed.reserve((size_t)m * std::max(n, 1));                       // written before it is read: no clearing
        else {
            h->d_gst.reserve((size_t)m * std::max(n, 1));
            UZL_HIP(hipMemsetAsync(h->d_gst.p, 0, sizeof(GateState) * (size_t)m * std::max(n, 1), s));
        }
        bool any_left = true;
        for (int32_t k = first; k < last; k++) srch[k] = run[k];            // which candidates the search stages still have to run
        GateWaveArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.n = n; wa.n_query = m; wa.poses = h->d_poses.p; wa.rec = h->d_rec.p; wa.adj_nbr = h->d_adj_nbr.p;
        wa.cand = h->d_cand.p + first; wa.run = h->d_run.p + first; wa.gst = h->d_gst.p; wa.gclosed = h->d_gclosed.p;
        wa.min_score = h->cfg.min_matching_score; wa.max_T = h->cfg.max_edge_distance_T; wa.max_R = h->cfg.max_edge_distance_R;
        wa.ssf = h->cfg.scope_size_factor;
        wa.skip_decided = astar_dist ? 0 : 1;             // nobody asked for the path lengths: searches whose verdict is known are skipped
        wa.pre_ok = h->d_pre.p + first; wa.heur_ok = h->d_heur.p + first; wa.dist = h->d_dist.p + first; wa.redo = h->d_redo.p + first;
        if (h->dbg_on) { h->d_dbg.reserve((size_t)m * 8); UZL_HIP(hipMemsetAsync(h->d_dbg.p, 0, sizeof(long long) * 8 * (size_t)m, s)); wa.dbg = h->d_dbg.p; }
        // Nobody asked for the path lengths: a search that only decides first (gate_bound_kernel).  What it cannot decide goes on to the
        // reference's search below, the rest keeps its verdict (keep_unrun).
        if (wa.skip_decided && lds_path && !h->lane_kernel_only && launch_gate_bound(wa, s)) {
            UZL_HIP(hipGetLastError());
            UZL_HIP(hipMemcpyAsync(h->h_redo.p + first, h->d_redo.p + first, (size_t)m, hipMemcpyDeviceToHost, s));
            UZL_HIP(hipStreamSynchronize(s));
            bool any = false;
            for (int32_t k = first; k < last; k++) {
                const bool undecided = run[k] && h->h_redo.p[k] == 2;
                if (run[k] && !undecided) h->n_bound++;
                srch[k] = undecided ? 1 : 0; any = any || undecided;
                h->h_redo.p[k] = 0;
            }
            UZL_HIP(hipMemcpyAsync(h->d_run.p + first, srch.data() + first, (size_t)m, hipMemcpyHostToDevice, s));
            UZL_HIP(hipStreamSynchronize(s));
            wa.keep_unrun = 1; any_left = any;
        }
        if (!h->lane_kernel_only && any_left) {
            bool reg_ok = lds_path && launch_gate_reg(wa, h->reg_slots, s);
            if (!reg_ok) {
                if (lds_path) {                                                                // the attribute could not be raised: the HBM-state kernel
                    h->d_gst.reserve((size_t)m * std::max(n, 1));
                    UZL_HIP(hipMemsetAsync(h->d_gst.p, 0, sizeof(GateState) * (size_t)m * std::max(n, 1), s));
                    wa.gst = h->d_gst.p;
                }
                launch_gate_wave(wa, s);
            }
            UZL_HIP(hipGetLastError());
            UZL_HIP(hipMemcpyAsync(h->h_redo.p + first, h->d_redo.p + first, (size_t)m, hipMemcpyDeviceToHost, s));
            UZL_HIP(hipStreamSynchronize(s));
            if (h->dbg_on) { h->dbg_last.assign((size_t)m * 8, 0); UZL_HIP(hipMemcpy(h->dbg_last.data(), h->d_dbg.p, sizeof(long long) * 8 * (size_t)m, hipMemcpyDeviceToHost)); }
            bool overflowed = false;
            for (int32_t k = first; k < last && !overflowed; k++) overflowed = h->h_redo.p[k] != 0;
            if (reg_ok && overflowed && h->reg_slots < 4) {
                // searches whose open list outgrew two entries per lane: once more with four (and later calls start there)
                h->reg_slots = 4;
                std::vector<uint8_t> run4((size_t)m);
                for (int32_t k = 0; k < m; k++) run4[k] = (srch[first + k] && h->h_redo.p[first + k]) ? 1 : 0;
                UZL_HIP(hipMemcpyAsync(h->d_run.p + first, run4.data(), (size_t)m, hipMemcpyHostToDevice, s));
                wa.keep_unrun = 1;
                launch_gate_reg(wa, 4, s);
                UZL_HIP(hipGetLastError());
                std::vector<uint8_t> redo4((size_t)m);
                UZL_HIP(hipMemcpyAsync(redo4.data(), h->d_redo.p + first, (size_t)m, hipMemcpyDeviceToHost, s));
                UZL_HIP(hipStreamSynchronize(s));                                              // (also: run4 is a local)
                for (int32_t k = 0; k < m; k++) h->h_redo.p[first + k] = run4[k] ? redo4[k] : 0;
            }
        }
        bool any_redo = h->lane_kernel_only;
        for (int32_t k = first; k < last && !any_redo; k++) any_redo = h->h_redo.p[k] != 0;
        for (int32_t k = first; k < last; k++) {
            if (!srch[k]) continue;
            if (h->lane_kernel_only || h->h_redo.p[k]) h->n_lane++; else h->n_wave++;
        }
        if (any_redo) {
            // candidates whose open list outgrew LDS (or all of them under the A/B switch): the lane kernel with its heap in HBM.
            // Its outputs overwrite pre_ok / heur_ok / dist of every candidate it runs for.
            std::vector<uint8_t> run2((size_t)m);
            for (int32_t k = 0; k < m; k++) run2[k] = (srch[first + k] && (h->lane_kernel_only || h->h_redo.p[first + k])) ? 1 : 0;
            UZL_HIP(hipMemcpyAsync(h->d_run.p + first, run2.data(), (size_t)m, hipMemcpyHostToDevice, s));
            UZL_HIP(hipStreamSynchronize(s));
        }
        const int heap_cap = (int)std::min<size_t>((size_t)h->adj_nbr.size() + (size_t)n + 1024, (size_t)1 << 28);
        if (any_redo) {
            h->d_gs.reserve((size_t)m * std::max(n, 1)); h->d_st.reserve((size_t)m * std::max(n, 1));
            h->d_heap.reserve((size_t)m * heap_cap);
            UZL_HIP(hipMemsetAsync(h->d_st.p, 0, (size_t)m * std::max(n, 1), s));
        }
        UZL_HIP(hipMemsetAsync(h->d_over.p, 0, 4, s));
        GateArgs a;
        memset(&a, 0, sizeof(a));
        a.n = n; a.n_query = m; a.poses = h->d_poses.p; a.adj_ptr = h->d_adj_ptr.p; a.adj_nbr = h->d_adj_nbr.p;
        a.cand = h->d_cand.p + first; a.run = h->d_run.p + first;
        a.gs = h->d_gs.p; a.st = h->d_st.p; a.heap = h->d_heap.p; a.heap_cap = heap_cap;
        a.min_score = h->cfg.min_matching_score; a.max_T = h->cfg.max_edge_distance_T; a.max_R = h->cfg.max_edge_distance_R;
        a.ssf = h->cfg.scope_size_factor;
        a.skip_decided = astar_dist ? 0 : 1;
        a.pre_ok = h->d_pre.p + first; a.heur_ok = h->d_heur.p + first; a.dist = h->d_dist.p + first; a.overflow = h->d_over.p;
        a.keep_unrun = h->lane_kernel_only ? 0 : 1;
        if (any_redo) launch_gate(a, s);
        UZL_HIP(hipGetLastError());
        UZL_HIP(hipMemcpyAsync(h->h_pre.p + first, h->d_pre.p + first, (size_t)m, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_heur.p + first, h->d_heur.p + first, (size_t)m, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_dist.p + first, h->d_dist.p + first, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipMemcpyAsync(h->h_over.p, h->d_over.p, 4, hipMemcpyDeviceToHost, s));
        UZL_HIP(hipStreamSynchronize(s));
        if (h->h_over.p[0]) return fail(h, UZL_ERR_STATE, "graph search ran out of heap space");
        // replay newEdgeCallback in candidate order over the search results
        int32_t k = first;
        for (; k < last; k++) {
            const uzl_gate_edge& c = cand[k];
            if (astar_dist) astar_dist[k] = -1.;
            if (!run[k]) continue;
            const uint64_t key = pair_key(c.from, c.to, c.type);
            if (h->pair_type.count(key)) continue;                                  // joined the graph earlier in this call
            if (!h->h_pre.p[k]) continue;
            if (astar_dist) astar_dist[k] = h->h_dist.p[k];
            if (!h->h_heur.p[k]) continue;
            const int v = c.matching_score >= h->cfg.min_accept_valid ? 1 : 0;      // :809-811
            add_edge(h, c.from, c.to, c.type, v);                                   // :812
            accept[k] = 1; if (valid) valid[k] = (uint8_t)v;
            if (h->adj_dirty) { k++; break; }                                       // reachability changed: search the rest again
        }
        first = k;
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

// parity tests: searches run so far by the wave kernel and by the lane kernel
int uzl_gate_search_counts(uzl_gate* h, int64_t* n_wave, int64_t* n_lane)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_wave) *n_wave = h->n_wave;
    if (n_lane) *n_lane = h->n_lane;
    return UZL_OK;
}

// diagnostic build only (not part of include/uzl_mi355x.h): counters of the last launch of gate_reg_kernel (8 per search: expansions,
// shader clocks, 100 MHz ticks, largest list, shader clocks in the pop / the popped node's loads / the neighbours up to the push decision /
// the pushes)
#ifdef UZL_DIAG
UZL_DIAG_EXPORT int uzl_debug_gate_profile(uzl_gate* h, long long* out, int32_t cap)
{
    if (!h || !out) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const int32_t nq = (int32_t)std::min<size_t>(h->dbg_last.size() / 8, (size_t)std::max(cap, 0));
    for (int32_t i = 0; i < 8 * nq; i++) out[i] = h->dbg_last[i];
    return nq;
}

#endif

int uzl_gate_edge_count(uzl_gate* h)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return (int)h->edges.size();
}

}  // extern "C"
