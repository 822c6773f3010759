// uzl_pgo.hip — host side of the pose-graph half: graph flattening bookkeeping (skip rules), gauge
// fixing, block-CSR structure, the Levenberg-Marquardt driver and the C ABI (uzl_pgo_*).
// Mirrors G2oOptimizer (graph_optimization/src/g2o_optimizer.cpp:55-349): add_graph = addGraphImpl,
// optimize = optimizeImpl (initializeOptimization + setFixedNodes + optimize(iterations)),
// store = storeImpl.  All arithmetic on poses, measurements and the normal equations runs in the HIP
// kernels of pgo_kernels.hip; the host only makes the integer/structural decisions and the scalar LM
// accept/reject logic of g2o's OptimizationAlgorithmLevenberg [EXT].
#include "pgo_handle.hpp"
#include "pgo_lm.hpp"
#include "uzl_streams.hpp"

using namespace uzl;

// ---- RCCL through dlopen: the collective library is only loaded by processes that shard a graph -----------------------------
#include <dlfcn.h>
#include <exception>
#include <system_error>
#include <thread>
namespace {
struct RcclApi {
    typedef struct { char internal[UZL_RCCL_UNIQUE_ID_BYTES]; } UniqueId;     // ncclUniqueId (rccl.h:43)
    typedef void* Comm;                                                        // ncclComm_t
    int (*GetUniqueId)(UniqueId*) = nullptr;                                   // ncclResult_t: 0 = ncclSuccess
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int kDouble = 8, kSum = 0;                                // ncclDouble (rccl.h:467), ncclSum (rccl.h:448)
    bool ok = false;
    std::string error;
};
RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!so) { api.error = std::string("dlopen(librccl.so) failed: ") + dlerror(); return; }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(so, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(so, "ncclCommCount"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(so, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
        if (!api.ok) api.error = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
    });
    return api;
}
}  // namespace

namespace uzl {
const int kUpperNs = 4;                   // Newton-Schulz steps of the dense levels above the composite level (even: the result ends in Ydense[l])
const bool kAlwaysRefresh = diag_flag("UZL_ML_ALWAYS_REFRESH");             // A/B switch
const double kRefreshRel = diag_double("UZL_ML_REFRESH_REL", 1e-3);
// ... and where the rebuild is synchronous (large loopy graphs: its GEMMs are 1.4 ms at 10k / 50k, 7 ms at 20k / 100k, in front of the
// solve): the chi2 rule only while the problem still changes wholesale, the PCG-rate rule (kRateDrop, pgo_lm.hpp) from then on
const double kRefreshRelSync = diag_double("UZL_ML_REFRESH_REL_SYNC", 3e-2);
// ... and its rate rule: with the dense operator of that class a rebuild pays as soon as the rate has fallen to 0.75 of the fresh one
// (0.6 elsewhere; tests/diag/r5_ratedrop.sh: -3 ... -8 % on eight of nine such shapes, the 30k / 150k graph - no dense operator - +14 %)
const double kRateDropSyncDense = diag_double("UZL_ML_RATE_DROP_SYNC", 0.75);
// Newton-Schulz steps of a synchronous set-up at LM iteration `it` for a structure that asks for `structure_steps` (4 on large loopy
// graphs): 2 in the first kNsEarlyIts iterations (tests/diag/knob_sweep.sh "UZL_ML_NS_EARLY_ITS=0" against the default: -3.4 % over fourteen
// large shapes at the same PCG iteration count; 4 iterations instead of 2 gain on the largest and lose at 10k, 8 lose everywhere; NO step at
// all in those two - the cycle's operator as it comes, UZL_ML_NS_EARLY_STEPS=0 - another -2.5 %, not taken: that operator is what the
// residual guard exists for)
int ml_ns_steps_at(int structure_steps, int it)
{
    static const int early_its = diag_int("UZL_ML_NS_EARLY_ITS", 2), early_steps = diag_int("UZL_ML_NS_EARLY_STEPS", 2);
    return (structure_steps > 2 && it < early_its) ? early_steps : structure_steps;
}
double ml_rate_drop(const uzl_pgo* h) { return (!ml_async_level(h) && h->ml_comp) ? kRateDropSyncDense : kRateDrop; }
const double kLambdaRetake = diag_double("UZL_LAMBDA_RETAKE", 32.);         // lambda grown by this factor since the inverses were taken: take them again
const int kGraphPairs = std::max(1, diag_int("UZL_GRAPH_PAIRS", 8));        // one graph replay = 2 x pairs PCG iterations
int pgo_fail(uzl_pgo* h, int code, const char* msg)
{
    h->last_error = msg;
    return code;
}
}  // namespace uzl

namespace {

static const bool& always_refresh = kAlwaysRefresh;
static const double& refresh_rel = kRefreshRel;
inline int fail(uzl_pgo* h, int code, const char* msg) { return pgo_fail(h, code, msg); }

struct Timed {
    uzl_pgo* h;
    Timed(uzl_pgo* h_, const char* name) : h(h_) { h->timer.begin(name, h->stream); }
    ~Timed() { h->timer.end(h->stream); }
};

// Device scalars -> host.  A one-workgroup kernel at the end of the queued work writes them straight into pinned
// coherent host memory and bumps a sequence word; the host spins on that word (bounded, then falls back to a stream
// synchronise so that a failed launch surfaces as an error instead of a hang).
void fetch_scal(uzl_pgo* h)
{
    const uint32_t seq = ++h->pub_seq;
    k_publish(h->D, h->d_pub, seq, h->stream);
    volatile PgoHostScal* pub = h->h_scal.p;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (int spin = 0; !seen; spin++) {
        seen = __atomic_load_n(&h->h_scal.p->seq, __ATOMIC_ACQUIRE) == seq;
        if (!seen && (spin & 1023) == 1023 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 50.0) break;
    }
    (void)pub;
    if (!seen || h->timer.on) {
        UZL_HIP(hipStreamSynchronize(h->stream));
        if (__atomic_load_n(&h->h_scal.p->seq, __ATOMIC_ACQUIRE) != seq) throw HipError{hipErrorUnknown, "publish_kernel did not run", __FILE__, __LINE__};
    }
    h->timer.resolve();
}

// When a linear solve stops.  What the parity bar constrains is the pose, i.e. the error e = dx - dx* of each LM step in metres and
// radians - not a residual norm: a relative residual test solves a 1e-7 m step of a converged LM iteration to the same twelve digits
// as the metre-sized first one (config 5's last solve spent 5000 of its 6660 PCG iterations on steps below 1e-6 m), and on badly
// conditioned graphs it still lets too much error through in the soft modes (round 2 carried two corrective heuristics for that:
// a 10x tighter tolerance while chi2 still moved, and a tolerance shrinking with the previous solve's iteration count).  Both are
// replaced by an a-posteriori estimate of e itself, taken every kProgressEvery iterations from how far x still moves (progress_decide_ml,
// pgo_device.hpp: inside the iteration kernels), against an absolute target derived from cfg.pcg_tol:
//     largest translation component of e  <=  kStepT * pcg_tol  [m]      (default 1e-5: 1e-5 m   = 1/100 of the 1e-3 m bar per LM step)
//     largest rotation component of e     <=  kStepR * pcg_tol  [q_xyz]  (default       1e-6     ~ 2e-6 rad = 1/50 of the 1e-4 rad bar)
// LM is self-correcting, so the per-step errors do not add up coherently; twenty of them stay an order of magnitude inside the bar.
// The relative test on r.M^-1 r remains as a floor two orders below pcg_tol (kTolFloor2 on its square): it ends solves whose target is
// below what the arithmetic can settle.
// (kStepT, kStepR, kTolFloor2: pgo_lm.hpp)
constexpr int kProgressEveryBJ = 8;           // block-Jacobi path: PCG iterations between two looks (a launch of their own; the multilevel path looks every
                                              // kProgressEvery iterations inside its kernels, pgo_types.hpp)
inline double tol_factor2(const uzl_pgo_cfg& c) { return pgo_tol_f2(c); }
// How stale is a kept preconditioner?  Not the iteration count of the last solve (with an absolute stop test that follows the size of
// the LM step and lambda, not the operator): the CONTRACTION it delivers, nats of r.M^-1 r per PCG iteration.  rz_stop = scal[1] is
// pcg_tol^2 * tol_f2 * (r_0.M^-1 r_0).  A rebuild is due when the rate has fallen below kRateDrop of what the operator delivered when fresh.
// (kRateDrop, lm_pcg_rate: pgo_lm.hpp - shared with the device-resident loop)
inline double pcg_rate(double rz_stop, double rz_end, int its, double tol2, double tol_f2) { return lm_pcg_rate(rz_stop, rz_end, its, tol2, tol_f2); }

void set_lambda(uzl_pgo* h, double lambda, double tol_f2)
{
    k_set_trial(h->D.scal, lambda, tol_f2, pgo_eps_t(h->cfg), pgo_eps_r(h->cfg), h->stream);
    h->lambda_now = lambda;
}

// exchange step of the sharded solve: sum `count` doubles at dev_ptr over all ranks (caller-supplied RCCL all-reduce)
void shard_allreduce(uzl_pgo* h, double* ptr, int64_t count)
{
    if (!h->sharded || count <= 0) return;
    const auto t0 = std::chrono::steady_clock::now();
    // native: stream-ordered between the producing and the consuming kernel, the host does not wait
    const int rc = h->rccl_comm ? rccl().AllReduce(ptr, ptr, (size_t)count, RcclApi::kDouble, RcclApi::kSum, h->rccl_comm, h->stream)
                                : h->allreduce(ptr, count, (void*)h->stream, h->allreduce_user);
    h->exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->exchange_calls++;
    if (rc != 0) throw HipError{hipErrorUnknown, h->rccl_comm ? "ncclAllReduce failed" : "all-reduce callback failed", __FILE__, __LINE__};
}
// chi2 is a sum over edges: partial per rank
void shard_allreduce_chi2(uzl_pgo* h)
{
    if (!h->sharded) return;
    h->d_red.reserve(2);
    UZL_HIP(hipMemcpyAsync(h->d_red.p, h->D.scal + 4, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    shard_allreduce(h, h->d_red.p, 1);
    UZL_HIP(hipMemcpyAsync(h->D.scal + 4, h->d_red.p, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
}

// allocate everything that depends on (n, e) only
void alloc_problem(uzl_pgo* h, bool keep_poses = false)
{
    const size_t n = std::max(h->n, 1), e = std::max(h->e, 1);
    const bool cur_b = keep_poses && h->cur != nullptr && h->cur == h->pose_b.p;       // (the estimate sits in whichever buffer the last solve left it)
    h->pose_a.reserve(n * 8, keep_poses, h->stream); h->pose_b.reserve(n * 8, keep_poses, h->stream); h->pose_init.reserve(n * 8);
    h->d_zinv.reserve(e * 7); h->d_info.reserve(e * 36); h->d_robust.reserve(e);
    h->d_ei.reserve(e); h->d_ej.reserve(e);
    h->d_v2b.reserve(n);
    h->d_part_a.reserve(kMaxPartials); h->d_part_b.reserve(kMaxPartials); h->d_part_c.reserve(2 * kMaxPartials);     // (part_c: two maxima per ml_cg workgroup at a look)
    h->d_scal.reserve(16); h->d_flags.reserve(4);
    h->h_scal.reserve(1, hipHostMallocMapped | hipHostMallocCoherent); h->h_lambda.reserve(1);
    memset(h->h_scal.p, 0, sizeof(PgoHostScal));
    UZL_HIP(hipHostGetDevicePointer((void**)&h->d_pub, h->h_scal.p, 0));
    h->cur = cur_b ? h->pose_b.p : h->pose_a.p; h->trial = cur_b ? h->pose_a.p : h->pose_b.p;
}

// the skip rules of addGraphImpl over uzl_pgo::in_edges: odometry edges are added while iterating (:78-79), filtered feature edges after (:100-103)
void flatten_edges(uzl_pgo* h)
{
    h->ij.clear(); h->src.clear(); h->robust.clear(); h->edge_w.clear();
    const int n_nodes = h->n, n_edges = (int)h->in_edges.size();
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < n_edges; k++) {
            const uzl_pgo::InEdge& ed = h->in_edges[k];
            if (ed.from < 0 || ed.to < 0 || ed.from >= n_nodes || ed.to >= n_nodes) continue;     // :77, :194-201, :263-268
            if (ed.from == ed.to) continue;                                                        // degenerate self edge
            if ((pass == 0) != (ed.odom != 0)) continue;
            if (ed.odom) {
                if (h->fixed_in[ed.from] && !h->fixed_in[ed.to]) continue;                         // :203-206
            } else {
                if (!ed.valid) continue;                                                           // not in validEdges() (:98)
                if (h->fixed_in[ed.from] && h->fixed_in[ed.to]) continue;                          // :270-274
            }
            h->ij.push_back(ed.from); h->ij.push_back(ed.to);
            h->src.push_back(k);
            h->edge_w.push_back(ed.w);
            h->robust.push_back(ed.odom ? 0 : 1);                                                  // Huber on feature edges (:292-294)
        }
    }
    h->e = (int32_t)h->src.size();
}
uzl_pgo::InEdge in_edge_of(const uzl_edge& ed)
{
    double tr = 0.; for (int r = 0; r < 6; r++) tr += ed.information[r * 7];
    return {ed.from, ed.to, (uint8_t)(ed.type == UZL_EDGE_TYPE_2D_WHEEL_ODOMETRY ? 1 : 0), (uint8_t)(ed.valid ? 1 : 0), tr};
}

void upload_edges_common(uzl_pgo* h)
{
    hipStream_t s = h->stream;
    const int e = h->e;
    h->srec_stale = true;               // (values, not structure: the slot records follow in prepare_optimize, also when the structure is kept)
    if (e > 0) {
        std::vector<int32_t> ei((size_t)e), ej((size_t)e);
        for (int k = 0; k < e; k++) { ei[k] = h->ij[2 * k]; ej[k] = h->ij[2 * k + 1]; }
        UZL_HIP(hipMemcpyAsync(h->d_ei.p, ei.data(), sizeof(int32_t) * e, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_ej.p, ej.data(), sizeof(int32_t) * e, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_robust.p, h->robust.data(), (size_t)e, hipMemcpyHostToDevice, s));
        UZL_HIP(hipStreamSynchronize(s));   // ei/ej are stack vectors
    }
}

// G2: setFixedNodes (g2o_optimizer.cpp:301-349): per connected component (over the system edges) without a
// fixed vertex, fix the vertex with the smallest index (= lexicographically smallest node id, :338).
int32_t uf_find(std::vector<int32_t>& p, int32_t x)
{
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}
}  // namespace
namespace uzl {
int32_t gauge_fix(uzl_pgo* h)
{
    const int n = h->n;
    std::vector<int32_t> p((size_t)n);
    std::iota(p.begin(), p.end(), 0);
    for (int k = 0; k < h->e; k++) {
        int32_t a = uf_find(p, h->ij[2 * k]), b = uf_find(p, h->ij[2 * k + 1]);
        if (a != b) { if (a < b) p[b] = a; else p[a] = b; }
    }
    std::vector<uint8_t> has((size_t)n, 0);
    for (int v = 0; v < n; v++) if (h->fixed_eff[v]) has[uf_find(p, v)] = 1;
    int32_t cnt = 0;
    for (int v = 0; v < n; v++) {
        const int32_t r = uf_find(p, v);
        if (!has[r]) { h->fixed_eff[r] = 1; has[r] = 1; cnt++; }
    }
    return cnt;
}
}  // namespace uzl
namespace {

// block-CSR structure over the free vertices: one slot per (free endpoint, system edge)

// Aggregation hierarchy of the multilevel preconditioner (pgo_types.hpp): symbolic part, once per structure.
// (nb, nslots, d_*: the block system the PCG solves - the full one or the Schur-reduced one)
void build_ml(uzl_pgo* h, const std::vector<int32_t>& row_ptr0, const std::vector<int32_t>& col0, int nb, int nslots,
              const int32_t* d_row_ptr, const int32_t* d_col, double* d_blk, double* d_hdiag)
{
    h->ml_levels = 0; h->ml_n.assign(1, nb); h->ml_nslots.assign(1, nslots); h->ml_chunks.assign(1, 0); h->ml_inner_aggs = 0;
    if (h->cfg.preconditioner == 0 || nb <= kMlTopMax) return;
    // Up to here the level-1 dense operator applies (6 n_1 <= 3072: ml_cg_comp_kernel<16>); above, AGG = 4 with the level-2 one.  Its
    // rebuild (Newton-Schulz GEMMs, n^3) outgrows what the exact level-1 solve saves in PCG iterations between 3000 and 4000 vertices on
    // loopy graphs (>= 3 edges per vertex: 3000/12000 20.6 -> 18.3 ms, 4000/16000 24.5 -> 26.3 ms) and later on sparse ones - the shape of a
    // Schur-reduced online graph (4000/6000 26.0 -> 17.2 ms; config 5's last solve 2328 -> 1288 PCG iterations).
    static const int agg1_env = diag_int("UZL_ML_AGG1_MAX", 0);
    const int agg1_max = agg1_env > 0 ? agg1_env : (nslots >= 6 * nb ? 3072 : 4096);
    h->ml_agg = (nb <= agg1_max && !h->red.strong_blocks) ? 1 : 4;           // (strong aggregates in blocks of 4 x 8 rows are laid out for AGG = 4)
    int L = 0;
    h->ml_fan.assign(1, 1);
    // composite path: one aggregate per workgroup, at least two coarse levels, 6 n_1 <= 960 (<= 1280 free vertices)
    static const bool comp_off = diag_flag("UZL_ML_NO_COMP");                // A/B switch
    // large graphs (AGG = 4, gather level 2): the same construction one level up - the hierarchy above level 2 as one dense
    // operator that ml_cg_kernel<4> applies instead of its LDS walk (measured 733 -> 332 ms at 20k / 100k, the rebuild's
    // Newton-Schulz GEMMs take 7 ms there).  6 n_2 <= 18432 - the cap was 4096 (21.8k vertices)
    // until round 5, and a 30k / 150k graph took 2.39 s (11.9 k PCG iterations on the walked hierarchy) where it takes 0.32 s with the
    // operator (1.8 k), 40k / 200k 5.13 -> 0.62 s, 50k / 250k 10.9 -> 1.56 s (tests/diag/big_graphs.py; at n = 7500 a GEMM is 10 ms, half of
    // that solve), 64k / 320k ~11 -> 2.3 s, 90k / 450k 29.3 -> 6.3 s.  The path ends where ml_cg's gather-level vector no longer fits the LDS
    // (95k vertices: 6 n_2 = 17.9k, 2.6 GB per matrix, a GEMM 136 ms); kMaxPartials ml_spmv workgroups admit 131k
    static const bool comp4_off = diag_flag("UZL_ML_NO_COMP4");             // A/B switch
    static const int comp4_max = diag_int("UZL_ML_COMP4_MAX", 18432);
    static const int top_wide = diag_int("UZL_ML_TOP_WIDE", kMlTopWide);    // A/B switch (8 = the round-3 hierarchy)
    // A level above the composite one may be the top with up to kMlTopWide aggregates: config 2 (1000 vertices: 125 / 16 / 2) loses its
    // 2-aggregate level and with it ten launches per rebuild (the cycle around it and four Newton-Schulz steps of the 96-row level)
    auto top_max = [&](int lvl) {
        const bool comp_here = h->ml_agg == 1 ? (lvl >= 2 && 6 * h->ml_n[1] <= 3072) : (!comp4_off && lvl >= 3 && 6 * h->ml_n[2] <= comp4_max);
        return (!comp_off && comp_here) ? std::min(std::max(top_wide, kMlTopMax), kMlTopWide) : kMlTopMax;
    };
    while (h->ml_n.back() > top_max(L) && L < kMlMaxLevels) {
        const int fan = (L == 1 && h->ml_agg == 4) ? kMlFanout2 : kMlFanout;     // large graphs: level 2 = 4 level-1 aggregates
        h->ml_fan.push_back(fan);
        h->ml_n.push_back((h->ml_n.back() + fan - 1) / fan);
        L++;
    }
    // the PCG kernels' LDS: with the dense level-2 operator ml_cg stages nothing but the gather-level vector (ml_cg_variant); the walked
    // hierarchy needs every level above the gather level.  Beyond either limit (and beyond kMaxPartials ml_spmv workgroups = 131k
    // vertices): block-Jacobi
    const bool comp4_here = !comp_off && !comp4_off && h->ml_agg == 4 && L >= 3 && 6 * h->ml_n[2] <= comp4_max;
    const bool fits = comp4_here ? ml_comp4_fits(nb, h->ml_n[2]) : ml_fits_lds(h->ml_n.data(), L, h->ml_agg);
    if (!fits) { h->ml_n.assign(1, nb); return; }
    h->ml_levels = L;
    h->ml_lds = comp4_here ? ml_comp4_lds(h->ml_n[2]) : ml_cg_lds_bytes(h->ml_n.data(), L, h->ml_agg);      // what the variant in use asks for: never above kMlLdsLimit
    // per-level host index arrays
    struct Lv { std::vector<int32_t> row_ptr, col, srow, off_ptr, diag_ptr, cslot, chunk; int32_t n_off = 0; };      // cslot / chunk: ml_galerkin_kernel's work list
    std::vector<Lv> lv((size_t)L + 1);
    lv[0].col = col0;
    lv[0].srow.resize(col0.size());
    for (int a = 0; a < nb; a++) for (int s = row_ptr0[a]; s < row_ptr0[a + 1]; s++) lv[0].srow[s] = a;
    for (int f = 0; f < L; f++) {
        Lv& F = lv[f]; Lv& C = lv[f + 1];
        const int nc = h->ml_n[f + 1];
        const int ns = (int)F.col.size();
        const int fan_c = h->ml_fan[f + 1];
        struct Off { int32_t A, C, s; };
        std::vector<Off> off; std::vector<std::pair<int32_t, int32_t>> dg;
        for (int s = 0; s < ns; s++) {
            const int c = F.col[s];
            if (c < 0) continue;
            const int A = F.srow[s] / fan_c, Cc = c / fan_c;
            if (A != Cc) off.push_back({A, Cc, s}); else dg.push_back({A, s});
        }
        // order by (A, C, s) / (A, s).  The slots come in ascending s and a coarse row holds a few dozen of them: a stable counting pass by A,
        // then a small sort inside every row (one std::sort over all of level 0's 100k slots was most of the 4 ms this function took at 10k / 50k)
        {
            std::vector<int32_t> cnt((size_t)nc + 1, 0);
            for (const Off& o : off) cnt[o.A + 1]++;
            for (int a = 0; a < nc; a++) cnt[a + 1] += cnt[a];
            std::vector<Off> tmp(off.size());
            std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
            for (const Off& o : off) tmp[pos[o.A]++] = o;
            off.swap(tmp);
            for (int a = 0; a < nc; a++)
                std::sort(off.begin() + cnt[a], off.begin() + cnt[a + 1], [](const Off& x, const Off& y) { return x.C != y.C ? x.C < y.C : x.s < y.s; });
            std::fill(cnt.begin(), cnt.end(), 0);
            for (const auto& d : dg) cnt[d.first + 1]++;
            for (int a = 0; a < nc; a++) cnt[a + 1] += cnt[a];
            std::vector<std::pair<int32_t, int32_t>> dt(dg.size());
            pos.assign(cnt.begin(), cnt.end() - 1);
            for (const auto& d : dg) dt[pos[d.first]++] = d;                  // (s ascending within A already)
            dg.swap(dt);
        }
        C.row_ptr.assign((size_t)nc + 1, 0);
        for (size_t k = 0; k < off.size(); k++) {
            if (k == 0 || off[k].A != off[k - 1].A || off[k].C != off[k - 1].C) {
                C.srow.push_back(off[k].A); C.col.push_back(off[k].C); C.off_ptr.push_back((int32_t)k);
                C.row_ptr[off[k].A + 1]++;
            }
        }
        C.off_ptr.push_back((int32_t)off.size());
        for (int a = 0; a < nc; a++) C.row_ptr[a + 1] += C.row_ptr[a];
        C.n_off = (int32_t)off.size();
        C.diag_ptr.assign((size_t)nc + 1, 0);
        for (size_t k = 0; k < dg.size(); k++) C.diag_ptr[dg[k].first + 1]++;
        for (int a = 0; a < nc; a++) C.diag_ptr[a + 1] += C.diag_ptr[a];
        // The Galerkin product as a GATHER (ml_galerkin_kernel): contribution q comes from fine slot cslot[q]; a workgroup takes a chunk of
        // consecutive output blocks whose contributions (<= kGalItems) it transforms into LDS and sums in order.  chunk = {kind (0: off-
        // diagonal blocks, 1: diagonal blocks), first output, outputs, first contribution, contributions}.
        C.cslot.resize(off.size() + dg.size());
        for (size_t k = 0; k < off.size(); k++) C.cslot[k] = off[k].s;
        for (size_t k = 0; k < dg.size(); k++) C.cslot[off.size() + k] = dg[k].second;
        {
            const int nso = (int)C.col.size();
            for (int b = 0; b < nso;) {                                      // off-diagonal outputs
                int e = b, items = 0;
                while (e < nso && (e == b || items + (C.off_ptr[e + 1] - C.off_ptr[e]) <= kGalItems) && e - b < kGalOutputs) { items += C.off_ptr[e + 1] - C.off_ptr[e]; e++; }
                const int32_t c5[5] = {0, b, e - b, C.off_ptr[b], items};
                C.chunk.insert(C.chunk.end(), c5, c5 + 5);
                b = e;
            }
            const int nf = h->ml_n[f];
            for (int a = 0; a < nc;) {                                       // diagonal outputs: + two items (G, M) per child
                int e = a, items = 0;
                auto cost = [&](int A) { return (C.diag_ptr[A + 1] - C.diag_ptr[A]) + 2 * (std::min(nf, (A + 1) * fan_c) - A * fan_c); };
                while (e < nc && (e == a || items + cost(e) <= kGalItems) && e - a < kGalOutputs) { items += cost(e); e++; }
                const int32_t c5[5] = {1, a, e - a, C.n_off + C.diag_ptr[a], C.diag_ptr[e] - C.diag_ptr[a]};
                C.chunk.insert(C.chunk.end(), c5, c5 + 5);
                a = e;
            }
        }
        h->ml_nslots.push_back((int32_t)C.col.size());
        h->ml_chunks.push_back((int32_t)(C.chunk.size() / 5));
        h->ml_inner_aggs += nc;                                   // one sibling block per aggregate of every coarse level
    }
    // ---- one arena for everything: first the int arrays (staged on the host), then the doubles
    size_t bytes = 0;
    auto take = [&](size_t b) { size_t o = bytes; bytes = (bytes + b + 255) / 256 * 256; return o; };
    struct IntOff { size_t row_ptr, col, srow, off_ptr, diag_ptr, cslot, chunk; };
    std::vector<IntOff> io((size_t)L + 1);
    for (int l = 0; l <= L; l++) {
        Lv& X = lv[l];
        io[l].row_ptr = take(std::max<size_t>(X.row_ptr.size(), 1) * 4);
        io[l].col = take(std::max<size_t>(X.col.size(), 1) * 4);
        io[l].srow = take(std::max<size_t>(X.srow.size(), 1) * 4);
        io[l].off_ptr = take(std::max<size_t>(X.off_ptr.size(), 1) * 4);
        io[l].diag_ptr = take(std::max<size_t>(X.diag_ptr.size(), 1) * 4);
        io[l].cslot = take(std::max<size_t>(X.cslot.size(), 1) * 4);
        io[l].chunk = take(std::max<size_t>(X.chunk.size(), 1) * 4);
    }
    const size_t int_bytes = bytes;
    struct DblOff { size_t blk, G, M, Winv, geo, cen, r, y; };
    std::vector<DblOff> dof((size_t)L + 1);
    for (int l = 0; l <= L; l++) {
        const size_t n = (size_t)std::max(h->ml_n[l], 1), ns = (size_t)std::max(h->ml_nslots[l], 1);
        dof[l].blk = (l == 0) ? 0 : take(ns * 36 * 8);
        dof[l].G = (l == 0) ? 0 : take(n * 36 * 8);
        dof[l].M = (l == 0) ? 0 : take(n * 36 * 8);
        dof[l].Winv = (l < L) ? take((size_t)std::max(h->ml_n[l + 1], 1) * (size_t)(36 * h->ml_fan[l + 1] * h->ml_fan[l + 1]) * 8) : 0;
        dof[l].geo = (l == 0) ? take(n * 12 * 8) : 0;          // levels >= 1: one contiguous blob (geo_blob below)
        dof[l].cen = take(n * 4 * 8);
        dof[l].r = take(n * 6 * 8);
        dof[l].y = take(n * 6 * 8);
    }
    size_t geo_blob_doubles = 0;
    std::vector<size_t> geo_sub((size_t)L + 1, 0);
    for (int l = 1; l <= L; l++) { geo_sub[l] = geo_blob_doubles; geo_blob_doubles += (size_t)std::max(h->ml_n[l], 1) * 3; }
    const size_t o_geo_blob = take(geo_blob_doubles * 8 + 64);     // ml_cg copies levels g..L-1 with one linear loop
    const bool comp1 = !comp_off && h->ml_agg == 1 && L >= 2 && 6 * h->ml_n[1] <= 3072;     // ml_cg_comp_kernel<5> / <8> / <12> / <16>
    const bool comp4 = comp4_here;                  // (one predicate: the admission test above)
    h->ml_comp = comp1 || comp4;
    h->ml_cl = comp1 ? 1 : (comp4 ? 2 : 0);
    const int cl = h->ml_cl;
    std::vector<size_t> o_dense((size_t)L + 1, 0);
    if (h->ml_comp) for (int l = cl; l < L; l++) o_dense[l] = take((size_t)(6 * h->ml_n[l]) * (size_t)(6 * h->ml_n[l]) * 8);
    static const bool mult_off = diag_flag("UZL_ML_ADDITIVE");                 // A/B switch
    // A handle whose graphs made the multiplicative operator break down (chain-like graphs: few loop closures per vertex, the
    // shape of an online run) keeps the additive operator for its later structures instead of failing once per add_graph.
    h->ml_mult = h->ml_comp && !mult_off && !h->mult_banned;
    const size_t n12 = h->ml_mult ? (size_t)h->ml_n[cl] * h->ml_n[cl + 1] * 36 * 8 : 0;
    const size_t o_mQ = take(n12), o_mQY = take(n12);
    // Newton-Schulz steps of the composite operator per rebuild: 2; 4 on large loopy graphs (AGG = 4, >= 6 slots per row), where two
    // more GEMM pairs per rebuild buy a quarter of the PCG iterations (10k/50k 1882 -> 1455 per solve, 107.7 -> 94.1 ms; 5k/25k 68.6 ->
    // 62.3; 20k/100k 242 -> 224) - on chain-like graphs of that size they cost more than they save (20k/21.7k: 209 -> 261 ms), on
    // small graphs the count barely moves (config 2: 538 -> 511 for +0.3 ms).  tests/diag/ns_sweep.sh
    static const int ns_env = diag_int("UZL_ML_NS_STEPS", -1);
    const int ns_auto = (h->ml_agg == 4 && nslots >= 6 * nb) ? 4 : 2;
    h->ml_ns_steps = h->ml_mult ? std::max(0, std::min(ns_env >= 0 ? ns_env : ns_auto, 4)) : 0;
    const size_t nsq = h->ml_mult ? (size_t)(6 * h->ml_n[cl]) * (size_t)(6 * h->ml_n[cl]) * 8 : 0;     // also the scratch of the levels above cl
    const size_t o_nsT = take(nsq), o_nsX = take(nsq);
    const int c32_stride = h->ml_comp ? ((6 * h->ml_n[cl] + 3) & ~3) : 0;
    const size_t o_c32 = take(h->ml_comp ? (size_t)(6 * h->ml_n[cl]) * c32_stride * 4 + 16384 : 0);      // f32 copy of Y_cl: what the PCG kernels read (+ slack: ml_cg_kernel<4, true, true> prefetches 18 x 512 B per row unconditionally)
    // slot ranges by parent aggregate, for every level the multiplicative cycle is built at (cl .. L-1): [n_l*n_{l+1}] begin | end
    std::vector<std::vector<int32_t>> grp((size_t)L + 1);
    std::vector<size_t> o_grp((size_t)L + 1, 0);
    if (h->ml_mult) {
        for (int l = cl; l < L; l++) {
            const int n1 = h->ml_n[l], n2 = h->ml_n[l + 1], fan2 = h->ml_fan[l + 1];
            grp[l].assign((size_t)2 * n1 * n2, 0);
            for (int i = 0; i < n1; i++) {
                int s = lv[l].row_ptr[i];
                const int send = lv[l].row_ptr[i + 1];
                for (int p = 0; p < n2; p++) {
                    grp[l][(size_t)i * n2 + p] = s;
                    while (s < send && lv[l].col[s] / fan2 == p) s++;
                    grp[l][(size_t)n1 * n2 + (size_t)i * n2 + p] = s;
                }
            }
            o_grp[l] = take(grp[l].size() * 4);
        }
    }
    const size_t o_top = take((size_t)(6 * kMlTopWide) * (6 * kMlTopWide) * 8);
    const int gl = (h->ml_agg == 1 || L < 2) ? 1 : 2;
    const size_t ngz = (size_t)std::max(h->ml_n[gl], 1) * 6 * 8 * 2;          // (x 2: the gather-level-2 Sg holds two parts per entity)
    const size_t o_sg = take(ngz), o_rga = take(ngz), o_rgb = take(ngz), o_vg = take(ngz);
    const size_t buf_bytes = (bytes + 255) / 256 * 256;
    h->ml_arena.reserve(2 * buf_bytes);
    h->ml_copy_stride = buf_bytes;
    std::vector<uint8_t> stage(int_bytes, 0);
    auto put = [&](size_t o, const std::vector<int32_t>& v) { if (!v.empty()) memcpy(stage.data() + o, v.data(), v.size() * 4); };
    for (int l = 0; l <= L; l++) {
        put(io[l].row_ptr, lv[l].row_ptr); put(io[l].col, lv[l].col); put(io[l].srow, lv[l].srow);
        put(io[l].off_ptr, lv[l].off_ptr); put(io[l].diag_ptr, lv[l].diag_ptr);
        put(io[l].cslot, lv[l].cslot); put(io[l].chunk, lv[l].chunk);
    }
    hipStream_t s = h->stream;
    h->d_ml.reserve(2);
    h->d_scal2.reserve(16);
    MlDev Mh[2];
    for (int bi = 0; bi < 2; bi++) {
        uint8_t* base = h->ml_arena.p + (size_t)bi * buf_bytes;
        UZL_HIP(hipMemsetAsync(base, 0, bytes, s));            // padding between arrays takes part in the sharded all-reduce
        UZL_HIP(hipMemcpyAsync(base, stage.data(), int_bytes, hipMemcpyHostToDevice, s));
        MlDev& M = Mh[bi];
        memset(&M, 0, sizeof(M));
        M.levels = L;
        for (int l = 0; l <= L; l++) {
            MlLevel& X = M.lv[l];
            X.n = h->ml_n[l]; X.nslots = h->ml_nslots[l];
            X.fan = h->ml_fan[l];
            X.span = 1; for (int q = 1; q <= l; q++) X.span *= h->ml_fan[q];
            X.row_ptr = (l == 0) ? d_row_ptr : reinterpret_cast<const int32_t*>(base + io[l].row_ptr);
            X.col = (l == 0) ? d_col : reinterpret_cast<const int32_t*>(base + io[l].col);
            X.srow = reinterpret_cast<const int32_t*>(base + io[l].srow);
            X.off_ptr = reinterpret_cast<const int32_t*>(base + io[l].off_ptr);
            X.diag_ptr = reinterpret_cast<const int32_t*>(base + io[l].diag_ptr);
            X.n_off_contrib = lv[l].n_off;
            X.cslot = reinterpret_cast<const int32_t*>(base + io[l].cslot);
            X.chunk = reinterpret_cast<const int32_t*>(base + io[l].chunk);
            X.n_chunks = (int32_t)(lv[l].chunk.size() / 5);
            X.blk = (l == 0) ? d_blk : reinterpret_cast<double*>(base + dof[l].blk);
            X.G = (l == 0) ? d_hdiag : reinterpret_cast<double*>(base + dof[l].G);
            X.M = (l == 0) ? nullptr : reinterpret_cast<double*>(base + dof[l].M);
            X.Winv = (l < L) ? reinterpret_cast<double*>(base + dof[l].Winv) : nullptr;
            X.geo = (l == 0) ? reinterpret_cast<double*>(base + dof[l].geo) : reinterpret_cast<double*>(base + o_geo_blob) + geo_sub[l];
            X.cen = reinterpret_cast<double*>(base + dof[l].cen);
            X.r = reinterpret_cast<double*>(base + dof[l].r);
            X.y = reinterpret_cast<double*>(base + dof[l].y);
        }
        M.top_inv = reinterpret_cast<double*>(base + o_top);
        for (int l = 1; l < L; l++) M.Ydense[l] = (h->ml_comp && l >= cl) ? reinterpret_cast<double*>(base + o_dense[l]) : nullptr;
        for (int l = 0; l <= kMlMaxLevels; l++) h->ml_dense_ptr[bi][l] = (l >= 1 && l < L) ? M.Ydense[l] : nullptr;
        M.comp_level = cl;
        if (h->ml_mult) {
            for (int l = cl; l < L; l++) {
                UZL_HIP(hipMemcpyAsync(base + o_grp[l], grp[l].data(), grp[l].size() * 4, hipMemcpyHostToDevice, s));
                M.grp_beg[l] = reinterpret_cast<const int32_t*>(base + o_grp[l]);
                M.grp_end[l] = M.grp_beg[l] + (size_t)h->ml_n[l] * h->ml_n[l + 1];
            }
        }
        M.nsT = reinterpret_cast<double*>(base + o_nsT); M.nsX = reinterpret_cast<double*>(base + o_nsX);
        M.mQ = reinterpret_cast<double*>(base + o_mQ); M.mQY = reinterpret_cast<double*>(base + o_mQY);
        M.Sg = reinterpret_cast<double*>(base + o_sg);
        uzl_pgo::MlBuf& B = h->mlb[bi];
        B.dml = h->d_ml.p + bi;
        B.nsT = M.nsT; B.nsX = M.nsX; B.y1 = h->ml_comp ? M.Ydense[cl] : nullptr;
        B.rg[0] = reinterpret_cast<double*>(base + o_rga);
        B.rg[1] = reinterpret_cast<double*>(base + o_rgb);
        B.lambda_setup = 0.;
        MlHot& Hh = B.hot;
        memset(&Hh, 0, sizeof(Hh));
        Hh.levels = L;
        for (int l = 0; l <= L; l++) { Hh.n[l] = h->ml_n[l]; Hh.fan[l] = h->ml_fan[l]; Hh.geo[l] = M.lv[l].geo; Hh.Winv[l] = M.lv[l].Winv; }
        Hh.geo0 = M.lv[0].geo; Hh.top_inv = M.top_inv; Hh.Sg = M.Sg; Hh.Vg = reinterpret_cast<double*>(base + o_vg);
        Hh.Cmat = h->ml_comp ? ((h->ml_ns_steps & 1) ? M.nsX : M.Ydense[cl]) : nullptr;   // Newton-Schulz steps ping-pong Y_cl <-> nsX
        Hh.Cmat32 = h->ml_comp ? reinterpret_cast<const float*>(base + o_c32) : nullptr;
        Hh.c32_stride = c32_stride;
        B.l1_span_ptr = M.lv[1].blk;
        h->l1_span = (int64_t)((M.lv[1].M + (size_t)std::max(h->ml_n[1], 1) * 36) - M.lv[1].blk);
    }
    h->ml_ix = 0; h->ml_pending = false;
    UZL_HIP(hipMemcpyAsync(h->d_ml.p, Mh, sizeof(Mh), hipMemcpyHostToDevice, s));
    UZL_HIP(hipStreamSynchronize(s));      // stage / Mh are locals
}

// numeric part, once per linearisation: geometry, then A_{l+1} = P^T A_l P level by level
}  // namespace
namespace uzl {
// Asynchronous rebuild (second stream, second copy of the hierarchy) pays on the composite level-1 path, and for the reduced system of a
// chain-like graph on the level-2 path: there a rebuild (0.8 ms) is as long as the LM iteration it would otherwise hold up.  (Other
// level-2 graphs - config 4 - measured +9 % PCG iterations for no net gain: UZL_ML_ASYNC_LARGE.)
bool ml_async_level(const uzl_pgo* h)
{
    static const bool async_large = diag_flag("UZL_ML_ASYNC_LARGE");          // A/B switches
    static const int async_strong = diag_int("UZL_ML_ASYNC_STRONG", 1);
    return h->ml_cl == 1 || (h->ml_cl == 2 && (async_large || (h->red.on && h->red.strong && async_strong)));
}
void ml_setup_numeric(uzl_pgo* h, int bi, hipStream_t s, const PgoDev& D, bool timed)
{
    if (h->ml_levels == 0) return;
    const int L = h->ml_levels;
    MlDev* dml = h->mlb[bi].dml;
    if (timed) h->timer.begin("ml_geometry", s);
    if (h->ml_n[1] <= kGeoAllMaxHost) k_ml_geometry(D, dml, h->cur, 0, 1, s);            // all levels by one workgroup
    else for (int l = 1; l <= L; l++) k_ml_geometry(D, dml, h->cur, l, h->ml_n[l], s);
    if (timed) h->timer.end(s);
    for (int f = 0; f < L; f++) {
        if (timed) h->timer.begin("ml_galerkin", s);
        k_ml_galerkin(D, dml, f, h->ml_chunks[f + 1], s);
        if (timed) h->timer.end(s);
        if (f == 0) shard_allreduce(h, h->mlb[bi].l1_span_ptr, h->l1_span);     // level 1 complete on every rank: levels >= 2 need no exchange
    }
}
}  // namespace uzl
namespace {

// lambda-dependent part: inverse sibling blocks, top level, dense operator (+ multiplicative cycle, Newton-Schulz refinement)
}  // namespace
namespace uzl {
void ml_setup_trial(uzl_pgo* h, int bi, hipStream_t s, const PgoDev& D, bool timed)
{
    uzl_pgo::MlBuf& B = h->mlb[bi];
    if (timed) h->timer.begin("ml_sibling", s);
    k_ml_sibling(D, B.dml, h->ml_inner_aggs, s);
    if (timed) h->timer.end(s);
    if (h->ml_comp) {
        if (timed) h->timer.begin("ml_dense", s);
        const int cl = h->ml_cl, L = h->ml_levels;
        if (!h->ml_mult) {                                                   // additive operator: Y_l = blockdiag(W_l^-1) + P Y_{l+1} P^T
            for (int l = L - 1; l >= cl; l--) k_ml_dense_level(B.dml, l, h->ml_n[l], s);
            k_ml_cmat32(B.hot, 6 * h->ml_n[cl], s);
            if (timed) h->timer.end(s);
            return;
        }
        // Multiplicative operator.  The cycle X0 = 2S - S A S + Q Y Q^T has eig(X0 A) in (0, 1] - and Newton-Schulz then converges
        // monotonically - only if its coarse operator Y does not OVER-correct (eig(Y A_c) <= 2).  The additive operator of the
        // levels above does (eig up to ~3 on chain-like graphs: tests/diag/cycle_spectrum.py), so those levels are built the same
        // way from the top down: cycle around the (numerically) exact level above, then kUpperNs Newton-Schulz steps.  They are
        // small ((6 n_l)^2 with n_l <= n_cl / 8): a few launches per level.
        for (int l = L - 1; l > cl; l--) {
            k_ml_mult_level(D, B.dml, l, h->ml_n[l], h->ml_n[l + 1], s);
            double* xa = h->ml_dense_ptr[bi][l]; double* xb = B.nsX;
            for (int k = 0; k < kUpperNs; k++) { k_ml_ns_step(D, B.dml, l, h->ml_n[l], xa, B.nsT, xb, s); std::swap(xa, xb); }
        }
        k_ml_mult_level(D, B.dml, cl, h->ml_n[cl], h->ml_n[cl + 1], s);
        if (timed) h->timer.end(s);
        double* xa = B.y1; double* xb = B.nsX;
        const int ns_now = h->ml_ns_now >= 0 ? h->ml_ns_now : h->ml_ns_steps;    // (the same parity as ml_ns_steps: the result lands in the same buffer)
        for (int k = 0; k < ns_now; k++) {
            hipEvent_t ea = nullptr, eb = nullptr;
            if (timed) h->timer.pair("ml_ns_gemm", &ea, &eb);                  // the f64 matrix-core GEMM of the refinement, on its own
            const bool last = k == ns_now - 1;                                 // its epilogue also writes the f32 copy the PCG kernels read
            k_ml_ns_step(D, B.dml, cl, h->ml_n[cl], xa, B.nsT, xb, s, ea, eb, last ? const_cast<float*>(B.hot.Cmat32) : nullptr, B.hot.c32_stride);
            std::swap(xa, xb);
        }
        if (h->ml_ns_steps == 0) k_ml_cmat32(B.hot, 6 * h->ml_n[cl], s);
    }
}
}  // namespace uzl
namespace {

// Order of the free vertices in the block system.  The multilevel preconditioner aggregates 8 CONSECUTIVE blocks, which is only a
// good coarse space when consecutive blocks are strongly coupled.  In a single session the node ids are time-ordered (std::map
// order = trajectory order, graph_slam_node.cpp:294) and the index order is already that; in merged / global-scope graphs
// (graph_slam_node.cpp:401-576, 665-777: ids of several sessions interleave) or graphs without an odometry chain it is not, and
// the PCG iteration count grows 4-7x (tests/diag/vertex_order.py).  So the order is computed from the graph: starting at the
// lowest-index unvisited free vertex, follow the heaviest edge (trace of the information matrix; odometry edges are the stiff
// ones) to an unvisited free vertex until stuck, then extend the same chain backwards from the start; repeat.  Ties go to the lower
// index.  A time-ordered single-session graph comes out in index order (nothing changes for it); O(N + E), deterministic.
std::vector<int32_t> aggregation_order(const uzl_pgo* h)
{
    const int n = h->n, e = h->e;
    std::vector<int32_t> ptr((size_t)n + 1, 0);
    auto is_free = [&](int v) { return !h->fixed_eff[v]; };
    for (int k = 0; k < e; k++) {
        const int a = h->ij[2 * k], b = h->ij[2 * k + 1];
        if (is_free(a) && is_free(b)) { ptr[a + 1]++; ptr[b + 1]++; }
    }
    for (int v = 0; v < n; v++) ptr[v + 1] += ptr[v];
    std::vector<int32_t> nbr((size_t)std::max(ptr[n], 1));
    std::vector<double> wgt((size_t)std::max(ptr[n], 1));
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    const bool have_w = h->edge_w.size() == (size_t)e;
    for (int k = 0; k < e; k++) {
        const int a = h->ij[2 * k], b = h->ij[2 * k + 1];
        if (!(is_free(a) && is_free(b))) continue;
        const double w = have_w ? h->edge_w[k] : 1.;
        nbr[fill[a]] = b; wgt[fill[a]++] = w;
        nbr[fill[b]] = a; wgt[fill[b]++] = w;
    }
    std::vector<uint8_t> seen((size_t)std::max(n, 1), 0);
    std::vector<int32_t> order, back;
    order.reserve((size_t)n);
    auto next_of = [&](int v) {
        int best = -1; double bw = -1.;
        for (int q = ptr[v]; q < ptr[v + 1]; q++) {
            const int u = nbr[q];
            if (seen[u]) continue;
            if (wgt[q] > bw || (wgt[q] == bw && u < best)) { bw = wgt[q]; best = u; }
        }
        return best;
    };
    for (int start = 0; start < n; start++) {
        if (!is_free(start) || seen[start]) continue;
        const size_t first = order.size();
        for (int v = start; v >= 0; v = next_of(v)) { seen[v] = 1; order.push_back(v); }
        back.clear();
        for (int v = next_of(start); v >= 0; v = next_of(v)) { seen[v] = 1; back.push_back(v); }
        if (!back.empty()) {                                     // chain = reverse(back) + forward part
            order.insert(order.begin() + (std::ptrdiff_t)first, back.rbegin(), back.rend());
        }
    }
    return order;
}

}  // namespace
namespace uzl {
void build_structure(uzl_pgo* h)
{
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {                       // UZL_VERBOSE: where the host side of a new structure spends its time
        if (!h->cfg.verbose) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[uzl_pgo] structure: %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    destroy_pcg_graph(h);
    tick("drop the captured graphs");
    const int n = h->n, e = h->e;
    std::vector<int32_t> v2b((size_t)std::max(n, 1), -1), b2v = aggregation_order(h);
    tick("aggregation order");
    const int nb = (int)b2v.size();
    for (int b = 0; b < nb; b++) v2b[b2v[b]] = b;
    h->nb = nb;
    std::vector<int32_t> row_ptr((size_t)nb + 1, 0);
    for (int k = 0; k < e; k++) {
        const int a = v2b[h->ij[2 * k]], b = v2b[h->ij[2 * k + 1]];
        if (a >= 0) row_ptr[a + 1]++;
        if (b >= 0) row_ptr[b + 1]++;
    }
    for (int a = 0; a < nb; a++) row_ptr[a + 1] += row_ptr[a];
    const int nslots = row_ptr[nb];
    h->nslots = nslots;
    std::vector<int32_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    std::vector<int32_t> col((size_t)std::max(nslots, 1)), slot_i((size_t)std::max(e, 1)), slot_j((size_t)std::max(e, 1)), slot_edge((size_t)std::max(nslots, 1));
    for (int k = 0; k < e; k++) {
        const int a = v2b[h->ij[2 * k]], b = v2b[h->ij[2 * k + 1]];
        slot_i[k] = -1; slot_j[k] = -1;
        if (a >= 0) { const int s = fill[a]++; slot_i[k] = s; col[s] = b; slot_edge[s] = 2 * k; }
        if (b >= 0) { const int s = fill[b]++; slot_j[k] = s; col[s] = a; slot_edge[s] = 2 * k + 1; }
    }
    hipStream_t s = h->stream;
    const size_t nbz = std::max(nb, 1), nsz = std::max(nslots, 1);
    h->d_b2v.reserve(nbz); h->d_row_ptr.reserve(nbz + 1); h->d_col.reserve(nsz);
    h->d_blk.reserve(nsz * 36); h->d_slot_edge.reserve(nsz); h->d_srec.reserve(nsz * 44); h->d_smeta.reserve(nsz);
    h->srec_stale = true;
    std::vector<int4> smeta(nsz);
    for (int q = 0; q < nslots; q++) {
        const int k = slot_edge[q] >> 1, other = (slot_edge[q] & 1) ? slot_i[k] : slot_j[k];
        smeta[q] = make_int4(slot_edge[q], h->ij[2 * k], h->ij[2 * k + 1], (other + 1) | (h->robust[k] ? 1 << 30 : 0));
    }
    if (nslots > 0) UZL_HIP(hipMemcpyAsync(h->d_smeta.p, smeta.data(), sizeof(int4) * nslots, hipMemcpyHostToDevice, s));
    h->d_hdiag.reserve(nbz * 42); h->d_minv.reserve(nbz * 36);              // [H_aa | b] contiguous: one all-reduce when sharded
    h->d_x.reserve(nbz * 6); h->d_xs.reserve(nbz * 6); h->d_r.reserve(nbz * 6); h->d_z.reserve(nbz * 6); h->d_p.reserve(nbz * 6); h->d_p2.reserve(nbz * 6); h->d_ap.reserve(nbz * 12 + kMaxPartials);   // [A p | restricted A p | partials]
    if (n > 0) UZL_HIP(hipMemcpyAsync(h->d_v2b.p, v2b.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    if (nb > 0) UZL_HIP(hipMemcpyAsync(h->d_b2v.p, b2v.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, s));
    UZL_HIP(hipMemcpyAsync(h->d_row_ptr.p, row_ptr.data(), sizeof(int32_t) * (nb + 1), hipMemcpyHostToDevice, s));
    if (nslots > 0) UZL_HIP(hipMemcpyAsync(h->d_col.p, col.data(), sizeof(int32_t) * nslots, hipMemcpyHostToDevice, s));
    if (nslots > 0) UZL_HIP(hipMemcpyAsync(h->d_slot_edge.p, slot_edge.data(), sizeof(int32_t) * nslots, hipMemcpyHostToDevice, s));
    std::vector<int32_t> rowhdr((size_t)nbz * kRowHdr, -1);
    for (int a = 0; a < nb; a++) {
        rowhdr[(size_t)a * kRowHdr] = row_ptr[a]; rowhdr[(size_t)a * kRowHdr + 1] = row_ptr[a + 1];
        for (int k = 0; k < 20 && row_ptr[a] + k < row_ptr[a + 1]; k++) rowhdr[(size_t)a * kRowHdr + 2 + k] = col[row_ptr[a] + k];
    }
    // row blocks of the Hessian build: consecutive rows with <= 256 slots and <= 42 rows, so that a workgroup makes one pass (a row with
    // more slots is a block of its own and loops); beyond kMaxPartials blocks (> ~1M slots) blocks hold more and loop as well
    std::vector<int32_t> rb_ptr(1, 0);
    {
        const int cap_slots = std::max(256, (int)(((int64_t)nslots + kMaxPartials / 2 - 1) / (kMaxPartials / 2) + 255) / 256 * 256);
        const int cap_rows = std::max(42, (nb + kMaxPartials / 2 - 1) / (kMaxPartials / 2));
        int rows = 0, slots = 0;
        for (int a = 0; a < nb; a++) {
            const int d = row_ptr[a + 1] - row_ptr[a];
            if (rows > 0 && (rows >= cap_rows || slots + d > cap_slots)) { rb_ptr.push_back(a); rows = 0; slots = 0; }
            rows++; slots += d;
        }
        rb_ptr.push_back(nb);
    }
    h->d_rb_ptr.reserve(rb_ptr.size());
    UZL_HIP(hipMemcpyAsync(h->d_rb_ptr.p, rb_ptr.data(), sizeof(int32_t) * rb_ptr.size(), hipMemcpyHostToDevice, s));
    h->d_rowhdr.reserve(nbz * kRowHdr);
    UZL_HIP(hipMemcpyAsync(h->d_rowhdr.p, rowhdr.data(), sizeof(int32_t) * nbz * kRowHdr, hipMemcpyHostToDevice, s));
    // blocks of slots whose neighbour is fixed are never written: keep them defined
    if (nslots > 0) UZL_HIP(hipMemsetAsync(h->d_blk.p, 0, sizeof(double) * 36 * (size_t)nslots, s));      // (and slots of edges other ranks own stay 0)
    UZL_HIP(hipStreamSynchronize(s));       // host vectors go out of scope
    PgoDev& D = h->D;
    D.n = n; D.nb = nb; D.e = e; D.nslots = nslots;
    D.pose = h->cur; D.pose_trial = h->trial;
    D.v2b = h->d_v2b.p; D.b2v = h->d_b2v.p; D.ei = h->d_ei.p; D.ej = h->d_ej.p;
    D.zinv = h->d_zinv.p; D.info = h->d_info.p; D.robust = h->d_robust.p;
    D.row_ptr = h->d_row_ptr.p; D.col = h->d_col.p; D.rowhdr = h->d_rowhdr.p;
    D.rb_ptr = h->d_rb_ptr.p; D.n_rb = (int32_t)rb_ptr.size() - 1; D.pad_rb = 0; D.srec = h->d_srec.p; D.smeta = h->d_smeta.p; D.blk = h->d_blk.p; D.hdiag = h->d_hdiag.p; D.minv = h->d_minv.p;
    D.b = h->d_hdiag.p + (size_t)nb * 36; D.x = h->d_x.p; D.xs = h->d_xs.p; D.r = h->d_r.p; D.z = h->d_z.p; D.p = h->d_p.p; D.ap = h->d_ap.p;
    D.part_a = h->d_ap.p + (size_t)nbz * 12; D.part_b = h->d_part_b.p; D.part_c = h->d_part_c.p;
    D.scal = h->d_scal.p; D.flags = h->d_flags.p;
    D.e_begin = 0; D.e_end = e; D.diag_owner = 1; D.sibling0 = 1;      // sibling0 finalised after build_ml
    // ---- Schur reduction of the chain interiors (pgo_schur.hpp): when a third or more of the free vertices carry nothing but their two
    //      chain edges, the PCG runs on the Schur complement over the others (sharded solves included: see SchurDev::runblk).
    uzl_pgo::Reduced& Rd = h->red;
    Rd.on = false; Rd.n_int = 0; Rd.n_runs = 0; Rd.longest_run = 0; Rd.strong = false; Rd.strong_blocks = false; Rd.n_sep = 0;
    PgoDev& Dp = h->Dp;
    static const int schur_diag = diag_int("UZL_SCHUR", 1);                  // A/B switches (diagnostic build)
    // longest run: 24.  (Twelve for small graphs paid 3 - 8 % while one wave walked a whole run; with a run eliminated from both ends the
    // chain is twelve steps anyway, and stars of 20-vertex arms keep being eliminated completely: tests/test_schur_gpu.py.)
    static const int schur_cap = diag_int("UZL_SCHUR_CAP", 24);
    static const int schur_min_pct = diag_int("UZL_SCHUR_MIN_PCT", 33);
    const bool may_shard = h->allreduce != nullptr || h->rccl_comm != nullptr;
    std::vector<int32_t> rrow_ptr, rcol;
    if (h->cfg.schur_reduce >= 0 && schur_diag && nb > 0) {
        tick("block-CSR + uploads");
        // The reduced system of a large chain-like graph is numbered by strong aggregates (pgo_schur.hpp) and takes the AGG = 4 hierarchy: the
        // separators an aggregate holds then move (nearly) rigidly together, which is what its six coarse modes can represent.  In row order
        // "8 consecutive separators" put loop-closure partners into different aggregates and the two ends of a long soft run into the same
        // one: config 5's last re-optimisation took 70 - 140 PCG iterations per LM iteration (tests/diag/reduced_proto.py: 70 -> 23).
        static const int strong_env = diag_int("UZL_SCHUR_STRONG_MIN", -1);     // A/B switch: separators from which on (0 = never)
        static const int strong_theta_pct = diag_int("UZL_SCHUR_STRONG_THETA", 25);
        int strong_min = (may_shard || h->cfg.preconditioner == 0) ? 0 : (strong_env >= 0 ? strong_env : kSchurStrongMin);
        // Which numbering (uzl_pgo_cfg::reduced_numbering)?  Strong aggregates pay where loop closures are stiffer than the runs between
        // separators (config 5: 71 -> 31 PCG iterations per LM iteration); where the runs are the stiff part the matching follows the chain,
        // the groups are runs of consecutive separators anyway, and the row order with its level-1 path is better (tests/diag/strong_ab.py).
        // A handle's first structure goes by that shape; afterwards by what its own solves measured - PCG iterations per LM trial of the last
        // solve in either numbering, an iteration on the padded AGG = 4 layout counted as 1.3 (the measured 1.5x per iteration less the rebuilds it saves; up to 256 groups
        // the strong layout runs on the level-1 path, at the row order's cost per iteration) - so an online session that
        // started on the wrong foot corrects itself.  Iteration counts only: deterministic.
        double max_contig = 2.;
        const double cost_strong = 1.;                                       // (the weight of the layout is in the figure: do_optimize)
        if (h->cfg.reduced_numbering == 1) strong_min = 0;
        else if (h->cfg.reduced_numbering != 2 && strong_min > 0) {
            const double ir = h->num_its[0], is = h->num_its[1];
            bool strong;
            if (ir >= 0. && is >= 0.) strong = cost_strong * is < ir;        // both known: the cheaper (the figures are of different solves: a hard
                                                                              // interval can send it the wrong way for one solve - whose own figure sets it right)
            else if (ir >= 0.) strong = ir > 40.;                            // only the row order known: if it is doing badly, try
            else if (is >= 0.) strong = !(is > 40.);                         // only strong aggregates known: likewise
            else { strong = true; max_contig = 0.6; }                        // first structure: by the shape of the groups
            if (!strong) strong_min = 0;
        }
        std::vector<double> slot_w;
        if (strong_min > 0 && h->edge_w.size() == (size_t)e) {
            slot_w.resize((size_t)std::max(nslots, 1));
            for (int q = 0; q < nslots; q++) slot_w[q] = h->edge_w[slot_edge[q] >> 1];
        }
        static const int one_level_max = diag_int("UZL_SCHUR_STRONG_ONE_MAX", kSchurStrongOneMax);
        // (reduced when >= 64 rows and >= schur_min_pct of the rows go: the plan stops early otherwise)
        const int min_int = std::max(64, (int)(((int64_t)schur_min_pct * nb + 99) / 100));
        SchurPlan P = schur_plan(nb, row_ptr, col, schur_cap, slot_w.empty() ? nullptr : slot_w.data(), strong_min, 0.01 * strong_theta_pct, max_contig, one_level_max, min_int);
        tick("Schur plan");
        if (P.n_int >= 64 && (int64_t)100 * P.n_int >= (int64_t)schur_min_pct * nb) {
            Rd.on = true; Rd.n_int = P.n_int; Rd.n_runs = P.n_runs; Rd.longest_run = P.longest_run; Rd.strong = P.strong; Rd.strong_blocks = P.strong && P.n_strong2 > 0; Rd.n_sep = P.n_sep;
            const size_t nr = (size_t)std::max(P.nbr, 1), nsr = (size_t)std::max(P.nslots_r, 1), ni = (size_t)P.n_int, nru = (size_t)P.n_runs;
            auto up = [&](DevBuf<int32_t>& b, const std::vector<int32_t>& v, size_t min_n) {
                b.reserve(std::max(v.size(), min_n));
                if (!v.empty()) UZL_HIP(hipMemcpyAsync(b.p, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice, s));
            };
            up(Rd.run_ptr, P.run_ptr, 1); up(Rd.run_rows, P.run_rows, 1); up(Rd.slotP, P.slotP, 1); up(Rd.slotN, P.slotN, 1);
            up(Rd.endL, P.endL, 1); up(Rd.endR, P.endR, 1); up(Rd.sep_rows, P.sep_rows, 1); up(Rd.rsrc, P.rsrc, 1);
            up(Rd.inc_ptr, P.inc_ptr, 1); up(Rd.inc, P.inc, 1); up(Rd.row_ptr, P.row_ptr, 1); up(Rd.col, P.col, 1);
            std::vector<int32_t> rb2v((size_t)P.nbr), rhdr(nr * kRowHdr, -1);
            for (int i = 0; i < P.nbr; i++) {
                rb2v[i] = P.sep_rows[i] >= 0 ? b2v[P.sep_rows[i]] : -1;
                rhdr[(size_t)i * kRowHdr] = P.row_ptr[i]; rhdr[(size_t)i * kRowHdr + 1] = P.row_ptr[i + 1];
                for (int k = 0; k < 20 && P.row_ptr[i] + k < P.row_ptr[i + 1]; k++) rhdr[(size_t)i * kRowHdr + 2 + k] = P.col[P.row_ptr[i] + k];
            }
            up(Rd.b2v, rb2v, 1); up(Rd.rowhdr, rhdr, 1);
            Rd.elim.reserve(std::max<size_t>(ni, 1) * kSchurElim); Rd.runout.reserve(std::max<size_t>(nru, 1) * kSchurRunOut);
            Rd.blk.reserve(nsr * 36); Rd.hdiag.reserve(nr * 42); Rd.minv.reserve(nr * 36);
            Rd.x.reserve(nr * 6); Rd.xs.reserve(nr * 6); Rd.r.reserve(nr * 6); Rd.z.reserve(nr * 6); Rd.p.reserve(nr * 6); Rd.p2.reserve(nr * 6); Rd.ap.reserve(nr * 12 + kMaxPartials);
            UZL_HIP(hipStreamSynchronize(s));                                  // P's vectors and the two locals go out of scope
            SchurDev& S = Rd.S;
            S.n_runs = P.n_runs; S.n_int = P.n_int; S.nbr = P.nbr; S.nslots_r = P.nslots_r;
            S.run_ptr = Rd.run_ptr.p; S.run_rows = Rd.run_rows.p; S.slotP = Rd.slotP.p; S.slotN = Rd.slotN.p; S.endL = Rd.endL.p; S.endR = Rd.endR.p;
            S.sep_rows = Rd.sep_rows.p; S.rsrc = Rd.rsrc.p; S.inc_ptr = Rd.inc_ptr.p; S.inc = Rd.inc.p; S.elim = Rd.elim.p; S.runout = Rd.runout.p;
            // Sharded solve: a run is eliminated by EVERY rank (the reduced system's diagonal blocks and right-hand side are then complete
            // everywhere, like H_aa | b after its all-reduce), so each needs the run's chain blocks whole - one more all-reduce per
            // linearisation, [E_m per eliminated vertex | C_1 per run] gathered into a contiguous buffer.  Fill blocks enter A p on rank 0 only.
            S.runblk = nullptr;
            if (may_shard) { Rd.runblk.reserve((ni + nru) * 36); S.runblk = Rd.runblk.p; }
            Dp = D;
            Dp.nb = P.nbr; Dp.nslots = P.nslots_r; Dp.b2v = Rd.b2v.p; Dp.row_ptr = Rd.row_ptr.p; Dp.col = Rd.col.p; Dp.rowhdr = Rd.rowhdr.p;
            Dp.blk = Rd.blk.p; Dp.hdiag = Rd.hdiag.p; Dp.minv = Rd.minv.p; Dp.b = Rd.hdiag.p + (size_t)P.nbr * 36;
            Dp.x = Rd.x.p; Dp.xs = Rd.xs.p; Dp.r = Rd.r.p; Dp.z = Rd.z.p; Dp.p = Rd.p.p; Dp.ap = Rd.ap.p; Dp.part_a = Rd.ap.p + nr * 12;
            Dp.smeta = nullptr; Dp.srec = nullptr;                            // the reduced system is assembled by schur_assemble_kernel
            rrow_ptr.swap(P.row_ptr); rcol.swap(P.col);
        }
    }
    PgoDev& Dsys = Rd.on ? Dp : D;                                             // what the PCG kernels get
    const int nbp = Dsys.nb;
    double* apbuf = Rd.on ? Rd.ap.p : h->d_ap.p;
    tick("reduced system uploads");
    if (Rd.on) build_ml(h, rrow_ptr, rcol, nbp, Dsys.nslots, Rd.row_ptr.p, Rd.col.p, Rd.blk.p, Rd.hdiag.p);
    else build_ml(h, row_ptr, col, nb, nslots, h->d_row_ptr.p, h->d_col.p, h->d_blk.p, h->d_hdiag.p);
    tick("hierarchy (host index arrays + uploads)");
    {   // per-iteration exchange buffer: [A p (6 nb) | restricted A p (6 n_g) | p.Ap partials]
        const int gl = (h->ml_levels == 0) ? 0 : ((h->ml_agg == 1 || h->ml_levels < 2) ? 1 : 2);
        const size_t ng6 = gl ? (size_t)h->ml_n[gl] * 6 * (gl == 2 ? 2 : 1) : 0;      // gather level 2: two half-aggregate parts per entity (sg_at)
        if (gl) { h->mlb[0].hot.Sg = apbuf + (size_t)nbp * 6; h->mlb[1].hot.Sg = h->mlb[0].hot.Sg; }
        Dsys.part_a = apbuf + (size_t)nbp * 6 + ng6;
        h->iter_span = (int64_t)((size_t)nbp * 6 + ng6 + (gl ? (size_t)g_ml_spmv(nbp, h->ml_agg) : 0));
    }
    Dsys.sibling0 = (h->ml_levels > 0 && h->ml_agg == 1) ? 1 : 0;     // large graphs keep the level-0 smoother block-diagonal
    D.sibling0 = Dsys.sibling0;
    // ---- sharded solve (BASELINE config 4): this rank linearises a contiguous range of the system edges
    // (a callback with world_size 1 still runs every exchange step: that is how the RCCL callback is tested on one GPU;
    //  graphs too small for the multilevel path are simply solved redundantly by every rank)
    h->sharded = h->ml_levels > 0 && may_shard;
    if (h->sharded) {
        const int base = e / h->world, rem = e % h->world;
        D.e_begin = h->rank * base + std::min(h->rank, rem);
        D.e_end = D.e_begin + base + (h->rank < rem ? 1 : 0);
        D.diag_owner = h->rank == 0 ? 1 : 0;
        D.sibling0 = 0;
    }
    if (Rd.on) {                                                              // (Dp was copied from D before the rank's share was known)
        Dp.e_begin = D.e_begin; Dp.e_end = D.e_end; Dp.diag_owner = D.diag_owner; Dp.sibling0 = D.sibling0;
        Rd.S.runblk = h->sharded ? Rd.runblk.p : nullptr;
    }
    if (!Rd.on) Dp = D;
    h->pbuf[0] = Rd.on ? Rd.p.p : h->d_p.p; h->pbuf[1] = Rd.on ? Rd.p2.p : h->d_p2.p;
    h->structure_ready = true;
    h->structure_gen++;
}
}  // namespace uzl
namespace {

// enqueue `pairs` x 2 PCG iterations (p0 -> p1 -> p0); kernels no-op once the device `done` flag is set
void enqueue_pcg_pairs(uzl_pgo* h, int pairs, bool timed)
{
    hipStream_t s = h->stream;
    const PgoDev& D = h->Dp;
    const double tol2 = h->cfg.pcg_tol * h->cfg.pcg_tol;
    const bool ml = h->ml_levels > 0;
    const int ga = ml ? g_ml_spmv(D.nb, h->ml_agg) : g_pcg_spmv(D.nb), gu = ml ? g_ml_rows(D.nb, h->ml_agg) : g_pcg_update(D.nb);   // partials written by spmv / by cg
    double* pb[2] = {h->pbuf[0], h->pbuf[1]};
    auto progress = [&]() {                                                         // how far is x from settled: the stop test (pgo_kernels.hip)
        if (timed) h->timer.begin("pcg_progress", s);
        k_pcg_progress(D, s);
        if (timed) h->timer.end(s);
    };
    for (int i = 0; i < 2 * pairs; i++) {
        if (!ml && i > 0 && i % kProgressEveryBJ == 0) progress();
        double* po = pb[i & 1];
        double* pn = pb[(i & 1) ^ 1];
        hipEvent_t ea = nullptr, eb = nullptr;
        if (ml) {
            if (timed) h->timer.pair("pcg_spmv", &ea, &eb);                      // dispatch timestamps: agree with rocprofv3
            k_ml_spmv(D, h->mlb[h->ml_ix].hot, h->ml_agg, po, pn, gu, tol2, s, ea, eb);
            shard_allreduce(h, D.ap, h->iter_span);                              // the one exchange per PCG iteration
            ea = eb = nullptr;
            if (timed) h->timer.pair("ml_cg", &ea, &eb);
            UZL_HIP(k_ml_cg(D, h->mlb[h->ml_ix].hot, h->ml_agg, pn, h->mlb[h->ml_ix].rg[(i & 1) ^ 1], h->mlb[h->ml_ix].rg[i & 1], ga, 0, h->ml_lds, s, ea, eb));
        } else {
            if (timed) h->timer.begin("pcg_spmv", s);
            k_pcg_spmv(D, po, pn, gu, tol2, s);
            if (timed) h->timer.end(s);
            if (timed) h->timer.begin("pcg_update", s);
            k_pcg_update(D, pn, ga, s);
            if (timed) h->timer.end(s);
        }
    }
    if (!ml) progress();
}

// |r|^2 / |b|^2 a solve under the multiplicative operator must reach.  Deliberately loose: legitimate solves end at 1e-10 .. 1e-6 while
// the linearisation moves and at ~1e-3 once LM has converged and b itself is rounding noise (a 1e-4 guard tripped there and threw a
// healthy operator away); an operator that is not SPD leaves |r| of the order of |b| or above.
// (kResidualGuard = 0.25: pgo_lm.hpp)

}  // namespace
namespace uzl {
void destroy_pcg_graph(uzl_pgo* h)
{
    for (auto& B : h->mlb) {
        if (B.graph_exec) { (void)hipGraphExecDestroy(B.graph_exec); B.graph_exec = nullptr; }
        if (B.graph) { (void)hipGraphDestroy(B.graph); B.graph = nullptr; }
        if (B.graph_exec_s) { (void)hipGraphExecDestroy(B.graph_exec_s); B.graph_exec_s = nullptr; }
        if (B.graph_s) { (void)hipGraphDestroy(B.graph_s); B.graph_s = nullptr; }
    }
}
}  // namespace uzl
namespace {

// The launch-bound inner loop is captured once per problem structure and preconditioner copy: every kernel argument (pointers,
// partial counts, tolerance) is fixed, lambda and the CG scalars live in device memory.  Two lengths: 2 x kGraphPairs iterations,
// and 2 x kShortPairs for solves expected to end at once (a launch behind convergence is a no-op, but still ~1.2 us of stream time:
// a converged LM iteration's solve of 2 - 4 iterations used to pay for 28 of them).
void ensure_pcg_graph(uzl_pgo* h)
{
    uzl_pgo::MlBuf& B = h->mlb[h->ml_ix];
    if (B.graph_exec) return;
    const auto t0 = std::chrono::steady_clock::now();
    UZL_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_pcg_pairs(h, kGraphPairs, false);
    UZL_HIP(hipStreamEndCapture(h->stream, &B.graph));
    UZL_HIP(hipGraphInstantiate(&B.graph_exec, B.graph, nullptr, nullptr, 0));
    UZL_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_pcg_pairs(h, kShortPairs, false);
    UZL_HIP(hipStreamEndCapture(h->stream, &B.graph_s));
    UZL_HIP(hipGraphInstantiate(&B.graph_exec_s, B.graph_s, nullptr, nullptr, 0));
    if (h->cfg.verbose) fprintf(stderr, "[uzl_pgo] structure: %-28s %.3f ms\n", "PCG graphs of one copy", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    h->structure_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// one (H + lambda I) dx = b solve; returns PCG iterations used, sets *converged
int pcg_solve(uzl_pgo* h, bool* converged)
{
    hipStream_t s = h->stream;
    const PgoDev& D = h->Dp;
    if (D.nb == 0) { *converged = true; h->prev_pcg_iters = 0; h->last_residual_ratio = 0.; return 0; }     // every free vertex was Schur-eliminated: nothing left to iterate on
    const int max_it = h->cfg.pcg_max_iter > 0 ? h->cfg.pcg_max_iter : 6 * std::max(D.nb, 1);
    const bool timed = h->timer.on || h->no_graph || h->sharded;   // per-kernel events, rocprofv3 and the exchange callback need eager launches
    if (h->ml_levels > 0) {
        if (h->ml_trial_setup) {
            ml_setup_trial(h, h->ml_ix, s, D, true);
            h->ml_trial_setup = false;
            h->mlb[h->ml_ix].lambda_setup = h->lambda_now;
        }
        uzl_pgo::MlBuf& B = h->mlb[h->ml_ix];
        { Timed t(h, "pcg_init"); k_ml_init(D, B.hot, h->ml_agg, h->pbuf[0], h->pbuf[1], B.rg[0], s); }
        { Timed t(h, "ml_cg"); UZL_HIP(k_ml_cg(D, B.hot, h->ml_agg, h->pbuf[0], B.rg[0], B.rg[1], 0, 1, h->ml_lds, s)); }
    } else {
        { Timed t(h, "precond"); k_precond(D, s); }
        Timed t(h, "pcg_init"); k_pcg_init(D, h->pbuf[0], h->pbuf[1], s);
    }
    if (!timed) ensure_pcg_graph(h);
    int launched = 0;
    // first batch sized from the previous solve (in steps of 2 x kShortPairs iterations), then two short ones, then long ones; the
    // kernels no-op once `done` is set
    static const int first_pct = diag_int("UZL_FIRST_PCT", 95);
    constexpr int kStep = 2 * kShortPairs;
    const int kLong = 2 * kGraphPairs;
    auto round_up = [](int v) { return ((v + kStep - 1) / kStep) * kStep; };
    // (+ 1: a solve that ends by the stop test after k iterations is declared done by the ml_spmv of iteration k + 1)
    int want = h->prev_pcg_iters > 0 ? std::max(kStep, round_up((h->prev_pcg_iters * first_pct) / 100 + 1)) : kLong;
    for (int round = 0;; round++) {
        want = round_up(std::max(1, std::min(want, max_it - launched)));
        if (timed) enqueue_pcg_pairs(h, want / 2, h->timer.on);
        else {
            for (int i = 0; i < want / kLong; i++) UZL_HIP(hipGraphLaunch(h->mlb[h->ml_ix].graph_exec, s));
            for (int i = 0; i < (want % kLong + kStep - 1) / kStep; i++) UZL_HIP(hipGraphLaunch(h->mlb[h->ml_ix].graph_exec_s, s));
        }
        launched += want;
        k_residual_guard(D, s);                                   // a no-op until `done` is set
        fetch_scal(h);
        if (h->h_scal.p->flags[0] || launched >= max_it) break;
        want = round < 2 ? kStep : kLong;
    }
    UZL_HIP(hipGetLastError());
    *converged = h->h_scal.p->flags[0] != 0 && h->h_scal.p->flags[2] == 0;
    // The multiplicative cycle / Newton-Schulz operator is not SPD by construction (see the fallback in do_optimize).  The
    // recurrence residual r is the true residual of x whatever the preconditioner did, so a solve under that operator which
    // claims convergence in the M^-1 norm while |r| has not come down by kResidualGuard relative to |b| is refused and the
    // caller falls back to the additive operator (a sum of SPD terms, whose M^-1 norm is a norm).
    h->last_residual_ratio = h->h_scal.p->scal[7];
    if (*converged && (h->ml_mult || h->ml_ns_steps > 0) && !(h->last_residual_ratio <= kResidualGuard)) { *converged = false; h->guard_trips++; }
    const int iters = h->h_scal.p->flags[1];
    h->prev_pcg_iters = iters;
    return iters;
}

}  // namespace
namespace uzl {
// optimizeImpl's front part (initializeOptimization :139, setFixedNodes :144-146): the structure, cached until the next add_graph / set_graph
void prepare_optimize(uzl_pgo* h)
{
    h->t_start = std::chrono::steady_clock::now();
    h->structure_ms = 0.; h->exchange_ms = 0.; h->exchange_calls = 0;
    h->last_structure_reused = h->structure_ready;
    if (!h->structure_ready) {
        const auto ts = std::chrono::steady_clock::now();
        h->fixed_eff = h->fixed_in;
        h->n_gauge = gauge_fix(h);
        build_structure(h);
        h->structure_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
    }
    if (h->srec_stale) {                // the edges' values in slot order (new values on a kept structure, or a new structure)
        k_slot_records(h->d_zinv.p, h->d_info.p, h->e, h->d_slot_edge.p, h->nslots, h->d_srec.p, h->stream);
        h->srec_stale = false;
    }
}
int do_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st)
{
    if (!h->have_graph) return pgo_fail(h, UZL_ERR_STATE, "optimize before add_graph/set_graph");
    UZL_HIP(hipSetDevice(h->cfg.device));
    if (iterations <= 0) iterations = h->cfg.iterations;
    prepare_optimize(h);
    uzl_pgo_stats local;
    uzl_pgo_stats* sp = st ? st : &local;
    const int rc = lm_eligible(h) ? do_optimize_lm(h, iterations, sp) : do_optimize_host(h, iterations, sp);
    if (h->red.on && h->red.n_sep >= kSchurStrongMin && sp->lm_trials > 0 && (rc == UZL_OK || rc == UZL_ERR_NOT_CONVERGED)) {      // what the next structure's numbering goes by
        h->num_last = h->red.strong ? 1 : 0;
        h->num_its[h->num_last] = (h->red.strong_blocks ? 1.3 : 1.) * (double)sp->pcg_iterations / sp->lm_trials;      // an iteration on the padded AGG = 4 layout: 20 against 14 us, less the rebuilds it saves
    }
    return rc;
}
// The host-driven loop.  Called through do_optimize (structure prepared, device set), or by do_optimize_lm for a solve that met an anomaly.
int do_optimize_host(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st)
{
    const auto t0 = h->t_start;
    uzl_pgo_stats S;
    memset(&S, 0, sizeof(S));
    S.structure_reused = h->last_structure_reused ? 1 : 0;
    S.n_vertices = h->n; S.n_edges = h->e; S.n_gauge_fixed = h->n_gauge;
    hipStream_t s = h->stream;
    PgoDev& D = h->D;
    PgoDev& Dp = h->Dp;                       // the system the PCG solves: D, or the Schur complement over the separator vertices
    const bool red = h->red.on;
    const SchurDev& SD = h->red.S;
    S.n_eliminated = red ? h->red.n_int : 0; S.reduced_strong = (red && h->red.strong) ? 1 : 0;
    // (H + lambda I) with the chain interiors eliminated: once per lambda, i.e. per LM trial (pgo_schur.hpp)
    auto schur_reduce = [&]() {
        { Timed t(h, "schur_eliminate"); k_schur_eliminate(D, SD, s); }
        { Timed t(h, "schur_assemble"); k_schur_assemble(D, Dp, SD, s); }
    };
    const double delta = h->cfg.huber_delta;
    h->timer.reset();
    h->prev_pcg_iters = 0;
    int rc = UZL_OK;
    if (h->nb == 0 || h->e == 0) {
        // nothing to optimise: chi2 only
        if (h->e > 0) {
            int g;
            { Timed t(h, "chi2"); g = k_chi2(D, h->cur, delta, s); }
            k_finalize(D, g, 0, 0, 0, s);
            fetch_scal(h);
            S.chi2_initial = S.chi2_final = h->h_scal.p->scal[4];
        }
        S.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (st) *st = S;
        return UZL_OK;
    }
    // optimizer_.optimize(iterations) (:148) -> OptimizationAlgorithmLevenberg::solve [EXT]
    double lambda = 0., ni = 2., current_chi = 0.;
    double last_rel = 1e300;
    double rate_ref = -1., rate_last = -1.;
    // Asynchronous rebuild: from the second LM iteration on a wanted rebuild runs on stream2 into the OTHER copy of the
    // hierarchy while this iteration's PCG still uses the current one (any SPD preconditioner gives the same solution; one
    // that is one linearisation old costs a few iterations, a rebuild on the critical path costs ~0.4 ms).  The copy is
    // adopted at the start of the next iteration, which has to wait for it anyway before it overwrites H and the poses.
    static const bool async_off = diag_flag("UZL_ML_SYNC_REBUILD");             // A/B switch
    // (small graphs only: at 10k vertices the rebuild's Newton-Schulz GEMMs take more from the overlapped PCG than they give back:
    // 113.2 -> 115.1 ms; config 2: 11.09 -> 10.67 ms with 540 instead of 517 PCG iterations)
    const bool async_ok = !async_off && h->ml_levels > 0 && ml_async_level(h) && !h->sharded && !h->timer.on && h->stream2 != nullptr;
    bool adopted = false;
    h->ml_ix = 0; h->ml_pending = false;
    struct DrainRebuild {                      // an exception must not leave a rebuild running on stream2 behind the handle's back
        uzl_pgo* h;
        ~DrainRebuild() { if (h->ml_pending) { (void)hipStreamSynchronize(h->stream2); h->ml_pending = false; } }
    } drain{h};
    // diagnostic build, UZL_PHASES=1: GPU time between phase marks of every LM iteration (events on the solver's stream, graph replays as
    // in production), printed after the solve
    static const bool phases_on = diag_flag("UZL_PHASES");
    std::vector<hipEvent_t> ph_ev;
    std::vector<int> ph_tag;
    auto mark = [&](int tag) {
        if (!phases_on) return;
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s); ph_ev.push_back(e); ph_tag.push_back(tag);
    };
    for (int it = 0; it < iterations; it++) {
        int gl, ga;
        adopted = false;
        mark(0);
        if (h->ml_pending) {                                                      // the copy built during the last iteration
            UZL_HIP(hipStreamWaitEvent(s, h->ev_setup, 0));
            h->ml_ix ^= 1; h->ml_pending = false; adopted = true;
        }
        D.pose = h->cur; D.pose_trial = h->trial;
        Dp.pose = h->cur; Dp.pose_trial = h->trial;
        // (computeActiveErrors: chi2 of the linearisation point is needed in the first iteration, and by every rank of a sharded solve,
        //  which exchanges it; later iterations carry the accepted trial's over - as lm_head_kernel does)
        {                                                                       // computeActiveErrors + buildSystem
            hipEvent_t ea = nullptr, eb = nullptr;
            h->timer.pair("linearize", &ea, &eb);                               // (profiling: dispatch timestamps, as rocprofv3 reports the kernel)
            UZL_HIP(k_hessian(D, h->cur, delta, &gl, &ga, s, it == 0 || h->sharded, ea, eb));
        }
        if (h->sharded) {                                                         // H_aa, b: sums over all ranks' edges
            shard_allreduce(h, h->d_hdiag.p, (int64_t)h->nb * 42);
            ga = k_diagmax(D, s);
        }
        { Timed t(h, "finalize"); k_finalize(D, gl, 0, ga, 2, s); }
        shard_allreduce_chi2(h);
        // The multilevel preconditioner is rebuilt only while the linearisation still moves: any SPD preconditioner
        // gives the same PCG solution, and once chi2 changes by less than refresh_rel per step the hierarchy of the
        // previous iteration is as good as a fresh one (geometry + Galerkin + inverses are ~170 us per rebuild).
        // A rebuild is also forced when the iteration count has grown by a third since the last one.
        // (an asynchronous rebuild is for the NEXT iteration: none in the last one)
        const bool refresh = lm_refresh(it, iterations, always_refresh, !async_ok, last_rel, ml_async_level(h) ? refresh_rel : kRefreshRelSync, rate_ref, rate_last, ml_rate_drop(h));
        bool launch_async = false;
        bool fetched = false;
        if (red) {                                      // the hierarchy is built on the reduced system, which needs lambda: lambda_0 first
            if (h->sharded && SD.runblk) {              // the runs' chain blocks, summed over the ranks (once per linearisation)
                k_schur_gather(D, SD, s);
                shard_allreduce(h, SD.runblk, (int64_t)(SD.n_int + SD.n_runs) * 36);
            }
            if (it == 0) {
                fetch_scal(h); fetched = true;
                current_chi = h->h_scal.p->scal[4];
                S.chi2_initial = current_chi;
                lambda = 1e-5 * h->h_scal.p->scal[6];                             // computeLambdaInit: tau * max diag
                ni = 2.;
            }
            k_set_scalar(D.scal + 3, lambda, s);
            schur_reduce();
        }
        h->ml_ns_now = ml_ns_steps_at(h->ml_ns_steps, it);       // synchronous set-ups of this iteration (the device-resident loop: lm_drive)
        if (refresh) {
            S.precond_builds++;
            if (it == 0 || !async_ok) { ml_setup_numeric(h, h->ml_ix, s, Dp, true); h->ml_trial_setup = true; }
            else launch_async = true;                                             // needs this iteration's lambda: below
        }
        if ((it == 0 || h->sharded) && !fetched) {     // later iterations carry chi2 over from the accepted trial: no round trip
            fetch_scal(h);
            current_chi = h->h_scal.p->scal[4];
        }
        if (it == 0 && !fetched) {
            S.chi2_initial = current_chi;
            lambda = 1e-5 * h->h_scal.p->scal[6];                                 // computeLambdaInit: tau * max diag
            ni = 2.;
        }
        if (launch_async) {
            const int nb_ix = h->ml_ix ^ 1;
            UZL_HIP(hipEventRecord(h->ev_lin, s));                                // H, b and the poses of this linearisation are final
            UZL_HIP(hipStreamWaitEvent(h->stream2, h->ev_lin, 0));
            PgoDev D2 = Dp;
            D2.scal = h->d_scal2.p;                                               // the trial loop below moves scal[3] on the main stream
            k_set_scalar(D2.scal + 3, lambda, h->stream2);
            ml_setup_numeric(h, nb_ix, h->stream2, D2, false);
            { const int keep = h->ml_ns_now; h->ml_ns_now = -1; ml_setup_trial(h, nb_ix, h->stream2, D2, false); h->ml_ns_now = keep; }      // (a rebuild that runs ahead: the structure's steps)
            h->mlb[nb_ix].lambda_setup = lambda;
            UZL_HIP(hipEventRecord(h->ev_setup, h->stream2));
            h->ml_pending = true;
        }
        double rho = 0.;
        int qmax = 0;
        const double tol_f2 = tol_factor2(h->cfg);
        mark(1);
        do {
            set_lambda(h, lambda, tol_f2);                                        // setLambda (+ this iteration's PCG tolerance)
            if (red && qmax > 0) {                                                // a rejected step moved lambda: the Schur complement with it
                if (h->ml_pending) UZL_HIP(hipStreamWaitEvent(s, h->ev_setup, 0));   // (a rebuild running ahead on stream2 still reads the old one)
                schur_reduce();
            }
            bool conv = false;
            // the lambda-dependent inverses of the hierarchy are kept across trials; after rejected steps lambda grows
            // geometrically and inverses taken at a much smaller lambda stop being a preconditioner at all
            if (h->ml_levels > 0 && lambda > kLambdaRetake * h->mlb[h->ml_ix].lambda_setup) h->ml_trial_setup = true;
            const bool fresh = h->ml_trial_setup || (adopted && qmax == 0);
            int pcg_its = pcg_solve(h, &conv);                                    // _solver->solve()
            S.pcg_iterations += pcg_its;
            if (!conv && !fresh && h->ml_levels > 0) {                            // stale hierarchy: retake the inverses once
                h->ml_trial_setup = true;
                pcg_its = pcg_solve(h, &conv);
                S.pcg_iterations += pcg_its;
            }
            if (!conv && (h->ml_mult || h->ml_ns_steps > 0)) {
                if (h->cfg.verbose)
                    fprintf(stderr, "[uzl_pgo] it %d trial %d lambda %.3e: multiplicative operator refused (pcg %d, rz %.3e, breakdown %d, |r|2/|b|2 %.3e) -> additive\n",
                            it, qmax, lambda, pcg_its, h->h_scal.p->scal[0], (int)h->h_scal.p->flags[2], h->last_residual_ratio);
                // The multiplicative cycle / its Newton-Schulz refinement is SPD only while the level-1 smoother contracts
                // (lambda_max(Y_1 A_1) < 2), which block-Jacobi does not guarantee on every graph: fall back, for the rest
                // of this handle's structure, to the additive operator (a sum of SPD terms) and solve again.
                if (h->ml_pending) {                                              // the copy being built on stream2 still holds the
                    UZL_HIP(hipStreamSynchronize(h->stream2));                    // multiplicative operator: drop it and rebuild
                    h->ml_pending = false; last_rel = 1e300;                      // synchronously at the next linearisation
                }
                h->ml_mult = false; h->ml_ns_steps = 0; h->ml_ns_now = -1; h->mult_banned = true;      // (the additive operator takes no refinement steps)
                for (auto& B : h->mlb) B.hot.Cmat = B.y1;
                destroy_pcg_graph(h);                                             // MlHot is a by-value kernel argument
                h->structure_gen++;                                               // (slot tables that carry it are rebuilt: uzl_pgo_lm.hip, batches)
                h->ml_trial_setup = true;
                pcg_its = pcg_solve(h, &conv);
                S.pcg_iterations += pcg_its;
            }
            {
                const double rate = pcg_rate(h->h_scal.p->scal[1], h->h_scal.p->scal[0], pcg_its, h->cfg.pcg_tol * h->cfg.pcg_tol, tol_f2);
                if (rate > 0.) { rate_last = rate; if (fresh || rate_ref < 0.) rate_ref = rate; }
            }
            if (h->cfg.verbose)
                fprintf(stderr, "[uzl_pgo] it %d trial %d lambda %.3e pcg %d  rz_end %.3e  rz_stop %.3e  |r|2/|b|2 %.3e  conv %d  chi2 %.9g\n", it, qmax, lambda, pcg_its,
                        h->h_scal.p->scal[0], h->h_scal.p->scal[1], h->last_residual_ratio, (int)conv, current_chi);
            if (!conv) { S.pcg_not_converged++; rc = UZL_ERR_NOT_CONVERGED; }
            S.lm_trials++;
            mark(2);
            if (red) { Timed t(h, "schur_backsub"); k_schur_backsub(D, Dp, SD, s); }   // dx of the eliminated vertices from the separators'
            int go, gc;
            { Timed t(h, "oplus"); go = k_oplus(D, h->cur, h->trial, s); }        // push + update
            { Timed t(h, "chi2"); gc = k_chi2_trial(D, h->cur, delta, s); }       // computeActiveErrors (at trial poses made on the fly: the arithmetic of the device-resident loop's eval_lm_kernel)
            { Timed t(h, "finalize"); k_finalize(D, gc, go, 0, 1, s); }
            shard_allreduce_chi2(h);
            fetch_scal(h);
            mark(3);
            const double temp_chi = h->h_scal.p->scal[4];
            const LmStep step = lm_step(current_chi, temp_chi, h->h_scal.p->scal[5], lambda, ni);     // rho, lambda, ni (pgo_lm.hpp)
            rho = step.rho;
            if (step.accepted) {                                                  // good step
                last_rel = step.last_rel;
                current_chi = temp_chi;
                std::swap(h->cur, h->trial);                                      // discardTop
            }                                                                     // else pop: h->cur untouched
            qmax++;
        } while (rho < 0 && qmax < 10);
        S.iterations_done = it + 1;
        if (qmax == 10 || rho == 0) { S.terminated_early = 1; break; }           // Terminate
    }
    h->ml_ns_now = -1;
    S.chi2_final = current_chi;
    S.lambda_final = lambda;
    if (h->ml_pending) { UZL_HIP(hipStreamSynchronize(h->stream2)); h->ml_pending = false; }   // a rebuild nobody will use: let it drain
    UZL_HIP(hipStreamSynchronize(s));
    if (phases_on && ph_ev.size() > 1) {
        double acc[4] = {0., 0., 0., 0.};
        static const char* nm[4] = {"linearise+set-up (0->1)", "solve incl. init (1->2)", "evaluate+round trip (2->3)", "host decision / next (3->0)"};
        static const bool phases_each = diag_flag("UZL_PHASES_EACH");
        for (size_t i = 0; i + 1 < ph_ev.size(); i++) {
            float ms = 0.f; (void)hipEventElapsedTime(&ms, ph_ev[i], ph_ev[i + 1]); acc[ph_tag[i] & 3] += ms;
            if (phases_each) fprintf(stderr, "%s%d:%.0f", ph_tag[i] == 0 ? "\n[uzl_pgo]   " : " ", ph_tag[i], 1e3 * ms);
        }
        if (phases_each) fprintf(stderr, "\n");
        fprintf(stderr, "[uzl_pgo] phases over %d LM iterations (GPU event time, ms):", S.iterations_done);
        for (int k = 0; k < 4; k++) fprintf(stderr, "  %s %.3f", nm[k], acc[k]);
        fprintf(stderr, "\n");
        for (hipEvent_t e : ph_ev) (void)hipEventDestroy(e);
    }
    S.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    S.structure_ms = h->structure_ms; S.exchange_ms = h->exchange_ms; S.exchange_calls = h->exchange_calls;
    if (st) *st = S;
    if (rc != UZL_OK) h->last_error = "PCG hit pcg_max_iter in at least one LM trial";
    return rc;
}
}  // namespace uzl
namespace {

}  // namespace

// what the structure of a handle was built from (moved out before a new graph is read in, compared afterwards)
struct StructureKey {
    bool ready = false;
    int32_t n = 0;
    std::vector<uint8_t> fixed_in, robust;
    std::vector<int32_t> ij;
    std::vector<double> edge_w;
};
static StructureKey take_structure_key(uzl_pgo* h)
{
    StructureKey k;
    k.ready = h->structure_ready && h->have_graph;
    if (!k.ready) return k;
    k.n = h->n;
    k.fixed_in.swap(h->fixed_in); k.robust.swap(h->robust); k.ij.swap(h->ij); k.edge_w.swap(h->edge_w);
    return k;
}
static bool same_structure(const uzl_pgo* h, const StructureKey& k)
{
    return k.ready && k.n == h->n && k.fixed_in == h->fixed_in && k.ij == h->ij && k.robust == h->robust && k.edge_w == h->edge_w;
}

// What a handle learned about the numbering of its reduced systems (num_its, build_structure) belongs to the session it came from: it is
// kept while the graph is the previous one, unchanged or GROWN - at least as many nodes, the old nodes' fixed flags in front, AND the old
// system edges still there in their order (nine in ten of them found in order in the new list: an online session appends edges and
// invalidates a few, graph_slam_node.cpp:1138-1150) - and dropped for anything else: an unrelated graph on a reused handle then solves
// as on a fresh one (round 5 looked at the fixed flags only, which for most graphs is "node 0 is fixed").
static bool edges_survive_in_order(const std::vector<int32_t>& old_ij, const std::vector<int32_t>& new_ij)
{
    const size_t n_old = old_ij.size() / 2, n_new = new_ij.size() / 2;
    if (n_old == 0) return true;
    constexpr size_t kWindow = 4096;                                 // how far ahead a surviving edge may have moved
    size_t at = 0, found = 0;
    for (size_t k = 0; k < n_old; k++) {
        const int32_t a = old_ij[2 * k], c = old_ij[2 * k + 1];
        const size_t end = std::min(n_new, at + kWindow);
        for (size_t q = at; q < end; q++)
            if (new_ij[2 * q] == a && new_ij[2 * q + 1] == c) { found++; at = q + 1; break; }
    }
    return 10 * found >= 9 * n_old;
}
static void keep_or_drop_numbering_history(uzl_pgo* h, const StructureKey& k)
{
    if (!k.ready || h->structure_ready) return;                     // nothing learned yet / the same structure again
    const bool grown = h->n >= k.n && (size_t)k.n <= k.fixed_in.size() && std::equal(k.fixed_in.begin(), k.fixed_in.begin() + k.n, h->fixed_in.begin()) &&
                       edges_survive_in_order(k.ij, h->ij);
    if (!grown) { h->num_its[0] = h->num_its[1] = -1.; h->num_last = -1; }
}

#define UZL_GUARD_BEGIN(h)                       \
    if (!(h)) return UZL_ERR_BAD_ARG;            \
    std::lock_guard<std::mutex> lock_((h)->mu);  \
    try {
#define UZL_GUARD_END(h)                                                             \
    } catch (const ::uzl::HipError& e) { return ::uzl::report((h)->last_error, e); } \
    catch (const std::bad_alloc&) { (h)->last_error = "host out of memory"; return UZL_ERR_OOM; } \
    catch (...) { (h)->last_error = "unexpected exception"; return UZL_ERR_HIP; }

// A handle with a stream pair of its own from the device's pool (uzl_pgo_create), or - the graphs of a batch - on the batch's streams: a
// stream costs the runtime ~3.5 ms to make and ~2 ms to destroy (round 4's uzl_pgo_create 6.9 ms, measured: tests/diag/create_cost.py),
// which a batch of 64 graphs paid 128 times over although its solves never use its handles' streams.  A batch's handle takes a pair
// of its own the first time it is solved through uzl_pgo_optimize (own_streams).
static int pgo_create_on(const uzl_pgo_cfg* cfg, hipStream_t shared, hipStream_t shared2, uzl_pgo** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_pgo_cfg c;
    if (cfg) c = *cfg; else uzl_pgo_cfg_default(&c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UZL_ERR_NO_DEVICE;
    if (c.device < 0 || c.device >= ndev) return UZL_ERR_NO_DEVICE;
    uzl_pgo* h = new (std::nothrow) uzl_pgo();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    memset(&h->D, 0, sizeof(h->D));
    { const char* ng = getenv("UZL_NO_GRAPH"); h->no_graph = ng && ng[0] == '1'; }
    if (getenv("UZL_VERBOSE")) h->cfg.verbose = 1;                                    // diagnostic: per-trial PCG log on stderr
    bool ok = hipSetDevice(c.device) == hipSuccess;
    if (ok && shared) { h->stream = shared; h->stream2 = shared2; h->streams_borrowed = true; }
    else if (ok) {
        // The solver's stream and the stream its rebuilds run ahead on (at the higher priority: they overlap with the PCG they are for)
        // come from the device's pool too: a pair that shares a compute pipe costs a config-5 run a quarter of its solver time (two
        // OnlineSlam sessions in one process: 1.34 against 1.67 s of optimize, tests/diag/online_passes.py - the second session's handle
        // had taken its streams as they came).  Streams go back to the pool with the handle: making a handle no longer costs two
        // hipStreamCreates (7 ms) once the pool holds a pair.
        h->stream = stream_lease(c.device, 0, {}, false);
        h->stream2 = h->stream ? stream_lease(c.device, diag_int("UZL_S2_PRIO", -1), {h->stream}, false) : nullptr;
        ok = h->stream && h->stream2;
    }
    ok = ok && hipEventCreateWithFlags(&h->ev_lin, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->ev_setup, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        if (!h->streams_borrowed) { stream_release(c.device, h->stream); stream_release(c.device, h->stream2); }
        if (h->ev_lin) (void)hipEventDestroy(h->ev_lin);
        if (h->ev_setup) (void)hipEventDestroy(h->ev_setup);
        delete h;
        return UZL_ERR_HIP;
    }
    *out = h;
    return UZL_OK;
}
// a batch's handle that is solved on its own: from now on with its own streams (captures and rebuild events of two such handles driven
// from two threads must not meet on one stream).  `drain_borrowed`: wait for the borrowed streams first - right for a handle that is
// taken out of an idle batch (uzl_pgo_optimize), WRONG from inside a running batch: every handle of a batch borrows launch sequence 0's
// streams, sequence 0 may be capturing on them on another thread (a synchronize fails with hipErrorStreamCaptureUnsupported and
// invalidates that capture), and the handle has no work of its own on them - its sequence has synchronized the stream it ran on.
void uzl::own_streams(uzl_pgo* h, bool drain_borrowed)
{
    if (!h->streams_borrowed) return;
    UZL_HIP(hipSetDevice(h->cfg.device));
    if (drain_borrowed) {
        UZL_HIP(hipStreamSynchronize(h->stream));
        if (h->stream2) UZL_HIP(hipStreamSynchronize(h->stream2));
    }
    hipStream_t a = stream_lease(h->cfg.device, 0, {}, false);
    hipStream_t b = a ? stream_lease(h->cfg.device, diag_int("UZL_S2_PRIO", -1), {a}, false) : nullptr;
    if (!a || !b) { stream_release(h->cfg.device, a); throw HipError{hipErrorUnknown, "stream_lease", __FILE__, __LINE__}; }
    h->stream = a; h->stream2 = b; h->streams_borrowed = false;
}

extern "C" {

static_assert(sizeof(uzl_pgo_cfg) == 64, "uzl_pgo_cfg::pass_history occupies the former tail padding: the layout of ABI version 3 is unchanged");

void uzl_pgo_cfg_default(uzl_pgo_cfg* cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->iterations = 20;               // cfg/GraphOptimizer.cfg:10
    cfg->use_odometry_parameters = 0;   // :11
    cfg->optimize_xy_only = 0;          // :12
    cfg->device = 0;
    // relative M^-1-norm residual.  g2o's own LinearSolverPCG stops at 1e-6 on the SQUARED norm (1e-3 here) [EXT];
    // 1e-5 keeps the result within ~1e-5 m / 1e-6 rad of the direct (CSparse-like) solve at every LM iteration
    // count (measured deviation scales linearly with this value)
    cfg->pcg_tol = 1e-5;
    cfg->pcg_max_iter = 0;              // 0 = 6 * free vertices (system dimension)
    cfg->schur_reduce = 0;              // 0 = Schur-eliminate chain interiors when a third of the free vertices are (pgo_schur.hpp); -1 = never
    cfg->huber_delta = 1.0;             // g2o_optimizer.cpp:293
    cfg->verbose = 0;
    cfg->preconditioner = 1;            // additive multilevel (rigid-body-mode aggregation); 0 = block-Jacobi
    cfg->pcg_stop = 0;                  // step-error estimate; 1 = relative residual test only
    cfg->lm_loop = 0;                   // LM decisions on the device (captured passes); 1 = host-driven loop
    cfg->reduced_numbering = 0;         // the handle chooses between row order and strong aggregates (pgo_schur.hpp)
    cfg->pass_history = 0;              // pass sizes of a repeated optimize may come from the previous one's per-trial counts; 1 = never
}

int uzl_pgo_create(const uzl_pgo_cfg* cfg, uzl_pgo** out) { return pgo_create_on(cfg, nullptr, nullptr, out); }

void uzl_pgo_destroy(uzl_pgo* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    destroy_pcg_graph(h);
    lm_run_destroy(h->lm); h->lm = nullptr;
    if (h->rccl_comm) { (void)rccl().CommDestroy(h->rccl_comm); h->rccl_comm = nullptr; }
    if (h->ev_lin) (void)hipEventDestroy(h->ev_lin);
    if (h->ev_setup) (void)hipEventDestroy(h->ev_setup);
    if (!h->streams_borrowed) { stream_release(h->cfg.device, h->stream2); stream_release(h->cfg.device, h->stream); }      // back to the pool, verdicts kept
    delete h;
}

int uzl_pgo_set_config(uzl_pgo* h, const uzl_pgo_cfg* cfg)
{
    if (!h || !cfg) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (cfg->device != h->cfg.device) return fail(h, UZL_ERR_BAD_ARG, "device cannot change after create");
    if (cfg->iterations < 1 || cfg->pcg_tol <= 0. || cfg->huber_delta <= 0.) return fail(h, UZL_ERR_BAD_ARG, "bad config value");
    if (cfg->pcg_tol != h->cfg.pcg_tol) destroy_pcg_graph(h);      // the tolerance is a captured kernel argument
    if (cfg->preconditioner != h->cfg.preconditioner || cfg->schur_reduce != h->cfg.schur_reduce || cfg->reduced_numbering != h->cfg.reduced_numbering) h->structure_ready = false;
    h->cfg = *cfg;
    return UZL_OK;
}

const char* uzl_pgo_last_error(uzl_pgo* h) { return h ? h->last_error.c_str() : "null handle"; }

int uzl_pgo_add_graph(uzl_pgo* h, int32_t n_nodes, const uzl_node* nodes, int32_t n_edges, const uzl_edge* edges,
                      int32_t n_sensors, const double* sensors)
{
    UZL_GUARD_BEGIN(h)
    if (n_nodes < 0 || n_edges < 0 || n_sensors < 0 || (n_nodes > 0 && !nodes) || (n_edges > 0 && !edges) ||
        (n_sensors > 0 && !sensors))
        return fail(h, UZL_ERR_BAD_ARG, "null or negative-size input");
    UZL_HIP(hipSetDevice(h->cfg.device));
    // clear() (:57): the reference rebuilds everything.  Here the structure of the previous graph (gauge, block-CSR, aggregation order,
    // hierarchy arrays, captured PCG graph) is kept when the new graph has the same vertices, fixed flags, system edges and edge
    // weights - a timer-driven re-optimisation of an unchanged graph, or one whose poses / measurements only moved - and rebuilt
    // otherwise; either way the solve is the one a fresh handle would run (same order, same operators).
    StructureKey old_key = take_structure_key(h);
    // what the handle learned about the numbering of ITS reduced systems (num_its, build_structure) belongs to the session it came from: a
    // graph that is not the previous one grown (fewer nodes than before) starts as on a fresh handle
    if (n_nodes < h->n) { h->num_its[0] = h->num_its[1] = -1.; h->num_last = -1; }
    h->have_graph = false; h->structure_ready = false;
    h->n = n_nodes; h->e_in = n_edges;
    h->fixed_in.assign((size_t)n_nodes, 0);
    for (int v = 0; v < n_nodes; v++) h->fixed_in[v] = nodes[v].fixed ? 1 : 0;
    h->in_edges.resize((size_t)n_edges);
    for (int k = 0; k < n_edges; k++) h->in_edges[k] = in_edge_of(edges[k]);
    h->in_ready = true; h->n_sensors_in = n_sensors;
    flatten_edges(h);
    alloc_problem(h);
    hipStream_t s = h->stream;
    h->d_nodes.reserve((size_t)std::max(n_nodes, 1));
    h->d_edges.reserve((size_t)std::max(n_edges, 1));
    h->d_src.reserve((size_t)std::max(h->e, 1));
    h->d_stage.reserve((size_t)std::max(n_sensors, 1) * 12);
    if (n_nodes) UZL_HIP(hipMemcpyAsync(h->d_nodes.p, nodes, sizeof(uzl_node) * (size_t)n_nodes, hipMemcpyHostToDevice, s));
    if (n_edges) UZL_HIP(hipMemcpyAsync(h->d_edges.p, edges, sizeof(uzl_edge) * (size_t)n_edges, hipMemcpyHostToDevice, s));
    if (n_sensors) UZL_HIP(hipMemcpyAsync(h->d_stage.p, sensors, sizeof(double) * 12 * (size_t)n_sensors, hipMemcpyHostToDevice, s));
    if (h->e) UZL_HIP(hipMemcpyAsync(h->d_src.p, h->src.data(), sizeof(int32_t) * (size_t)h->e, hipMemcpyHostToDevice, s));
    k_prepare_nodes(h->d_nodes.p, n_nodes, h->cfg.optimize_xy_only, h->cur, s);
    k_prepare_edges(h->d_edges.p, h->d_src.p, h->e, h->d_stage.p, n_sensors, h->cfg.optimize_xy_only, h->cfg.use_odometry_parameters,
                    h->d_zinv.p, h->d_info.p, s);
    UZL_HIP(hipGetLastError());
    if (n_nodes) UZL_HIP(hipMemcpyAsync(h->pose_init.p, h->cur, sizeof(double) * 8 * (size_t)n_nodes, hipMemcpyDeviceToDevice, s));
    UZL_HIP(hipStreamSynchronize(s));                          // inputs are borrowed for the duration of the call only
    upload_edges_common(h);
    h->have_graph = true;
    h->structure_ready = same_structure(h, old_key);
    keep_or_drop_numbering_history(h, old_key);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_append_graph(uzl_pgo* h, int32_t n_new_nodes, const uzl_node* new_nodes, int32_t n_new_edges, const uzl_edge* new_edges,
                         int32_t n_flags, const int32_t* edge_index, const uint8_t* edge_valid)
{
    UZL_GUARD_BEGIN(h)
    if (n_new_nodes < 0 || n_new_edges < 0 || n_flags < 0 || (n_new_nodes > 0 && !new_nodes) || (n_new_edges > 0 && !new_edges) ||
        (n_flags > 0 && (!edge_index || !edge_valid)))
        return fail(h, UZL_ERR_BAD_ARG, "null or negative-size input");
    if (!h->have_graph || !h->in_ready) return fail(h, UZL_ERR_BAD_ARG, "uzl_pgo_append_graph needs a graph from uzl_pgo_add_graph to grow");
    const int32_t n_old = h->n, e_old = h->e_in;
    for (int q = 0; q < n_flags; q++) if (edge_index[q] < 0 || edge_index[q] >= e_old) return fail(h, UZL_ERR_BAD_ARG, "edge index out of range");
    UZL_HIP(hipSetDevice(h->cfg.device));
    StructureKey old_key = take_structure_key(h);
    h->have_graph = false; h->structure_ready = false;
    h->n = n_old + n_new_nodes; h->e_in = e_old + n_new_edges;
    if (old_key.ready) { h->fixed_in = old_key.fixed_in; }               // (the key took the vectors; the old flags are the first n_old of the new)
    for (int v = 0; v < n_new_nodes; v++) h->fixed_in.push_back(new_nodes[v].fixed ? 1 : 0);
    for (int q = 0; q < n_flags; q++) h->in_edges[edge_index[q]].valid = edge_valid[q] ? 1 : 0;
    for (int k = 0; k < n_new_edges; k++) h->in_edges.push_back(in_edge_of(new_edges[k]));
    flatten_edges(h);
    hipStream_t s = h->stream;
    alloc_problem(h, true);                                             // the estimates of the old nodes stay where the last solve left them
    h->d_nodes.reserve((size_t)std::max(n_new_nodes, 1));
    h->d_edges.reserve((size_t)std::max(h->e_in, 1), true, s);
    h->d_src.reserve((size_t)std::max(h->e, 1));
    if (n_new_nodes) UZL_HIP(hipMemcpyAsync(h->d_nodes.p, new_nodes, sizeof(uzl_node) * (size_t)n_new_nodes, hipMemcpyHostToDevice, s));
    if (n_new_edges) UZL_HIP(hipMemcpyAsync(h->d_edges.p + e_old, new_edges, sizeof(uzl_edge) * (size_t)n_new_edges, hipMemcpyHostToDevice, s));
    if (h->e) UZL_HIP(hipMemcpyAsync(h->d_src.p, h->src.data(), sizeof(int32_t) * (size_t)h->e, hipMemcpyHostToDevice, s));
    if (n_new_nodes) k_prepare_nodes(h->d_nodes.p, n_new_nodes, h->cfg.optimize_xy_only, h->cur + (size_t)n_old * 8, s);
    k_prepare_edges(h->d_edges.p, h->d_src.p, h->e, h->d_stage.p, h->n_sensors_in, h->cfg.optimize_xy_only, h->cfg.use_odometry_parameters,
                    h->d_zinv.p, h->d_info.p, s);
    UZL_HIP(hipGetLastError());
    if (h->n) UZL_HIP(hipMemcpyAsync(h->pose_init.p, h->cur, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
    UZL_HIP(hipStreamSynchronize(s));                          // inputs are borrowed for the duration of the call only
    upload_edges_common(h);
    h->have_graph = true;
    h->structure_ready = same_structure(h, old_key);
    keep_or_drop_numbering_history(h, old_key);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_set_graph(uzl_pgo* h, int32_t n, const double* poses, const uint8_t* fixed, int32_t e,
                      const int32_t* ij, const double* meas, const double* info, const uint8_t* robust)
{
    UZL_GUARD_BEGIN(h)
    if (n < 0 || e < 0 || (n > 0 && (!poses || !fixed)) || (e > 0 && (!ij || !meas || !info || !robust)))
        return fail(h, UZL_ERR_BAD_ARG, "null or negative-size input");
    for (int k = 0; k < e; k++)
        if (ij[2 * k] < 0 || ij[2 * k] >= n || ij[2 * k + 1] < 0 || ij[2 * k + 1] >= n || ij[2 * k] == ij[2 * k + 1])
            return fail(h, UZL_ERR_BAD_ARG, "edge endpoint out of range");
    UZL_HIP(hipSetDevice(h->cfg.device));
    StructureKey old_key = take_structure_key(h);
    if (n < h->n) { h->num_its[0] = h->num_its[1] = -1.; h->num_last = -1; }      // (as uzl_pgo_add_graph)
    h->have_graph = false; h->structure_ready = false;
    h->n = n; h->e_in = e; h->e = e;
    h->in_ready = false; h->in_edges.clear();                            // (d_stage, the sensors' place, is this call's staging area)
    h->fixed_in.assign(fixed, fixed + n);
    for (auto& f : h->fixed_in) f = f ? 1 : 0;
    h->ij.assign(ij, ij + 2 * (size_t)e);
    h->robust.assign(robust, robust + e);
    h->src.resize((size_t)e);
    std::iota(h->src.begin(), h->src.end(), 0);
    h->edge_w.resize((size_t)e);
    for (int k = 0; k < e; k++) { double tr = 0.; for (int r = 0; r < 6; r++) tr += info[36 * (size_t)k + r * 7]; h->edge_w[k] = tr; }
    alloc_problem(h);
    hipStream_t s = h->stream;
    const size_t stage = (size_t)std::max(n, 1) * 12 + (size_t)std::max(e, 1) * 48;
    h->d_stage.reserve(stage);
    double* d_p = h->d_stage.p;
    double* d_m = d_p + (size_t)std::max(n, 1) * 12;
    double* d_i = d_m + (size_t)std::max(e, 1) * 12;
    if (n) UZL_HIP(hipMemcpyAsync(d_p, poses, sizeof(double) * 12 * (size_t)n, hipMemcpyHostToDevice, s));
    if (e) {
        UZL_HIP(hipMemcpyAsync(d_m, meas, sizeof(double) * 12 * (size_t)e, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(d_i, info, sizeof(double) * 36 * (size_t)e, hipMemcpyHostToDevice, s));
    }
    k_prepare_flat_nodes(d_p, n, h->cur, s);
    k_prepare_flat_edges(d_m, d_i, e, h->d_zinv.p, h->d_info.p, s);
    UZL_HIP(hipGetLastError());
    if (n) UZL_HIP(hipMemcpyAsync(h->pose_init.p, h->cur, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToDevice, s));
    UZL_HIP(hipStreamSynchronize(s));
    upload_edges_common(h);
    h->have_graph = true;
    h->structure_ready = same_structure(h, old_key);
    keep_or_drop_numbering_history(h, old_key);
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_reset(uzl_pgo* h)
{
    UZL_GUARD_BEGIN(h)
    if (!h->have_graph) return fail(h, UZL_ERR_STATE, "reset before add_graph/set_graph");
    UZL_HIP(hipSetDevice(h->cfg.device));
    if (h->n) UZL_HIP(hipMemcpyAsync(h->cur, h->pose_init.p, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* stats)
{
    UZL_GUARD_BEGIN(h)
    own_streams(h, true);
    return do_optimize(h, iterations, stats);
    UZL_GUARD_END(h)
}

int uzl_pgo_store(uzl_pgo* h, double* poses, double* edge_error, uint8_t* edge_in_system)
{
    UZL_GUARD_BEGIN(h)
    if (!h->have_graph) return fail(h, UZL_ERR_STATE, "store before add_graph/set_graph");
    UZL_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;
    if (poses && h->n > 0) {
        h->d_out12.reserve((size_t)h->n * 12);
        k_poses_out(h->cur, h->n, h->d_out12.p, s);                                   // :110-117
        UZL_HIP(hipMemcpyAsync(poses, h->d_out12.p, sizeof(double) * 12 * (size_t)h->n, hipMemcpyDeviceToHost, s));
    }
    std::vector<double> err;
    if (edge_error && h->e > 0) {
        if (!h->structure_ready) {      // store without optimize: same structure optimize would build
            h->fixed_eff = h->fixed_in;
            h->n_gauge = gauge_fix(h);
            build_structure(h);
        }
        h->d_err.reserve((size_t)h->e);
        k_edge_error(h->D, h->cur, h->d_err.p, s);                                    // :124-131
        err.resize((size_t)h->e);
        UZL_HIP(hipMemcpyAsync(err.data(), h->d_err.p, sizeof(double) * (size_t)h->e, hipMemcpyDeviceToHost, s));
    }
    UZL_HIP(hipGetLastError());
    UZL_HIP(hipStreamSynchronize(s));
    if (edge_error) {
        for (int k = 0; k < h->e_in; k++) edge_error[k] = std::numeric_limits<double>::quiet_NaN();
        for (int k = 0; k < h->e; k++) edge_error[h->src[k]] = err[k];
    }
    if (edge_in_system) {
        memset(edge_in_system, 0, (size_t)h->e_in);
        for (int k = 0; k < h->e; k++) edge_in_system[h->src[k]] = 1;
    }
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_get_fixed(uzl_pgo* h, uint8_t* fixed)
{
    if (!h || !fixed) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->have_graph) return fail(h, UZL_ERR_STATE, "no graph");
    const std::vector<uint8_t>& f = h->fixed_eff.size() == (size_t)h->n ? h->fixed_eff : h->fixed_in;
    if (h->n) memcpy(fixed, f.data(), (size_t)h->n);
    return UZL_OK;
}

int uzl_pgo_set_profiling(uzl_pgo* h, int32_t on)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->timer.on = on != 0;
    return UZL_OK;
}

int uzl_pgo_kernel_times(uzl_pgo* h, int32_t cap, const char** names, double* ms, int32_t* launches)
{
    if (!h || cap < 0 || (cap > 0 && (!names || !ms || !launches))) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return h->timer.report(cap, names, ms, launches);
}

int uzl_pgo_set_shard(uzl_pgo* h, int32_t rank, int32_t world_size, uzl_allreduce_fn allreduce, void* user)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(h, UZL_ERR_BAD_ARG, "bad rank/world_size");
    if (world_size > 1 && !allreduce) return fail(h, UZL_ERR_BAD_ARG, "world_size > 1 needs an all-reduce callback");
    if (h->rccl_comm) { (void)hipStreamSynchronize(h->stream); (void)rccl().CommDestroy(h->rccl_comm); h->rccl_comm = nullptr; }
    h->rank = rank; h->world = world_size; h->allreduce = allreduce; h->allreduce_user = user;
    h->structure_ready = false;
    return UZL_OK;
}

int uzl_rccl_unique_id(void* id_out, int32_t cap)
{
    if (!id_out || cap < UZL_RCCL_UNIQUE_ID_BYTES) return UZL_ERR_BAD_ARG;
    RcclApi& R = rccl();
    if (!R.ok) return UZL_ERR_STATE;
    RcclApi::UniqueId id;
    if (R.GetUniqueId(&id) != 0) return UZL_ERR_HIP;
    memcpy(id_out, id.internal, UZL_RCCL_UNIQUE_ID_BYTES);
    return UZL_OK;
}

int uzl_pgo_set_shard_rccl(uzl_pgo* h, int32_t rank, int32_t world_size, const void* unique_id, int32_t id_bytes)
{
    UZL_GUARD_BEGIN(h)
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(h, UZL_ERR_BAD_ARG, "bad rank/world_size");
    if (!unique_id || id_bytes != UZL_RCCL_UNIQUE_ID_BYTES) return fail(h, UZL_ERR_BAD_ARG, "unique id must be UZL_RCCL_UNIQUE_ID_BYTES bytes");
    RcclApi& R = rccl();
    if (!R.ok) { h->last_error = R.error; return UZL_ERR_STATE; }
    UZL_HIP(hipSetDevice(h->cfg.device));
    UZL_HIP(hipStreamSynchronize(h->stream));
    if (h->rccl_comm) { (void)R.CommDestroy(h->rccl_comm); h->rccl_comm = nullptr; }
    RcclApi::UniqueId id;
    memcpy(id.internal, unique_id, UZL_RCCL_UNIQUE_ID_BYTES);
    RcclApi::Comm comm = nullptr;
    const int rc = R.CommInitRank(&comm, world_size, id, rank);               // collective over the ranks
    if (rc != 0 || !comm) {
        h->last_error = std::string("ncclCommInitRank failed: ") + (R.GetErrorString ? R.GetErrorString(rc) : "?");
        return UZL_ERR_HIP;
    }
    h->rccl_comm = comm;
    h->rank = rank; h->world = world_size; h->allreduce = nullptr; h->allreduce_user = nullptr;
    h->structure_ready = false;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_rccl_ranks(uzl_pgo* h)
{
    if (!h) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->rccl_comm) return 0;
    RcclApi& R = rccl();
    int n = 0;
    if (!R.CommCount || R.CommCount(h->rccl_comm, &n) != 0) return UZL_ERR_HIP;
    return n;
}

}  // extern "C"

// =====================================================================================================================
//  Batched solve: B independent graphs through ONE launch sequence (uzl_pgo_batch_*)
//
//  One config-2-sized graph leaves the chip ~97 % idle (125 workgroups per launch, two dependent launches per PCG iteration);
//  several handles on several streams do not recover it (tests/diag/multi_handle.py: 4 handles 1.9x, 16 handles 1.5x - the
//  launches serialise).  Here every kernel of the solve is launched once for all graphs (blockIdx.z = slot, arguments from a slot
//  table, each graph's Levenberg-Marquardt state on the device: uzl_pgo_lm.hip), so a PCG iteration of B graphs costs two launches,
//  like one graph's.  The kernels are the single-graph kernels' bodies and every graph's loop takes its own decisions from its own
//  state: poses, chi2 and iteration counts are bit-identical to a uzl_pgo_optimize of that graph alone.  Batched together are
//  graphs of the small-graph class (dense level-1 operator) with one hierarchy shape of the system the PCG solves - full or
//  Schur-reduced: chain-like graphs batch on their reduced systems; anything else - and any graph that meets an anomaly (PCG not
//  converged, breakdown) - is solved by the single-graph path, so results never depend on whether a graph was batched.
// =====================================================================================================================
struct uzl_pgo_batch {
    std::mutex mu;
    std::string last_error;
    uzl_pgo_cfg cfg;
    std::vector<uzl_pgo*> h;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // rebuilds that run ahead (into the hierarchy copy the PCG does not use), beside this iteration's solves
    int32_t resident = 0;                 // graphs solved at a time (0 = all): uzl_pgo_batch_set_resident
    int32_t last_batched = 0;
    uzl::LmRun* lm = nullptr;             // slot table, LM states, captured segments (uzl_pgo_lm.hip)
    // second launch sequence (batches of >= kBatchLaneMin graphs): the second half of the graphs on streams of its own, driven by a second
    // host thread for the duration of the call - its kernels fill the tails of the first half's and a pass is as long as the longest solve
    // of eight graphs, not sixteen.  Only with four streams that do not stand in each other's way (uzl_pgo_batch_create).
    hipStream_t stream_b = nullptr, stream2_b = nullptr;
    uzl::LmRun* lm_b = nullptr;
    // diagnostic build, UZL_BATCH_LANES=4: four sequences, one per compute pipe, each with its rebuilds on its own stream
    hipStream_t stream_x[2] = {nullptr, nullptr};
    uzl::LmRun* lm_x[2] = {nullptr, nullptr};
    KernelTimer timer;                    // profiling (uzl_pgo_batch_set_profiling): the two PCG kernels, launched eagerly with event pairs
};

namespace {

int bfail(uzl_pgo_batch* b, int code, const char* msg) { b->last_error = msg; return code; }

int batch_optimize(uzl_pgo_batch* b, int32_t iterations, uzl_pgo_stats* stats, int32_t* n_batched)
{
    const int B = (int)b->h.size();
    if (n_batched) *n_batched = 0;
    for (uzl_pgo* h : b->h) if (!h->have_graph) return bfail(b, UZL_ERR_STATE, "optimize before every graph of the batch has been set");
    UZL_HIP(hipSetDevice(b->cfg.device));
    // per graph: optimizeImpl's initializeOptimization + setFixedNodes (:139-146), structure
    for (int g = 0; g < B; g++) {
        uzl_pgo* h = b->h[g];
        if (iterations <= 0) iterations = h->cfg.iterations;
        prepare_optimize(h);
        UZL_HIP(hipStreamSynchronize(h->stream));
    }
    int rc_all = UZL_OK;
    if (!lm_batch_eligible(b->h)) {      // not one class / one shape: every graph through the single-graph path
        if (b->cfg.verbose)
            for (const uzl_pgo* h : b->h)
                fprintf(stderr, "[uzl_pgo_batch] not batched: levels %d agg %d comp %d mult %d cl %d sharded %d nb %d (pcg system %d) e %d timer %d n1 %d ns %d\n", h->ml_levels,
                        h->ml_agg, (int)h->ml_comp, (int)h->ml_mult, h->ml_cl, (int)h->sharded, h->nb, h->Dp.nb, h->e, (int)h->timer.on, h->ml_n.size() > 1 ? h->ml_n[1] : -1, h->ml_ns_steps);
        for (int g = 0; g < B; g++) {
            uzl_pgo* h = b->h[g];
            h->t_start = std::chrono::steady_clock::now();
            uzl_pgo_stats S;
            const int rc = lm_eligible(h) ? do_optimize_lm(h, iterations, &S) : do_optimize_host(h, iterations, &S);
            if (rc != UZL_OK && rc != UZL_ERR_NOT_CONVERGED) { b->last_error = h->last_error; return rc; }
            if (rc != UZL_OK) rc_all = rc;
            if (stats) stats[g] = S;
        }
        b->last_batched = 0;
        return rc_all;
    }
    static const int lanes_env = diag_int("UZL_BATCH_LANES", 2);               // A/B switch (diagnostic build): 1 = one launch sequence
    const bool eager = b->h[0]->no_graph, verbose = b->cfg.verbose != 0;
    static const bool no_s2 = diag_flag("UZL_BATCH_NO_S2");                   // A/B switch: rebuilds on the sequence's own stream
    hipStream_t s2a = no_s2 ? b->stream : b->stream2, s2b = no_s2 ? b->stream_b : b->stream2_b;
    auto one_sequence = [&]() {
        const int done = batch_optimize_lm(b->lm, b->h, b->resident, b->stream, s2a, iterations, eager, verbose, &b->timer, stats, &rc_all);
        if (done < 0) { b->last_error = b->h[(size_t)(-1 - done)]->last_error; return rc_all; }
        b->last_batched = done;
        if (n_batched) *n_batched = done;
        return rc_all;
    };
    static const int lane_min = diag_int("UZL_BATCH_LANE_MIN", kBatchLaneMin);      // A/B switch (diagnostic build)
    if (B < lane_min || lanes_env < 2 || b->resident == 1 || b->timer.on || !b->stream_b || !b->stream2_b) return one_sequence();
    // ---- L launch sequences: the graphs in L runs, the first from this thread, the others from helper threads.  The sequences share
    //      nothing but the device (every graph has its handle, every sequence its streams, slot table and captured segments), and a graph's
    //      result does not depend on its neighbours in the batch, so the split changes no bit of any result.
    const bool four = lanes_env >= 4 && b->stream_x[0] && b->stream_x[1] && B >= 4 * (lane_min / 2) && (b->resident == 0 || b->resident >= 4);
    const int L = four ? 4 : 2;
    struct Lane { uzl::LmRun** lm; hipStream_t s, s2; int first, count, resident, rc, done; std::exception_ptr ex; };
    std::vector<Lane> lanes((size_t)L);
    {
        uzl::LmRun** lms[4] = {&b->lm, &b->lm_b, &b->lm_x[0], &b->lm_x[1]};
        hipStream_t ss[4] = {b->stream, b->stream_b, b->stream_x[0], b->stream_x[1]};
        hipStream_t s2s[4] = {s2a, s2b, b->stream_x[0], b->stream_x[1]};
        int first = 0, res_left = b->resident;
        for (int l = 0; l < L; l++) {
            const int count = (B - first + (L - l) - 1) / (L - l);
            const int res = b->resident > 0 ? (res_left + (L - l) - 1) / (L - l) : 0;
            lanes[(size_t)l] = Lane{lms[l], ss[l], four ? ss[l] : s2s[l], first, count, res, UZL_OK, 0, nullptr};
            first += count; res_left -= res;
        }
    }
    auto run_lane = [&](Lane& ln) {
        try {
            UZL_HIP(hipSetDevice(b->cfg.device));
            const std::vector<uzl_pgo*> hs(b->h.begin() + ln.first, b->h.begin() + ln.first + ln.count);
            ln.done = batch_optimize_lm(*ln.lm, hs, ln.resident, ln.s, ln.s2, iterations, eager, verbose, nullptr, stats ? stats + ln.first : nullptr, &ln.rc);
        } catch (...) { ln.ex = std::current_exception(); }
    };
    std::vector<std::thread> helpers;
    try {
        for (int l = 1; l < L; l++) helpers.emplace_back([&, l] { run_lane(lanes[(size_t)l]); });
    } catch (const std::system_error&) {                                       // (no more threads to be had: the sequences not started run from this one)
        for (int l = (int)helpers.size() + 1; l < L; l++) run_lane(lanes[(size_t)l]);
    }
    run_lane(lanes[0]);
    for (std::thread& t : helpers) t.join();
    for (Lane& ln : lanes) if (ln.ex) std::rethrow_exception(ln.ex);
    int total = 0;
    for (Lane& ln : lanes) {
        if (ln.done < 0) { b->last_error = b->h[(size_t)(ln.first - 1 - ln.done)]->last_error; return ln.rc; }
        if (ln.rc != UZL_OK && rc_all == UZL_OK) rc_all = ln.rc;
        total += ln.done;
    }
    b->last_batched = total;
    if (n_batched) *n_batched = total;
    return rc_all;
}

}  // namespace

#define UZL_BGUARD_BEGIN(b)                      \
    if (!(b)) return UZL_ERR_BAD_ARG;            \
    std::lock_guard<std::mutex> lock_((b)->mu);  \
    try {
#define UZL_BGUARD_END(b)                                                            \
    } catch (const ::uzl::HipError& e) { return ::uzl::report((b)->last_error, e); } \
    catch (const std::bad_alloc&) { (b)->last_error = "host out of memory"; return UZL_ERR_OOM; } \
    catch (...) { (b)->last_error = "unexpected exception"; return UZL_ERR_HIP; }

extern "C" {

int uzl_pgo_batch_create(const uzl_pgo_cfg* cfg, int32_t n_graphs, uzl_pgo_batch** out)
{
    if (!out || n_graphs < 1 || n_graphs > kBatchMax) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_pgo_cfg c;
    if (cfg) c = *cfg; else uzl_pgo_cfg_default(&c);
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c.device < 0 || c.device >= ndev) return UZL_ERR_NO_DEVICE;
    }
    uzl_pgo_batch* b = new (std::nothrow) uzl_pgo_batch();
    if (!b) return UZL_ERR_OOM;
    b->cfg = c;
    if (getenv("UZL_VERBOSE")) b->cfg.verbose = 1;
    // The batch's streams come from the device's pool (uzl_streams.hip): none in another's way.  Batches of kBatchLaneMin graphs and more
    // get a second launch sequence if four such streams can be had within the pool's budget - otherwise, and with UZL_STREAM_PROBE=0,
    // the batch runs as ONE launch sequence whatever streams it got; a rebuild stream that cannot be had apart from the solver's is
    // replaced by any stream (slower, not wrong).
    static const int prio2 = diag_int("UZL_BATCH_S2_PRIO", 0);                     // A/B switches (diagnostic build)
    static const bool two_on = diag_int("UZL_BATCH_LANES", 2) >= 2;
    const int dev = c.device;
    b->stream = stream_lease(dev, 0, {}, false);
    bool ok = b->stream != nullptr;
    if (ok && two_on && n_graphs >= diag_int("UZL_BATCH_LANE_MIN", kBatchLaneMin)) b->stream_b = stream_lease(dev, 0, {b->stream}, true);
    if (ok && b->stream_b && diag_int("UZL_BATCH_LANES", 2) >= 4) {             // diagnostic build: two more solver streams, four pipes in all
        b->stream_x[0] = stream_lease(dev, 0, {b->stream, b->stream_b}, true);
        if (b->stream_x[0]) b->stream_x[1] = stream_lease(dev, 0, {b->stream, b->stream_b, b->stream_x[0]}, true);
    }
    const bool four = b->stream_x[1] != nullptr;                // (then the rebuild streams only serve batches too small for four sequences)
    if (ok) {
        b->stream2 = four ? stream_lease(dev, prio2, {b->stream}, false) : stream_lease(dev, prio2, {b->stream, b->stream_b}, false);
        ok = b->stream2 != nullptr;
    }
    if (ok && b->stream_b) {
        b->stream2_b = four ? stream_lease(dev, prio2, {b->stream, b->stream_b}, true) : stream_lease(dev, prio2, {b->stream, b->stream_b, b->stream2}, true);
        if (!b->stream2_b) { stream_release(dev, b->stream_b); b->stream_b = nullptr; }      // no fourth: one sequence
    }
    int rc_h = UZL_OK;
    for (int32_t g = 0; ok && g < n_graphs; g++) {              // the graphs' handles, on the batch's first launch sequence's streams
        uzl_pgo* h = nullptr;
        rc_h = pgo_create_on(&c, b->stream, b->stream2, &h);
        if (rc_h != UZL_OK) { ok = false; break; }
        b->h.push_back(h);
    }
    if (!ok) {
        for (uzl_pgo* x : b->h) uzl_pgo_destroy(x);
        for (hipStream_t q : {b->stream, b->stream2, b->stream_b, b->stream2_b, b->stream_x[0], b->stream_x[1]}) stream_release(dev, q);
        delete b;
        return rc_h != UZL_OK ? rc_h : UZL_ERR_HIP;
    }
    *out = b;
    return UZL_OK;
}

void uzl_pgo_batch_destroy(uzl_pgo_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->cfg.device);
    for (hipStream_t q : {b->stream2, b->stream, b->stream2_b, b->stream_b, b->stream_x[0], b->stream_x[1]}) if (q) (void)hipStreamSynchronize(q);
    lm_run_destroy(b->lm); b->lm = nullptr;
    lm_run_destroy(b->lm_b); b->lm_b = nullptr;
    for (uzl::LmRun*& r : b->lm_x) { lm_run_destroy(r); r = nullptr; }
    for (uzl_pgo* x : b->h) uzl_pgo_destroy(x);
    for (hipStream_t q : {b->stream, b->stream2, b->stream_b, b->stream2_b, b->stream_x[0], b->stream_x[1]}) stream_release(b->cfg.device, q);      // back to the pool, verdicts kept
    delete b;
}

const char* uzl_pgo_batch_last_error(uzl_pgo_batch* b) { return b ? b->last_error.c_str() : "null handle"; }
int uzl_pgo_batch_size(uzl_pgo_batch* b) { return b ? (int)b->h.size() : UZL_ERR_BAD_ARG; }
uzl_pgo* uzl_pgo_batch_graph(uzl_pgo_batch* b, int32_t i) { return (b && i >= 0 && i < (int32_t)b->h.size()) ? b->h[(size_t)i] : nullptr; }

int uzl_pgo_batch_set_resident(uzl_pgo_batch* b, int32_t n_resident)
{
    if (!b || n_resident < 0) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(b->mu);
    b->resident = n_resident;
    return UZL_OK;
}

int uzl_pgo_batch_set_profiling(uzl_pgo_batch* b, int32_t on)
{
    if (!b) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(b->mu);
    b->timer.on = on != 0;
    return UZL_OK;
}

int uzl_pgo_batch_kernel_times(uzl_pgo_batch* b, int32_t cap, const char** names, double* ms, int32_t* launches)
{
    if (!b || cap < 0 || (cap > 0 && (!names || !ms || !launches))) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(b->mu);
    return b->timer.report(cap, names, ms, launches);
}

int uzl_pgo_batch_optimize(uzl_pgo_batch* b, int32_t iterations, uzl_pgo_stats* stats, int32_t* n_batched)
{
    UZL_BGUARD_BEGIN(b)
    return batch_optimize(b, iterations, stats, n_batched);
    UZL_BGUARD_END(b)
}

}  // extern "C"
