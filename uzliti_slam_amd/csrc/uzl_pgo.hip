// uzl_pgo.hip — host side of the pose-graph half: graph flattening bookkeeping (skip rules), gauge
// fixing, block-CSR structure, the Levenberg-Marquardt driver and the C ABI (uzl_pgo_*).
// Mirrors G2oOptimizer (graph_optimization/src/g2o_optimizer.cpp:55-349): add_graph = addGraphImpl,
// optimize = optimizeImpl (initializeOptimization + setFixedNodes + optimize(iterations)),
// store = storeImpl.  All arithmetic on poses, measurements and the normal equations runs in the HIP
// kernels of pgo_kernels.hip; the host only makes the integer/structural decisions and the scalar LM
// accept/reject logic of g2o's OptimizationAlgorithmLevenberg [EXT].
#include "uzl_common.hpp"
#include "pgo_types.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <new>
#include <numeric>

namespace uzl {
void k_prepare_nodes(const uzl_node* nodes, int n, int xy, double* pose, hipStream_t s);
void k_prepare_flat_nodes(const double* poses12, int n, double* pose, hipStream_t s);
void k_prepare_edges(const uzl_edge* edges, const int32_t* src, int e, const double* sensors, int ns, int xy,
                     double* zinv, double* info, hipStream_t s);
void k_prepare_flat_edges(const double* meas12, const double* info36, int e, double* zinv, double* info, hipStream_t s);
int k_chi2(const PgoDev& D, const double* pose, double delta, hipStream_t s);
int k_linearize(const PgoDev& D, const double* pose, double delta, hipStream_t s);
int k_assemble(const PgoDev& D, hipStream_t s);
void k_finalize(const PgoDev& D, int na, int nb_, int nc, int what, hipStream_t s);
void k_precond(const PgoDev& D, hipStream_t s);
int k_pcg_init(const PgoDev& D, hipStream_t s);
void k_pcg_p(const PgoDev& D, int n_part, int first, double tol2, hipStream_t s);
int k_pcg_spmv(const PgoDev& D, hipStream_t s);
int k_pcg_update(const PgoDev& D, int n_part, hipStream_t s);
int k_oplus(const PgoDev& D, const double* pose_in, double* pose_out, hipStream_t s);
void k_edge_error(const PgoDev& D, const double* pose, double* err, hipStream_t s);
void k_poses_out(const double* pose, int n, double* out12, hipStream_t s);
}  // namespace uzl

using namespace uzl;

struct uzl_pgo {
    std::mutex mu;
    std::string last_error;
    uzl_pgo_cfg cfg;
    hipStream_t stream = nullptr;
    // ---- host-side structure of the current problem
    int32_t n = 0, e_in = 0, e = 0, nb = 0, nslots = 0;
    std::vector<uint8_t> fixed_in, fixed_eff;
    std::vector<int32_t> ij;       // system edges, 2 per edge
    std::vector<int32_t> src;      // system edge -> input edge
    std::vector<uint8_t> robust;
    bool have_graph = false, structure_ready = false;
    int32_t n_gauge = 0;
    // ---- device
    DevBuf<double> pose_a, pose_b;
    double* cur = nullptr;
    double* trial = nullptr;
    DevBuf<int32_t> d_v2b, d_b2v, d_ei, d_ej, d_slot_i, d_slot_j, d_row_ptr, d_col, d_src, d_flags;
    DevBuf<double> d_zinv, d_info, d_blk, d_dcon, d_gcon, d_hdiag, d_minv, d_b, d_x, d_r, d_z, d_p, d_ap;
    DevBuf<double> d_part_a, d_part_b, d_part_c, d_scal, d_err, d_out12, d_stage;
    DevBuf<uint8_t> d_robust;
    DevBuf<uzl_node> d_nodes;
    DevBuf<uzl_edge> d_edges;
    PinBuf<PgoHostScal> h_scal;
    PinBuf<double> h_lambda;
    PgoDev D;
    int prev_pcg_iters = 0;
    // shard (BASELINE config 4)
    int32_t rank = 0, world = 1;
    uzl_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    KernelTimer timer;
};

namespace {

int fail(uzl_pgo* h, int code, const char* msg)
{
    h->last_error = msg;
    return code;
}

struct Timed {
    uzl_pgo* h;
    Timed(uzl_pgo* h_, const char* name) : h(h_) { h->timer.begin(name, h->stream); }
    ~Timed() { h->timer.end(h->stream); }
};

void fetch_scal(uzl_pgo* h)
{
    UZL_HIP(hipMemcpyAsync(h->h_scal.p->scal, h->D.scal, sizeof(double) * 8, hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipMemcpyAsync(h->h_scal.p->flags, h->D.flags, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, h->stream));
    UZL_HIP(hipStreamSynchronize(h->stream));
    h->timer.resolve();
}

void set_lambda(uzl_pgo* h, double lambda)
{
    h->h_lambda.p[0] = lambda;
    UZL_HIP(hipMemcpyAsync(h->D.scal + 3, h->h_lambda.p, sizeof(double), hipMemcpyHostToDevice, h->stream));
}

// allocate everything that depends on (n, e) only
void alloc_problem(uzl_pgo* h)
{
    const size_t n = std::max(h->n, 1), e = std::max(h->e, 1);
    h->pose_a.reserve(n * 8); h->pose_b.reserve(n * 8);
    h->d_zinv.reserve(e * 7); h->d_info.reserve(e * 36); h->d_robust.reserve(e);
    h->d_ei.reserve(e); h->d_ej.reserve(e); h->d_slot_i.reserve(e); h->d_slot_j.reserve(e);
    h->d_v2b.reserve(n);
    h->d_part_a.reserve(kMaxPartials); h->d_part_b.reserve(kMaxPartials); h->d_part_c.reserve(kMaxPartials);
    h->d_scal.reserve(8); h->d_flags.reserve(4);
    h->h_scal.reserve(1); h->h_lambda.reserve(1);
    h->cur = h->pose_a.p; h->trial = h->pose_b.p;
}

void upload_edges_common(uzl_pgo* h)
{
    hipStream_t s = h->stream;
    const int e = h->e;
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

// block-CSR structure over the free vertices: one slot per (free endpoint, system edge)
void build_structure(uzl_pgo* h)
{
    const int n = h->n, e = h->e;
    std::vector<int32_t> v2b((size_t)std::max(n, 1)), b2v;
    int nb = 0;
    for (int v = 0; v < n; v++) {
        if (h->fixed_eff[v]) v2b[v] = -1; else { v2b[v] = nb++; b2v.push_back(v); }
    }
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
    std::vector<int32_t> col((size_t)std::max(nslots, 1)), slot_i((size_t)std::max(e, 1)), slot_j((size_t)std::max(e, 1));
    for (int k = 0; k < e; k++) {
        const int a = v2b[h->ij[2 * k]], b = v2b[h->ij[2 * k + 1]];
        slot_i[k] = -1; slot_j[k] = -1;
        if (a >= 0) { const int s = fill[a]++; slot_i[k] = s; col[s] = b; }
        if (b >= 0) { const int s = fill[b]++; slot_j[k] = s; col[s] = a; }
    }
    hipStream_t s = h->stream;
    const size_t nbz = std::max(nb, 1), nsz = std::max(nslots, 1);
    h->d_b2v.reserve(nbz); h->d_row_ptr.reserve(nbz + 1); h->d_col.reserve(nsz);
    h->d_blk.reserve(nsz * 36); h->d_dcon.reserve(nsz * 36); h->d_gcon.reserve(nsz * 6);
    h->d_hdiag.reserve(nbz * 36); h->d_minv.reserve(nbz * 36); h->d_b.reserve(nbz * 6);
    h->d_x.reserve(nbz * 6); h->d_r.reserve(nbz * 6); h->d_z.reserve(nbz * 6); h->d_p.reserve(nbz * 6); h->d_ap.reserve(nbz * 6);
    if (n > 0) UZL_HIP(hipMemcpyAsync(h->d_v2b.p, v2b.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    if (nb > 0) UZL_HIP(hipMemcpyAsync(h->d_b2v.p, b2v.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, s));
    UZL_HIP(hipMemcpyAsync(h->d_row_ptr.p, row_ptr.data(), sizeof(int32_t) * (nb + 1), hipMemcpyHostToDevice, s));
    if (nslots > 0) UZL_HIP(hipMemcpyAsync(h->d_col.p, col.data(), sizeof(int32_t) * nslots, hipMemcpyHostToDevice, s));
    if (e > 0) {
        UZL_HIP(hipMemcpyAsync(h->d_slot_i.p, slot_i.data(), sizeof(int32_t) * e, hipMemcpyHostToDevice, s));
        UZL_HIP(hipMemcpyAsync(h->d_slot_j.p, slot_j.data(), sizeof(int32_t) * e, hipMemcpyHostToDevice, s));
    }
    // blocks of slots whose neighbour is fixed are never written: keep them defined
    if (nslots > 0) UZL_HIP(hipMemsetAsync(h->d_blk.p, 0, sizeof(double) * 36 * (size_t)nslots, s));
    UZL_HIP(hipStreamSynchronize(s));       // host vectors go out of scope
    PgoDev& D = h->D;
    D.n = n; D.nb = nb; D.e = e; D.nslots = nslots;
    D.pose = h->cur; D.pose_trial = h->trial;
    D.v2b = h->d_v2b.p; D.b2v = h->d_b2v.p; D.ei = h->d_ei.p; D.ej = h->d_ej.p;
    D.zinv = h->d_zinv.p; D.info = h->d_info.p; D.robust = h->d_robust.p;
    D.slot_i = h->d_slot_i.p; D.slot_j = h->d_slot_j.p; D.row_ptr = h->d_row_ptr.p; D.col = h->d_col.p;
    D.blk = h->d_blk.p; D.dcon = h->d_dcon.p; D.gcon = h->d_gcon.p; D.hdiag = h->d_hdiag.p; D.minv = h->d_minv.p;
    D.b = h->d_b.p; D.x = h->d_x.p; D.r = h->d_r.p; D.z = h->d_z.p; D.p = h->d_p.p; D.ap = h->d_ap.p;
    D.part_a = h->d_part_a.p; D.part_b = h->d_part_b.p; D.part_c = h->d_part_c.p;
    D.scal = h->d_scal.p; D.flags = h->d_flags.p;
    h->structure_ready = true;
}

// one (H + lambda I) dx = b solve; returns PCG iterations used, sets *converged
int pcg_solve(uzl_pgo* h, bool* converged)
{
    hipStream_t s = h->stream;
    const PgoDev& D = h->D;
    const double tol2 = h->cfg.pcg_tol * h->cfg.pcg_tol;
    const int max_it = h->cfg.pcg_max_iter > 0 ? h->cfg.pcg_max_iter : 6 * std::max(h->nb, 1);
    { Timed t(h, "precond"); k_precond(D, s); }
    int gb;
    { Timed t(h, "pcg_init"); gb = k_pcg_init(D, s); }
    { Timed t(h, "pcg_p"); k_pcg_p(D, gb, 1, tol2, s); }
    int launched = 0;
    // first chunk sized from the previous solve, then fixed chunks; the kernels no-op once `done` is set
    int chunk = h->prev_pcg_iters > 0 ? std::max(8, (h->prev_pcg_iters * 9) / 10) : 32;
    while (true) {
        chunk = std::min(chunk, max_it - launched);
        for (int i = 0; i < chunk; i++) {
            int ga, gu;
            { Timed t(h, "pcg_spmv"); ga = k_pcg_spmv(D, s); }
            { Timed t(h, "pcg_update"); gu = k_pcg_update(D, ga, s); }
            { Timed t(h, "pcg_p"); k_pcg_p(D, gu, 0, tol2, s); }
        }
        launched += chunk;
        fetch_scal(h);
        if (h->h_scal.p->flags[0] || launched >= max_it) break;
        chunk = 16;
    }
    UZL_HIP(hipGetLastError());
    *converged = h->h_scal.p->flags[0] != 0 && h->h_scal.p->flags[2] == 0;
    const int iters = h->h_scal.p->flags[1];
    h->prev_pcg_iters = iters;
    return iters;
}

int do_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st)
{
    if (!h->have_graph) return fail(h, UZL_ERR_STATE, "optimize before add_graph/set_graph");
    if (h->world > 1) return fail(h, UZL_ERR_BAD_ARG, "sharded solve (world_size > 1) is not available in this build");
    UZL_HIP(hipSetDevice(h->cfg.device));
    const auto t0 = std::chrono::steady_clock::now();
    if (iterations <= 0) iterations = h->cfg.iterations;
    uzl_pgo_stats S;
    memset(&S, 0, sizeof(S));
    // optimizeImpl: initializeOptimization (:139), setFixedNodes (:144-146)
    h->fixed_eff = h->fixed_in;
    h->n_gauge = gauge_fix(h);
    build_structure(h);
    S.n_vertices = h->n; S.n_edges = h->e; S.n_gauge_fixed = h->n_gauge;
    hipStream_t s = h->stream;
    PgoDev& D = h->D;
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
    for (int it = 0; it < iterations; it++) {
        int gl, ga;
        D.pose = h->cur; D.pose_trial = h->trial;
        { Timed t(h, "linearize"); gl = k_linearize(D, h->cur, delta, s); }     // computeActiveErrors + buildSystem
        { Timed t(h, "assemble"); ga = k_assemble(D, s); }
        { Timed t(h, "finalize"); k_finalize(D, gl, 0, ga, 2, s); }
        fetch_scal(h);
        current_chi = h->h_scal.p->scal[4];
        if (it == 0) {
            S.chi2_initial = current_chi;
            lambda = 1e-5 * h->h_scal.p->scal[6];                                 // computeLambdaInit: tau * max diag
            ni = 2.;
        }
        double rho = 0.;
        int qmax = 0;
        do {
            set_lambda(h, lambda);                                                // setLambda
            bool conv = false;
            S.pcg_iterations += pcg_solve(h, &conv);                              // _solver->solve()
            if (!conv) { S.pcg_not_converged++; rc = UZL_ERR_NOT_CONVERGED; }
            S.lm_trials++;
            int go, gc;
            { Timed t(h, "oplus"); go = k_oplus(D, h->cur, h->trial, s); }        // push + update
            { Timed t(h, "chi2"); gc = k_chi2(D, h->trial, delta, s); }           // computeActiveErrors
            { Timed t(h, "finalize"); k_finalize(D, gc, go, 0, 1, s); }
            fetch_scal(h);
            const double temp_chi = h->h_scal.p->scal[4];
            const double scale = h->h_scal.p->scal[5] + 1e-3;                     // computeScale + 1e-3
            rho = (current_chi - temp_chi) / scale;
            if (rho > 0 && std::isfinite(temp_chi)) {                             // good step
                double alpha = 1. - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2.;
                current_chi = temp_chi;
                std::swap(h->cur, h->trial);                                      // discardTop
            } else {
                lambda *= ni;
                ni *= 2.;                                                          // pop: h->cur untouched
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        S.iterations_done = it + 1;
        if (qmax == 10 || rho == 0) { S.terminated_early = 1; break; }           // Terminate
    }
    S.chi2_final = current_chi;
    S.lambda_final = lambda;
    UZL_HIP(hipStreamSynchronize(s));
    S.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (st) *st = S;
    if (rc != UZL_OK) h->last_error = "PCG hit pcg_max_iter in at least one LM trial";
    return rc;
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

void uzl_pgo_cfg_default(uzl_pgo_cfg* cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->iterations = 20;               // cfg/GraphOptimizer.cfg:10
    cfg->use_odometry_parameters = 0;   // :11
    cfg->optimize_xy_only = 0;          // :12
    cfg->device = 0;
    cfg->pcg_tol = 1e-8;
    cfg->pcg_max_iter = 0;              // 0 = 6 * free vertices (system dimension)
    cfg->huber_delta = 1.0;             // g2o_optimizer.cpp:293
    cfg->verbose = 0;
}

int uzl_pgo_create(const uzl_pgo_cfg* cfg, uzl_pgo** out)
{
    if (!out) return UZL_ERR_BAD_ARG;
    *out = nullptr;
    uzl_pgo_cfg c;
    if (cfg) c = *cfg; else uzl_pgo_cfg_default(&c);
    if (c.use_odometry_parameters) return UZL_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UZL_ERR_NO_DEVICE;
    if (c.device < 0 || c.device >= ndev) return UZL_ERR_NO_DEVICE;
    uzl_pgo* h = new (std::nothrow) uzl_pgo();
    if (!h) return UZL_ERR_OOM;
    h->cfg = c;
    memset(&h->D, 0, sizeof(h->D));
    if (hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return UZL_ERR_HIP;
    }
    *out = h;
    return UZL_OK;
}

void uzl_pgo_destroy(uzl_pgo* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

int uzl_pgo_set_config(uzl_pgo* h, const uzl_pgo_cfg* cfg)
{
    if (!h || !cfg) return UZL_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (cfg->device != h->cfg.device) return fail(h, UZL_ERR_BAD_ARG, "device cannot change after create");
    if (cfg->use_odometry_parameters) return fail(h, UZL_ERR_BAD_ARG, "use_odometry_parameters is not supported");
    if (cfg->iterations < 1 || cfg->pcg_tol <= 0. || cfg->huber_delta <= 0.) return fail(h, UZL_ERR_BAD_ARG, "bad config value");
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
    h->have_graph = false; h->structure_ready = false;         // clear() (:57)
    h->n = n_nodes; h->e_in = n_edges;
    h->fixed_in.assign((size_t)n_nodes, 0);
    for (int v = 0; v < n_nodes; v++) h->fixed_in[v] = nodes[v].fixed ? 1 : 0;
    // skip rules; odometry edges are added while iterating (:78-79), filtered feature edges after (:100-103)
    h->ij.clear(); h->src.clear(); h->robust.clear();
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < n_edges; k++) {
            const uzl_edge& ed = edges[k];
            if (ed.from < 0 || ed.to < 0 || ed.from >= n_nodes || ed.to >= n_nodes) continue;     // :77, :194-201, :263-268
            if (ed.from == ed.to) continue;                                                        // degenerate self edge
            const bool odom = ed.type == UZL_EDGE_TYPE_2D_WHEEL_ODOMETRY;
            if ((pass == 0) != odom) continue;
            if (odom) {
                if (nodes[ed.from].fixed && !nodes[ed.to].fixed) continue;                         // :203-206
            } else {
                if (!ed.valid) continue;                                                           // not in validEdges() (:98)
                if (nodes[ed.from].fixed && nodes[ed.to].fixed) continue;                          // :270-274
            }
            h->ij.push_back(ed.from); h->ij.push_back(ed.to);
            h->src.push_back(k);
            h->robust.push_back(odom ? 0 : 1);                                                     // Huber on feature edges (:292-294)
        }
    }
    h->e = (int32_t)h->src.size();
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
    k_prepare_edges(h->d_edges.p, h->d_src.p, h->e, h->d_stage.p, n_sensors, h->cfg.optimize_xy_only,
                    h->d_zinv.p, h->d_info.p, s);
    UZL_HIP(hipGetLastError());
    UZL_HIP(hipStreamSynchronize(s));                          // inputs are borrowed for the duration of the call only
    upload_edges_common(h);
    h->have_graph = true;
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
    h->have_graph = false; h->structure_ready = false;
    h->n = n; h->e_in = e; h->e = e;
    h->fixed_in.assign(fixed, fixed + n);
    for (auto& f : h->fixed_in) f = f ? 1 : 0;
    h->ij.assign(ij, ij + 2 * (size_t)e);
    h->robust.assign(robust, robust + e);
    h->src.resize((size_t)e);
    std::iota(h->src.begin(), h->src.end(), 0);
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
    UZL_HIP(hipStreamSynchronize(s));
    upload_edges_common(h);
    h->have_graph = true;
    return UZL_OK;
    UZL_GUARD_END(h)
}

int uzl_pgo_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* stats)
{
    UZL_GUARD_BEGIN(h)
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
        if (!h->structure_ready) {      // store without optimize: D only needs the edge arrays
            h->fixed_eff = h->fixed_in;
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
    h->rank = rank; h->world = world_size; h->allreduce = allreduce; h->allreduce_user = user;
    return UZL_OK;
}

}  // extern "C"
