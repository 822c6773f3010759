// pgo_handle.hpp — the pose-graph handle (struct uzl_pgo), the kernel launchers and the host-side pieces shared by the translation
// units behind uzl_pgo_*: uzl_pgo.hip (C ABI, structure, the host-driven Levenberg-Marquardt loop), uzl_pgo_lm.hip (the device-resident
// loop: captured passes over slot twins).  Private to csrc/: the product surface is include/uzl_mi355x.h.
#pragma once
#include "uzl_common.hpp"
#include "pgo_types.hpp"
#include "pgo_schur.hpp"
#include "pgo_lm.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <new>
#include <numeric>

namespace uzl {
void k_prepare_nodes(const uzl_node* nodes, int n, int xy, double* pose, hipStream_t s);
void k_prepare_flat_nodes(const double* poses12, int n, double* pose, hipStream_t s);
void k_prepare_edges(const uzl_edge* edges, const int32_t* src, int e, const double* sensors, int ns, int xy, int odom_params,
                     double* zinv, double* info, hipStream_t s);
void k_prepare_flat_edges(const double* meas12, const double* info36, int e, double* zinv, double* info, hipStream_t s);
int k_chi2(const PgoDev& D, const double* pose, double delta, hipStream_t s);
int k_chi2_trial(const PgoDev& D, const double* pose, double delta, hipStream_t s);
hipError_t k_hessian(const PgoDev& D, const double* pose, double delta, int* g_edges, int* g_rows, hipStream_t s, bool with_chi2 = true, hipEvent_t ev_a = nullptr,
                     hipEvent_t ev_b = nullptr);
void k_finalize(const PgoDev& D, int na, int nb_, int nc, int what, hipStream_t s);
int k_diagmax(const PgoDev& D, hipStream_t s);
void k_precond(const PgoDev& D, hipStream_t s);
void k_publish(const PgoDev& D, PgoHostScal* out_dev, uint32_t seq, hipStream_t s);
void k_set_scalar(double* dst, double v, hipStream_t s);
void k_set_scalar2(double* dst_a, double va, double* dst_b, double vb, hipStream_t s);
void k_residual_guard(const PgoDev& D, hipStream_t s);
void k_pcg_progress(const PgoDev& D, hipStream_t s);
void k_set_trial(double* scal, double lambda, double tol_f2, double eps_t, double eps_r, hipStream_t s);
int k_pcg_init(const PgoDev& D, double* p0, double* p1, hipStream_t s);
int k_pcg_spmv(const PgoDev& D, const double* p_old, double* p_new, int n_part, double tol2, hipStream_t s);
int k_pcg_update(const PgoDev& D, const double* p, int n_part, hipStream_t s);
int g_pcg_spmv(int nb);
int g_pcg_update(int nb);
constexpr int kGeoAllMaxHost = 1024;                  // (= kGeoAllMax of pgo_ml_kernels.hip)
void k_ml_geometry(const PgoDev& D, const MlDev* ml, const double* pose, int l, int n_l, hipStream_t s);
void k_ml_galerkin(const PgoDev& D, const MlDev* ml, int f, int n_chunks, hipStream_t s);
void k_ml_sibling(const PgoDev& D, const MlDev* ml, int total_aggs, hipStream_t s);
void k_ml_dense_level(const MlDev* ml, int l, int n_l, hipStream_t s);
void k_ml_mult_level(const PgoDev& D, const MlDev* ml, int lev, int n1, int n2, hipStream_t s);
void k_ml_cmat32(const MlHot& hot, int n6, hipStream_t s);
void k_ml_ns_step(const PgoDev& D, const MlDev* ml, int lev, int n1, const double* X, double* T, double* Xn, hipStream_t s,
                  hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr, float* c32 = nullptr, int c32_stride = 0);
int g_ml_rows(int nb, int agg);
int g_ml_spmv(int nb, int agg);
size_t ml_cg_lds_bytes(const int* n, int levels, int agg);
bool ml_fits_lds(const int* n_per_level, int levels, int agg);
// The ONE statement of the PCG kernels' LDS budget and of what ml_cg stages when the dense level-2 operator is present (the gather-level
// vector and nothing else): build_ml's admission test, k_ml_cg, ml_cg_variant and ml_fits_lds all read these (tests/test_ml_admission.py
// holds the boundaries through uzl_debug_ml_admission)
// (kMlLdsLimit, ml_comp4_lds: pgo_types.hpp)
bool ml_comp4_fits(int nb, int n2);          // LDS of the comp4 variant and ml_spmv's partial count
void k_ml_init(const PgoDev& D, const MlHot& ml, int agg, double* p0, double* p1, double* rg, hipStream_t s);
void k_ml_spmv(const PgoDev& D, const MlHot& ml, int agg, const double* p_old, double* p_new, int n_part, double tol2, hipStream_t s,
               hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr);
hipError_t k_ml_cg(const PgoDev& D, const MlHot& ml, int agg, const double* p, const double* rg_old, double* rg_new, int n_part,
                   int init, size_t lds, hipStream_t s, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr);
int k_oplus(const PgoDev& D, const double* pose_in, double* pose_out, hipStream_t s);
int g_edges_for(int e);
void k_slot_records(const double* zinv, const double* info, int e, const int32_t* slot_edge, int nslots, double* srec, hipStream_t s);
int g_oplus_for(int n);
void k_edge_error(const PgoDev& D, const double* pose, double* err, hipStream_t s);
void k_poses_out(const double* pose, int n, double* out12, hipStream_t s);
void k_schur_gather(const PgoDev& D, const SchurDev& S, hipStream_t s);
void k_schur_eliminate(const PgoDev& D, const SchurDev& S, hipStream_t s);
void k_schur_assemble(const PgoDev& D, const PgoDev& R, const SchurDev& S, hipStream_t s);
void k_schur_backsub(const PgoDev& D, const PgoDev& R, const SchurDev& S, hipStream_t s);
// slot twins of the device-resident LM loop (pgo_types.hpp: LmSlot / LmDev / LmShape)
void k_lm_head(const LmSlot* slots, int nslots, int pass_flags, hipStream_t s);
void k_lm_tail(const LmSlot* slots, int nslots, hipStream_t s);
hipError_t kl_linearize(const LmSlot* sl, int nslots, int g_edges, int g_asm, hipStream_t s);
void kl_eval(const LmSlot* sl, int nslots, int g_edges, int g_oplus, hipStream_t s);
void kl_schur_reduce(const LmSlot* sl, int nslots, int max_runs, long max_items, hipStream_t s);
void kl_schur_backsub(const LmSlot* sl, int nslots, int max_grid, hipStream_t s);
void kl_ml_numeric(const LmSlot* sl, const LmShape& sh, int which, hipStream_t s);
void kl_ml_trial(const LmSlot* sl, const LmShape& sh, int which, hipStream_t s);
void ml_cg_variant(const MlHot& ml, int agg, size_t lds_full, int32_t* variant, int32_t* comp_u, uint64_t* lds);
hipError_t kl_ml_init(const LmSlot* sl, const LmSlot* host_slot, const LmShape& sh, hipStream_t s);
hipError_t kl_ml_pcg_its(const LmSlot* sl, const LmSlot* host_slot, const LmShape& sh, int first, int n, hipStream_t s, hipEvent_t* ev = nullptr);
}  // namespace uzl



namespace uzl { struct LmRun; }
using namespace uzl;      // (private header of the uzl_pgo_* translation units; the handle itself is the C ABI's global-namespace type)

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
    // what the skip rules need of every INPUT edge of uzl_pgo_add_graph (kept so that uzl_pgo_append_graph can grow the graph in place)
    struct InEdge { int32_t from, to; uint8_t odom, valid; double w; };
    std::vector<InEdge> in_edges;
    bool in_ready = false;         // in_edges / d_edges describe the current graph (add_graph or append_graph, not set_graph)
    int32_t n_sensors_in = 0;
    std::vector<uint8_t> robust;
    std::vector<double> edge_w;    // trace of each system edge's information matrix: the coupling strength the aggregation order follows
    bool have_graph = false, structure_ready = false;
    int32_t n_gauge = 0;
    // ---- device
    DevBuf<double> pose_a, pose_b, pose_init;
    double* cur = nullptr;
    double* trial = nullptr;
    DevBuf<int32_t> d_v2b, d_b2v, d_ei, d_ej, d_row_ptr, d_col, d_rowhdr, d_src, d_flags;
    DevBuf<int32_t> d_slot_edge, d_rb_ptr;
    DevBuf<double> d_srec;                     // slot records (pgo_kernels.hip: slot_records_kernel): values of the edges in the order of the structure's slots
    DevBuf<int4> d_smeta;
    int ml_ns_now = -1;                        // Newton-Schulz steps of the set-up in progress (-1: ml_ns_steps; host-driven loop)
    bool srec_stale = true;                    // edges' values or the structure changed since d_srec was written
    DevBuf<double> d_zinv, d_info, d_blk, d_hdiag, d_minv, d_b, d_x, d_xs, d_r, d_z, d_p, d_p2, d_ap;
    DevBuf<double> d_part_a, d_part_b, d_part_c, d_scal, d_err, d_out12, d_stage;
    DevBuf<uint8_t> d_robust;
    DevBuf<uzl_node> d_nodes;
    DevBuf<uzl_edge> d_edges;
    PinBuf<PgoHostScal> h_scal;      // pinned + coherent: written by publish_kernel, polled by the host
    PgoHostScal* d_pub = nullptr;    // its device-side address
    uint32_t pub_seq = 0;
    PinBuf<double> h_lambda;
    PgoDev D;
    // The system the PCG (and its preconditioner) sees: D itself, or - when chain interiors are Schur-eliminated (pgo_schur.hpp) - the
    // reduced system over the separator vertices.  scal / flags / part_b / part_c are shared with D.
    PgoDev Dp;
    double* pbuf[2] = {nullptr, nullptr};      // PCG direction, ping-pong (of the Dp system)
    struct Reduced {
        bool on = false, strong = false, strong_blocks = false;      // strong: numbered by strong aggregates, with empty rows (SchurPlan); _blocks: in blocks of 4 groups
        int32_t n_int = 0, n_runs = 0, longest_run = 0, n_sep = 0;
        DevBuf<int32_t> run_ptr, run_rows, slotP, slotN, endL, endR, sep_rows, rsrc, inc_ptr, inc, row_ptr, col, rowhdr, b2v;
        DevBuf<double> elim, runout, runblk, blk, hdiag, minv, x, xs, r, z, p, p2, ap;
        SchurDev S;
    } red;
    int prev_pcg_iters = 0;
    double num_its[2] = {-1., -1.};            // PCG iterations per LM trial of the last solve with the reduced system in row order / by strong aggregates
    int num_last = -1;                         // numbering of that solve (-1: none yet); see build_structure
    // multilevel preconditioner
    int ml_levels = 0;
    std::vector<int32_t> ml_n, ml_nslots, ml_chunks;       // per level: entities, off-diagonal blocks, work chunks of ml_galerkin_kernel
    int ml_inner_aggs = 0;
    bool ml_trial_setup = false;     // the preconditioner's per-trial part (sibling inverses, top, dense levels) is due
    bool ml_comp = false;
    int ml_cl = 0;                   // level of the dense operator (1: small graphs, 2: AGG = 4), 0 = none
    int ml_ns_steps = 0;             // Newton-Schulz refinements of the dense level-1 operator per rebuild
    // Two complete copies of the preconditioner's numeric state (arena, device descriptor, kernel-argument block, PCG graph): the
    // solver applies copy `ml_ix` while a rebuild for the next LM iteration runs on `stream2` into the other one.
    struct MlBuf {
        MlDev* dml = nullptr;                  // device copy of the descriptor
        MlHot hot;                             // hot subset, by-value kernel argument
        double* rg[2] = {nullptr, nullptr};    // double-buffered gather-level residual
        double* y1 = nullptr; double* nsT = nullptr; double* nsX = nullptr;
        double* l1_span_ptr = nullptr;         // level-1 Galerkin arrays (blk | G | M), all-reduced once per linearisation
        hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;        // 2 x kGraphPairs PCG iterations
        hipGraph_t graph_s = nullptr; hipGraphExec_t graph_exec_s = nullptr;    // 2 x kShortPairs: the solves of a converged LM iteration end after 2 - 4
        double lambda_setup = 0.;              // lambda of the last trial set-up of this copy
    } mlb[2];
    double* ml_dense_ptr[2][kMlMaxLevels + 1] = {};   // host copy of MlDev::Ydense per hierarchy copy
    int ml_ix = 0;
    bool ml_pending = false;                   // a rebuild into copy ml_ix ^ 1 is in flight on stream2
    hipStream_t stream2 = nullptr;
    bool streams_borrowed = false;             // stream / stream2 are a batch's (uzl_pgo_batch_create): not this handle's to destroy
    hipEvent_t ev_lin = nullptr, ev_setup = nullptr;
    DevBuf<double> d_scal2;                    // lambda slot (scal[3]) for the kernels of an asynchronous rebuild
    double lambda_now = 0.;          // lambda of the current trial
    bool ml_mult = false;            // level 1 of the composite operator is multiplicative (pgo_ml_kernels.hip)            // small graphs: hierarchy above level 1 folded into a dense operator (pgo_ml_kernels.hip)
    std::vector<int32_t> ml_fan;
    int ml_agg = 4;                            // level-1 aggregates per PCG workgroup (1: small graphs, 4: large)
    size_t ml_lds = 0;
    DevBuf<uint8_t> ml_arena;
    size_t ml_copy_stride = 0;                 // bytes between the two hierarchy copies inside ml_arena
    DevBuf<MlDev> d_ml;
    bool no_graph = false;          // UZL_NO_GRAPH=1: eager launches (rocprofv3 --kernel-trace crashes on hipGraphLaunch here)
    // shard (BASELINE config 4)
    int32_t rank = 0, world = 1;
    uzl_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;       // ncclComm_t owned by the handle (uzl_pgo_set_shard_rccl); the exchange then needs no callback
    bool sharded = false;            // an exchange (callback or communicator) is set and the multilevel path is active for this structure
    DevBuf<double> d_red;
    int64_t iter_span = 0;           // doubles all-reduced per PCG iteration: [A p | restricted A p | p.Ap partials]
    int64_t l1_span = 0;
    // per-optimize accounting (uzl_pgo_stats)
    double structure_ms = 0., exchange_ms = 0.;
    int32_t exchange_calls = 0;
    bool structure_reused = false;
    double last_residual_ratio = 0.;
    int32_t guard_trips = 0;
    uint64_t structure_gen = 0;      // bumped by build_structure: batches rebuild their slots when it moves
    bool mult_banned = false;        // the multiplicative operator broke down on a graph of this handle: later structures start additive
    KernelTimer timer;
    uzl::LmRun* lm = nullptr;        // the device-resident LM loop's slot table and captured passes (uzl_pgo_lm.hip)
    bool last_structure_reused = false;
    std::chrono::steady_clock::time_point t_start;      // start of the running uzl_pgo_optimize (solve_ms)
};

namespace uzl {
// ---- when a linear solve stops (uzl_pgo.hip "When a linear solve stops"; cfg.pcg_stop = 1: the plain relative test - no floor factor,
//      and a step accuracy of 0 switches the look off: progress_unit, pgo_device.hpp)
inline double pgo_tol_f2(const uzl_pgo_cfg& c) { return c.pcg_stop == 1 ? 1. : kTolFloor2; }
inline double pgo_eps_t(const uzl_pgo_cfg& c) { return c.pcg_stop == 1 ? 0. : kStepT * c.pcg_tol; }
inline double pgo_eps_r(const uzl_pgo_cfg& c) { return c.pcg_stop == 1 ? 0. : kStepR * c.pcg_tol; }
// ---- shared host-side pieces (uzl_pgo.hip)
int pgo_fail(uzl_pgo* h, int code, const char* msg);
int32_t gauge_fix(uzl_pgo* h);                        // G2: setFixedNodes (g2o_optimizer.cpp:301-349)
void build_structure(uzl_pgo* h);                     // block-CSR, Schur plan, hierarchy; bumps structure_gen
void destroy_pcg_graph(uzl_pgo* h);
bool ml_async_level(const uzl_pgo* h);
double ml_rate_drop(const uzl_pgo* h);
int ml_ns_steps_at(int structure_steps, int lm_iteration);
void ml_setup_numeric(uzl_pgo* h, int bi, hipStream_t s, const PgoDev& D, bool timed);
void ml_setup_trial(uzl_pgo* h, int bi, hipStream_t s, const PgoDev& D, bool timed);
void prepare_optimize(uzl_pgo* h);                    // optimizeImpl's front part: gauge + structure (cached), t_start
void own_streams(uzl_pgo* h, bool drain_borrowed);               // a batch's handle takes streams of its own (uzl_pgo.hip)
int do_optimize_host(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st);      // the host-driven loop (sharded / block-Jacobi / profiled solves, anomaly fallback)
int do_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st);           // picks the loop
// ---- the device-resident loop (uzl_pgo_lm.hip)
struct LmRun;                                          // captured passes + slot table of one handle (or one batch)
void lm_run_destroy(LmRun* r);
bool lm_eligible(const uzl_pgo* h);
int do_optimize_lm(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st);
bool lm_batch_eligible(const std::vector<uzl_pgo*>& hs);
int batch_optimize_lm(LmRun*& R, const std::vector<uzl_pgo*>& hs, int resident, hipStream_t s, hipStream_t s2, int32_t iterations, bool eager, bool verbose, KernelTimer* timer,
                      uzl_pgo_stats* stats, int* rc_all);
constexpr int kSchurStrongOneMax = 256;                // strong aggregates: up to this many groups as ONE level (level-1 path), beyond in blocks of 4 (pgo_schur.hpp)
constexpr int kSchurStrongMin = 32;                   // separators from which on the reduced system is numbered by strong aggregates
extern const int kUpperNs;                            // Newton-Schulz steps of the dense levels above the composite level
extern const bool kAlwaysRefresh;                     // A/B switches (diagnostic build)
extern const double kRefreshRel, kRefreshRelSync, kLambdaRetake;
extern const int kGraphPairs;                         // one long PCG replay = 2 x kGraphPairs iterations
constexpr int kShortPairs = 2;                        // ... a short one 2 x kShortPairs
}  // namespace uzl
