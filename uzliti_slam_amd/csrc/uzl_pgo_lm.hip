// uzl_pgo_lm.hip — the device-resident Levenberg-Marquardt loop behind uzl_pgo_optimize.
//
// G2oOptimizer::optimizeImpl (graph_optimization/src/g2o_optimizer.cpp:137-149) hands the graph to optimizer_.optimize(iterations),
// i.e. g2o's OptimizationAlgorithmLevenberg::solve [EXT]: linearise, then trials (lambda, solve, evaluate, accept / reject) until a step
// is accepted.  uzl_pgo.hip's do_optimize_host drives that loop from the host: ~12 eager launches and two to three host round trips
// per trial - at BASELINE config 2 more than half of a solve.  Here the loop's state lives on the device (LmDev), its decisions are
// taken by two one-workgroup kernels (pgo_lm_kernels.hip), every other kernel is a slot twin that predicates itself on that state, and
// one trial is a PASS: a fixed sequence of captured segments
//     head  = linearise, assemble | lm_head (chi2, lambda_0, adoption, refresh decision, setLambda) | [Schur reduction]
//     setup = numeric + trial part of the hierarchy copy in use        (first iteration; synchronous rebuilds; lambda grown 32x)
//     reb   = rebuild of the OTHER copy on the second stream           (lazy refresh: adopted by the next iteration)
//     init  = x = 0, r = b, first application of the preconditioner
//     pcg   = 2 x kGraphPairs / 2 x kShortPairs iterations             (no-ops once the solve's `done` flag is set)
//     tail  = residual guard, [back-substitution], retraction, chi2 | lm_tail (rho, accept / reject, next phase, snapshot for the host)
// The host enqueues a pass, looks at the snapshot ONCE, and chooses the next pass's segments and PCG count from it.  Its choices are
// predictions only: lm_head stalls a graph whose pass lacks a segment it needs (kLmNeedSetup) and the next pass brings it.  Same
// kernels' bodies, same order of operations, same scalar arithmetic (pgo_lm.hpp) as the host-driven loop: tests hold the two to
// array_equal poses.  The host-driven loop remains for sharded and profiled solves and block-Jacobi, and takes over - from the start
// poses - when a solve meets an anomaly (PCG breakdown, not converged, residual guard).
#include "pgo_handle.hpp"

namespace uzl {

struct LmRun {
    int device = 0;
    DevBuf<LmSlot> d_slots;
    DevBuf<LmDev> d_lm;
    PinBuf<LmHost> h_pub;                    // mapped + coherent: written by lm_tail_kernel, polled by the host
    LmHost* d_pub = nullptr;
    PinBuf<LmDev> h_init;                    // staging of the initial state
    std::vector<LmSlot> slots;               // host copies of the slots (by-value launches of a one-graph pass)
    LmShape shape{};
    DevBuf<double> d_start;                  // poses at the start of the solve (anomaly fallback)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool join_pending = false;               // a rebuild is in flight on the second stream
    uint32_t tails = 0;                      // lm_tail launches enqueued = sequence word expected next
    uint64_t gen = ~0ull;                    // structure generation the slot and the captured segments belong to
    int first_solve_its = 0;                 // PCG iterations of the first solve of the last optimize (sizes the first pass of the next)
    struct Seg { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; };
    Seg head[4], setup, reb, init, pcg_long, pcg_short, tail;
    void drop(Seg& q) { if (q.x) { (void)hipGraphExecDestroy(q.x); q.x = nullptr; } if (q.g) { (void)hipGraphDestroy(q.g); q.g = nullptr; } }
    void drop_all() { for (auto& q : head) drop(q); drop(setup); drop(reb); drop(init); drop(pcg_long); drop(pcg_short); drop(tail); }
};

void lm_run_destroy(LmRun* r)
{
    if (!r) return;
    r->drop_all();
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    if (r->ev_join) (void)hipEventDestroy(r->ev_join);
    delete r;
}

namespace {

static const bool lm_host_forced = diag_flag("UZL_LM_HOST");              // A/B switch (diagnostic build): the host-driven loop everywhere
static const bool lm_slot_ptr = diag_flag("UZL_LM_SLOT_PTR");             // A/B switch: PCG kernels read the slot through its pointer also for one graph

// ---- the segments of a pass, as launch sequences (captured once per structure, or launched as they are when graphs are off)
void enq_head(LmRun* R, int pass_flags, hipStream_t s)
{
    const LmShape& sh = R->shape;
    kl_linearize(R->d_slots.p, sh.nslots, sh.g_edges, sh.g_asm, s);
    k_lm_head(R->d_slots.p, sh.nslots, pass_flags, s);
    if (sh.red) kl_schur_reduce(R->d_slots.p, sh.nslots, sh.schur_runs, (long)sh.schur_items, s);
}
void enq_setup(LmRun* R, int which, hipStream_t s)
{
    kl_ml_numeric(R->d_slots.p, R->shape, which, s);
    kl_ml_trial(R->d_slots.p, R->shape, which, s);
}
const LmSlot* by_value_slot(LmRun* R) { return (R->shape.nslots == 1 && !lm_slot_ptr) ? R->slots.data() : nullptr; }
void enq_init(LmRun* R, hipStream_t s) { UZL_HIP(kl_ml_init(R->d_slots.p, by_value_slot(R), R->shape, s)); }
void enq_pcg(LmRun* R, int pairs, hipStream_t s) { UZL_HIP(kl_ml_pcg_pairs(R->d_slots.p, by_value_slot(R), R->shape, pairs, s)); }
void enq_tail(LmRun* R, hipStream_t s)
{
    const LmShape& sh = R->shape;
    if (sh.red) kl_schur_backsub(R->d_slots.p, sh.nslots, sh.schur_backsub_grid, s);
    kl_eval(R->d_slots.p, sh.nslots, sh.g_edges, sh.g_oplus, s);
    k_lm_tail(R->d_slots.p, sh.nslots, s);
}

template <class F>
void run_seg(LmRun::Seg& q, bool eager, hipStream_t s, F&& enqueue)
{
    if (eager) { enqueue(s); return; }
    if (!q.x) {
        UZL_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue(s);
        UZL_HIP(hipStreamEndCapture(s, &q.g));
        UZL_HIP(hipGraphInstantiate(&q.x, q.g, nullptr, nullptr, 0));
    }
    UZL_HIP(hipGraphLaunch(q.x, s));
}

// the slot of a handle's current structure
LmSlot make_slot(const uzl_pgo* h, LmDev* d_lm, LmHost* d_pub)
{
    LmSlot S;
    memset(&S, 0, sizeof(S));
    S.D = h->D; S.Dp = h->Dp;
    S.D.flags = d_lm->flags; S.Dp.flags = d_lm->flags;          // the PCG kernels' done / iterations / breakdown words live in the LM state
    S.D.pose = nullptr; S.D.pose_trial = nullptr; S.Dp.pose = nullptr; S.Dp.pose_trial = nullptr;
    S.SD = h->red.S; S.red = h->red.on ? 1 : 0;
    S.lm = d_lm; S.pub = d_pub;
    for (int c = 0; c < 2; c++) {
        S.hot[c] = h->mlb[c].hot; S.dml[c] = h->mlb[c].dml;
        S.rg[c][0] = h->mlb[c].rg[0]; S.rg[c][1] = h->mlb[c].rg[1];
        for (int l = 0; l <= kMlMaxLevels; l++) S.dense[c][l] = h->ml_dense_ptr[c][l];
        S.nsT[c] = h->mlb[c].nsT; S.nsX[c] = h->mlb[c].nsX;
    }
    S.pbuf[0] = h->pbuf[0]; S.pbuf[1] = h->pbuf[1];
    S.pose[0] = h->pose_a.p; S.pose[1] = h->pose_b.p;
    S.g_edges = g_edges_for(h->e); S.g_asm = g_asm_for(h->nb); S.g_oplus = g_oplus_for(h->n);
    S.g_rows = g_ml_rows(h->Dp.nb, h->ml_agg); S.g_spmv = g_ml_spmv(h->Dp.nb, h->ml_agg);
    S.copy_stride = (int64_t)h->ml_copy_stride;
    return S;
}

LmShape make_shape(const uzl_pgo* h, const LmSlot& S)
{
    LmShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.nslots = 1; sh.batch_geometry = 0;
    sh.levels = h->ml_levels; sh.cl = h->ml_comp ? h->ml_cl : 0; sh.agg = h->ml_agg;
    sh.mult = h->ml_mult ? 1 : 0; sh.ns_steps = h->ml_ns_steps; sh.upper_ns = kUpperNs;
    for (int l = 0; l <= h->ml_levels; l++) { sh.n_lv[l] = h->ml_n[l]; sh.work_t[l] = h->ml_nslots[l] + h->ml_n[l]; }
    sh.inner_aggs = h->ml_inner_aggs;
    ml_cg_variant(S.hot[0], h->ml_agg, h->ml_lds, &sh.cg_variant, &sh.comp_u, &sh.cg_lds);
    sh.g_edges = S.g_edges; sh.g_asm = S.g_asm; sh.g_oplus = S.g_oplus; sh.g_rows = S.g_rows; sh.g_spmv = S.g_spmv;
    sh.red = S.red;
    if (S.red) {
        sh.schur_runs = S.SD.n_runs;
        sh.schur_items = (int64_t)(S.SD.nslots_r + S.SD.nbr) * 36;
        sh.schur_backsub_grid = S.SD.n_runs + (S.SD.nbr * 6 + 63) / 64;
    }
    return sh;
}

// Waits until lm_tail launch number `seq` (or a later one) of slot `sl` has published and copies the snapshot; returns its number.
// lm_tail writes seq_begin, the fields, then seq (release): a copy is whole when both words agree around it.
uint32_t wait_pub(uzl_pgo* h, LmRun* R, int sl, uint32_t seq, LmHost* out)
{
    const auto t0 = std::chrono::steady_clock::now();
    volatile LmHost* pub = R->h_pub.p + sl;
    for (int spin = 0;; spin++) {
        const uint32_t s1 = __atomic_load_n(&R->h_pub.p[sl].seq, __ATOMIC_ACQUIRE);
        if ((int32_t)(s1 - seq) >= 0) {
            memcpy(out, const_cast<const LmHost*>(R->h_pub.p + sl), sizeof(LmHost));
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (pub->seq_begin == s1 && out->seq == s1) return s1;           // (else a later tail is writing: its seq will land)
        }
        if ((spin & 1023) == 1023 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 200.0) {
            UZL_HIP(hipStreamSynchronize(h->stream));
            const uint32_t s2 = __atomic_load_n(&R->h_pub.p[sl].seq, __ATOMIC_ACQUIRE);
            if ((int32_t)(s2 - seq) < 0) throw HipError{hipErrorUnknown, "lm_tail_kernel did not publish", __FILE__, __LINE__};
            memcpy(out, const_cast<const LmHost*>(R->h_pub.p + sl), sizeof(LmHost));
            return s2;
        }
    }
}

}  // namespace

// which solves take the device-resident loop
bool lm_eligible(const uzl_pgo* h)
{
    return !lm_host_forced && h->cfg.lm_loop != 1 && h->ml_levels > 0 && !h->sharded && h->nb > 0 && h->e > 0 && h->Dp.nb > 0 && !h->timer.on &&
           h->stream2 != nullptr;
}

int do_optimize_lm(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* st)
{
    hipStream_t s = h->stream;
    if (!h->lm) {
        h->lm = new LmRun();
        h->lm->device = h->cfg.device;
        UZL_HIP(hipEventCreateWithFlags(&h->lm->ev_fork, hipEventDisableTiming));
        UZL_HIP(hipEventCreateWithFlags(&h->lm->ev_join, hipEventDisableTiming));
    }
    LmRun* R = h->lm;
    const bool eager = h->no_graph;
    // ---- slot + shape of this structure (captured segments belong to it)
    if (R->gen != h->structure_gen) {
        const auto ts = std::chrono::steady_clock::now();
        UZL_HIP(hipStreamSynchronize(s));
        R->drop_all();
        R->d_slots.reserve(1); R->d_lm.reserve(1);
        R->h_pub.reserve(1, hipHostMallocMapped | hipHostMallocCoherent);
        R->h_init.reserve(1);
        UZL_HIP(hipHostGetDevicePointer((void**)&R->d_pub, R->h_pub.p, 0));
        R->slots.assign(1, make_slot(h, R->d_lm.p, R->d_pub));
        R->shape = make_shape(h, R->slots[0]);
        UZL_HIP(hipMemcpyAsync(R->d_slots.p, R->slots.data(), sizeof(LmSlot), hipMemcpyHostToDevice, s));
        UZL_HIP(hipStreamSynchronize(s));
        R->gen = h->structure_gen;
        h->structure_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
    }
    const LmShape& sh = R->shape;
    // ---- start state.  The current estimate goes to pose buffer 0 (accepted steps flip LmDev::cur; a previous solve may have left it in 1)
    if (h->cur != h->pose_a.p) {
        UZL_HIP(hipMemcpyAsync(h->pose_a.p, h->cur, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
        h->cur = h->pose_a.p; h->trial = h->pose_b.p;
    }
    R->d_start.reserve((size_t)std::max(h->n, 1) * 8);
    UZL_HIP(hipMemcpyAsync(R->d_start.p, h->cur, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
    const bool sync_rebuild = !(h->ml_cl == 1 && h->ml_comp);     // (do_optimize_host's async_ok: small graphs only - at 10k vertices the rebuild's GEMMs take more from the overlapped PCG than they give back)
    LmDev& I = *R->h_init.p;
    memset(&I, 0, sizeof(I));
    I.flags[0] = 1;                                          // PCG kernels are no-ops until a solve is initialised
    I.phase = kLmLin; I.cur = 0; I.ix = 0;
    I.init_pass = I.schur_pass = I.numeric_pass = I.trial_pass = I.build_pass = -1;
    I.iterations = iterations;
    I.max_it = h->cfg.pcg_max_iter > 0 ? h->cfg.pcg_max_iter : 6 * std::max(h->Dp.nb, 1);
    I.always_refresh = kAlwaysRefresh ? 1 : 0; I.sync_rebuild = sync_rebuild ? 1 : 0;
    I.guarded = (h->ml_mult || h->ml_ns_steps > 0) ? 1 : 0;
    I.ni = 2.; I.last_rel = 1e300; I.rate_ref = -1.; I.rate_last = -1.;
    I.tol_f2 = pgo_tol_f2(h->cfg); I.eps_t = pgo_eps_t(h->cfg); I.eps_r = pgo_eps_r(h->cfg);
    I.refresh_rel = kRefreshRel; I.tol2 = h->cfg.pcg_tol * h->cfg.pcg_tol; I.lambda_retake = kLambdaRetake; I.delta = h->cfg.huber_delta;
    UZL_HIP(hipMemcpyAsync(R->d_lm.p, &I, sizeof(LmDev), hipMemcpyHostToDevice, s));
    memset(R->h_pub.p, 0, sizeof(LmHost));
    R->tails = 0; R->join_pending = false;
    struct Drain {                           // an exception must not leave a rebuild running on stream2 behind the handle's back
        uzl_pgo* h; LmRun* R;
        ~Drain() { if (R->join_pending) { (void)hipStreamSynchronize(h->stream2); R->join_pending = false; } }
    } drain{h, R};
    // ---- passes.  The host runs ONE pass ahead: pass k + 1 is enqueued before the snapshot of pass k has been looked at, so the GPU goes
    // from the tail of one trial into the head of the next without waiting for the host (a kernel launch costs the host ~3 us, a look
    // at a finished pass ~15 us).  What a pass carries is therefore chosen from a state one pass old:
    //   * the solve's length K from the previous solve's count - a pass that ends before its solve does is simply continued by the next
    //     (whose head and init kernels no-op), one that overshoots pays ~1.2 us per no-op launch;
    //   * the set-up segments from the refresh rule on the old state - lm_head stalls a graph whose pass lacks what it needs, and a
    //     segment nobody wants no-ops.
    // The DEVICE takes every decision from the graph's own state, so the result does not depend on what the host guessed or when it looked.
    LmHost snap;                             // the latest snapshot (the start state before the first pass)
    memset(&snap, 0, sizeof(snap));
    snap.lm = I;
    LmDev& v = snap.lm;
    constexpr int kStep = 2 * kShortPairs;
    const int kLong = 2 * kGraphPairs;
    auto round_up = [](int x) { return ((x + kStep - 1) / kStep) * kStep; };
    static const bool run_ahead = diag_flag("UZL_LM_RUN_AHEAD");            // A/B switch (diagnostic build): the next pass is enqueued before this one has been looked at
    int32_t passes = 0, in_flight = 0, solve_passes = 0;
    uint32_t seen = 0;                       // snapshots consumed
    double enq_ms = 0., wait_ms = 0.;
    // enqueue one full pass; `ahead` = passes in flight whose outcome `v` does not know yet (each assumed to complete one LM iteration)
    // diagnostic build, UZL_PHASES=1: GPU time between the segment boundaries of every pass (events on the solver's stream)
    static const bool phases_on = diag_flag("UZL_PHASES");
    std::vector<hipEvent_t> ph_ev;
    std::vector<int> ph_tag;
    auto mark = [&](int tag) {
        if (!phases_on) return;
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s); ph_ev.push_back(e); ph_tag.push_back(tag);
    };
    auto enqueue_pass = [&](int ahead) {
        const auto tp0 = std::chrono::steady_clock::now();
        int pf = 0;
        if (ahead == 0 && v.phase == kLmSolve) pf = 0;
        else if (v.phase == kLmNeedSetup) pf = ((v.need & (kNeedNumeric | kNeedTrial)) ? kPassSetup : 0) | ((v.need & kNeedRebuild) ? kPassRebuild : 0);
        else {
            const int it_guess = v.it + ahead;
            if (v.phase == kLmLin || ahead > 0) {
                if (lm_refresh(it_guess, iterations, kAlwaysRefresh, sync_rebuild, v.last_rel, kRefreshRel, v.rate_ref, v.rate_last))
                    pf |= (it_guess == 0 || sync_rebuild) ? kPassSetup : kPassRebuild;
            }
            if (ahead == 0 && v.phase == kLmLin && v.it > 0 && v.lambda > kLambdaRetake * v.lambda_setup[v.ix ^ (v.pending ? 1 : 0)]) pf |= kPassSetup;
            if (ahead == 0 && v.phase == kLmRetry && v.lambda > kLambdaRetake * v.lambda_setup[v.ix]) pf |= kPassSetup;
        }
        const bool goes_on = ahead == 0 && v.phase == kLmSolve;      // a solve that outlasted its pass (known, not guessed): PCG + tail only
        mark(0);
        if (!goes_on && R->join_pending) { UZL_HIP(hipStreamWaitEvent(s, R->ev_join, 0)); R->join_pending = false; }      // the rebuild of an earlier pass reads H and the poses
        mark(1);
        if (!goes_on) enq_head(R, pf, s);
        mark(2);
        if (pf & kPassSetup) run_seg(R->setup, eager, s, [&](hipStream_t q) { enq_setup(R, 1, q); });
        if (pf & kPassRebuild) {
            UZL_HIP(hipEventRecord(R->ev_fork, s));
            UZL_HIP(hipStreamWaitEvent(h->stream2, R->ev_fork, 0));
            run_seg(R->reb, eager, h->stream2, [&](hipStream_t q) { enq_setup(R, 0, q); });
            UZL_HIP(hipEventRecord(R->ev_join, h->stream2));
            R->join_pending = true;
        }
        mark(3);
        if (!goes_on) enq_init(R, s);
        mark(4);
        // the solve's length: the previous solve's count + 1 (a solve that the stop test ends after k iterations is declared done by the
        // ml_spmv of iteration k + 1), rounded up to a pair.  Too many is a 1.2-us no-op per launch, too few another pass.  Short solves
        // are launched kernel by kernel (~3 us of host time each, and no fixed cost: a hipGraphLaunch costs ~10 us whatever it holds),
        // long ones as captured replays of 2 x kGraphPairs iterations plus a remainder.
        // The first solve of an optimize has no predecessor: the first solve of the handle's last optimize stands in (same structure or a
        // grown one: a re-optimisation), a fresh handle starts with two long replays.
        int want = v.pcg_last > 0 ? ((v.pcg_last + 2) & ~1) : (R->first_solve_its > 0 ? ((R->first_solve_its + 2) & ~1) : 2 * kLong);
        if (goes_on) want = solve_passes < 2 ? kStep : kLong;     // nothing says how much longer: two short batches, then long ones
        solve_passes = goes_on ? solve_passes + 1 : 0;
        want = std::max(2, std::min(want, round_up(I.max_it)));
        {
            const int n_long = eager ? 0 : want / kLong, rem = want - n_long * kLong;
            for (int i = 0; i < n_long; i++) run_seg(R->pcg_long, false, s, [&](hipStream_t q) { enq_pcg(R, kGraphPairs, q); });
            if (rem > 0) enq_pcg(R, rem / 2, s);
        }
        mark(5);
        enq_tail(R, s);
        mark(6);
        R->tails++; passes++; in_flight++;
        enq_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
        if (h->cfg.verbose) fprintf(stderr, "[uzl_pgo]   pass %d enqueued: segments %d, %d PCG iterations, %d ahead (state known: it %d trial %d phase %d)\n", (int)passes - 1, pf, want, ahead, v.it, v.qmax, v.phase);
    };
    enqueue_pass(0);
    for (;;) {
        // another pass behind the one in flight - unless the known state says the in-flight passes should finish the job
        const bool more_likely = v.phase == kLmNeedSetup || v.it + in_flight < iterations;
        if (run_ahead && in_flight < 2 && more_likely) enqueue_pass(in_flight);
        const auto tw0 = std::chrono::steady_clock::now();
        const uint32_t got = wait_pub(h, R, 0, seen + 1, &snap);      // the oldest pass in flight (or a later one, if the host was slow)
        wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        in_flight -= (int32_t)(got - seen); seen = got;
        if (h->cfg.verbose)
            fprintf(stderr, "[uzl_pgo] pass %u done -> it %d trial %d phase %d: pcg %d done %d (last solve %d) lambda %.3e chi2 %.9g |r|2/|b|2 %.3e need %d\n", got - 1, v.it, v.qmax,
                    v.phase, v.flags[1], v.flags[0], v.pcg_last, v.lambda, v.chi_cur, snap.scal[7], v.need);
        if (v.st_lm_trials == 1 && v.pcg_last > 0) R->first_solve_its = v.pcg_last;
        if (v.phase == kLmDone || v.phase == kLmAnomaly) break;
        if (in_flight == 0) enqueue_pass(0);
    }
    UZL_HIP(hipGetLastError());
    if (h->cfg.verbose) fprintf(stderr, "[uzl_pgo] device-resident loop: %d passes, host time enqueueing %.3f ms, waiting for snapshots %.3f ms\n", (int)passes, enq_ms, wait_ms);
    if (R->join_pending) { UZL_HIP(hipStreamSynchronize(h->stream2)); R->join_pending = false; }      // a rebuild nobody will use: let it drain
    UZL_HIP(hipStreamSynchronize(s));
    if (phases_on && ph_ev.size() > 1) {
        double acc[7] = {0, 0, 0, 0, 0, 0, 0};
        static const char* nm[7] = {"wait for the rebuild (0->1)", "linearise + head (1->2)", "set-up / fork (2->3)", "init (3->4)", "pcg (4->5)", "tail (5->6)", "between passes (6->0)"};
        static const bool each = diag_flag("UZL_PHASES_EACH");
        for (size_t i = 0; i + 1 < ph_ev.size(); i++) {
            float ms = 0.f; (void)hipEventElapsedTime(&ms, ph_ev[i], ph_ev[i + 1]); acc[ph_tag[i] % 7] += ms;
            if (each) fprintf(stderr, "%s%d:%.0f", ph_tag[i] == 0 ? "\n[uzl_pgo]   " : " ", ph_tag[i], 1e3 * ms);
        }
        if (each) fprintf(stderr, "\n");
        fprintf(stderr, "[uzl_pgo] segments over %d passes (GPU event time, ms):", (int)passes);
        for (int k = 0; k < 7; k++) fprintf(stderr, "  %s %.3f", nm[k], acc[k]);
        fprintf(stderr, "\n");
        for (hipEvent_t e : ph_ev) (void)hipEventDestroy(e);
    }
    if (v.phase == kLmAnomaly) {             // the host-driven loop knows the remedies (retake the inverses, additive operator): from the start poses
        if (h->cfg.verbose) fprintf(stderr, "[uzl_pgo] device-resident loop: anomaly %d at it %d trial %d -> host-driven loop\n", v.anomaly_code, v.it, v.qmax);
        UZL_HIP(hipMemcpyAsync(h->pose_a.p, R->d_start.p, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
        h->cur = h->pose_a.p; h->trial = h->pose_b.p;
        return do_optimize_host(h, iterations, st);
    }
    h->cur = v.cur ? h->pose_b.p : h->pose_a.p; h->trial = v.cur ? h->pose_a.p : h->pose_b.p;
    h->prev_pcg_iters = v.pcg_last;
    h->last_residual_ratio = snap.scal[7];
    if (st) {
        uzl_pgo_stats S;
        memset(&S, 0, sizeof(S));
        S.structure_reused = h->last_structure_reused ? 1 : 0;
        S.n_vertices = h->n; S.n_edges = h->e; S.n_gauge_fixed = h->n_gauge; S.n_eliminated = h->red.on ? h->red.n_int : 0;
        S.iterations_done = v.st_iterations_done; S.lm_trials = v.st_lm_trials; S.pcg_iterations = v.st_pcg_iterations;
        S.terminated_early = v.st_terminated_early; S.precond_builds = v.st_precond_builds;
        S.chi2_initial = v.chi2_initial; S.chi2_final = v.chi_cur; S.lambda_final = v.lambda;
        S.lm_passes = passes;
        S.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->t_start).count();
        S.structure_ms = h->structure_ms;
        *st = S;
    }
    return UZL_OK;
}

}  // namespace uzl
