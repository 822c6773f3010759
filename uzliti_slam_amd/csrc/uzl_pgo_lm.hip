// uzl_pgo_lm.hip — the device-resident Levenberg-Marquardt loop behind uzl_pgo_optimize and uzl_pgo_batch_optimize.
//
// G2oOptimizer::optimizeImpl (graph_optimization/src/g2o_optimizer.cpp:137-149) hands the graph to optimizer_.optimize(iterations),
// i.e. g2o's OptimizationAlgorithmLevenberg::solve [EXT]: linearise, then trials (lambda, solve, evaluate, accept / reject) until a step
// is accepted.  uzl_pgo.hip's do_optimize_host drives that loop from the host, scalar by scalar.  Here the loop's state lives on the
// device (LmDev), its decisions are taken by two one-workgroup kernels (pgo_lm_kernels.hip), every other kernel is a slot twin that
// predicates itself on that state, and one trial is a PASS, a fixed sequence of segments:
//     head  = linearise, assemble | lm_head (chi2, lambda_0, adoption, refresh decision, setLambda) | [Schur reduction]
//     setup = numeric + trial part of the hierarchy copy in use        (first iteration; synchronous rebuilds; lambda grown 32x)
//     reb   = rebuild of the OTHER copy on the second stream           (lazy refresh: adopted by the next iteration)
//     init  = x = 0, r = b, first application of the preconditioner
//     pcg   = K iterations                                             (no-ops once the solve's `done` flag is set)
//     tail  = [back-substitution], retraction, chi2 | lm_tail (residual guard, rho, accept / reject, next phase, snapshot for the host)
// The host enqueues a pass, looks at the snapshot ONCE, and chooses the next pass's segments and K from it.  Its choices are predictions
// only: lm_head stalls a graph whose pass lacks a segment it needs (kLmNeedSetup) and the next pass brings it; a solve that outlasts
// its pass goes on in the next.  The DEVICE takes every decision from the graph's own state, so results do not depend on what the host
// guessed.  Same kernel bodies, same order of operations, same scalar arithmetic (pgo_lm.hpp) as the host-driven loop: tests hold the
// two to array_equal poses.
//
// One driver serves one graph (uzl_pgo_optimize) and many (uzl_pgo_batch_optimize): R slots, blockIdx.z = slot, a queue of graphs
// behind them.  Graphs in different phases of their loops share the launches of a pass - none waits for another's trial count - and a
// slot whose graph is through takes the next one of the queue.  The host-driven loop remains for sharded and profiled single solves
// and block-Jacobi, and takes a graph over - from its start poses - when its solve meets an anomaly (PCG breakdown, not converged,
// residual guard): it knows the remedies (retake the inverses, additive operator).
#include "pgo_handle.hpp"

namespace uzl {

struct LmRun {
    int nslots = 0;
    DevBuf<LmSlot> d_slots;
    DevBuf<LmDev> d_lm;
    PinBuf<LmHost> h_pub;                    // mapped + coherent: written by lm_tail_kernel, polled by the host
    LmHost* d_pub = nullptr;
    PinBuf<LmDev> h_init;                    // staging of the slots' initial states
    std::vector<LmSlot> slots;               // host copies of the slots (by-value launches of a one-graph pass; refill copies)
    LmShape shape{};
    DevBuf<double> d_start;                  // poses at the start of the solve (anomaly fallback)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool join_pending = false;               // a rebuild is in flight on the second stream
    uint64_t gen = ~0ull;                    // single handle: structure generation the slot and the captured segments belong to
    int solves_of_gen = 0;                   // optimizes of the current structure so far (single-graph driver)
    int first_solve_its = 0;                 // PCG iterations of the first solve of the last optimize (sizes the first pass of the next)
    // PCG iterations of every trial of the last drive's jobs and of the running one's.  A re-optimisation
    // of an unchanged graph (uzl_pgo_reset, a timer-driven one) walks through much the same solves, trial for trial: where the count RISES from one trial to
    // the next (config 2: 24 -> 40 iterations from the first trial to the second) the last solve alone under-predicts and the solve
    // needs a second and third pass (tail, look and a fresh launch sequence each)
    std::vector<std::vector<int>> trial_its_prev, trial_its_cur;      // [job][trial]
    std::vector<uint64_t> hist_gen;          // structure generation of every job the counts belong to (a batch: its graphs in order)
    struct Seg { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; };
    Seg setup, reb, pcg_long;
    void drop(Seg& q) { if (q.x) { (void)hipGraphExecDestroy(q.x); q.x = nullptr; } if (q.g) { (void)hipGraphDestroy(q.g); q.g = nullptr; } }
    void drop_all() { drop(setup); drop(reb); drop(pcg_long); }
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

LmRun* new_run()
{
    LmRun* R = new LmRun();
    if (hipEventCreateWithFlags(&R->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&R->ev_join, hipEventDisableTiming) != hipSuccess) {
        lm_run_destroy(R);
        throw HipError{hipErrorUnknown, "hipEventCreate", __FILE__, __LINE__};
    }
    return R;
}
void reserve_slots(LmRun* R, int n)
{
    R->nslots = n;
    R->d_slots.reserve((size_t)n); R->d_lm.reserve((size_t)n);
    R->h_pub.reserve((size_t)n, hipHostMallocMapped | hipHostMallocCoherent);
    R->h_init.reserve((size_t)n);
    UZL_HIP(hipHostGetDevicePointer((void**)&R->d_pub, R->h_pub.p, 0));
    R->slots.resize((size_t)n);
}

// ---- the segments of a pass, as launch sequences
void enq_linearize(LmRun* R, hipStream_t s)
{
    const LmShape& sh = R->shape;
    UZL_HIP(kl_linearize(R->d_slots.p, sh.nslots, sh.g_edges, sh.g_asm, s));
}
void enq_head(LmRun* R, int pass_flags, bool linearized, hipStream_t s)
{
    const LmShape& sh = R->shape;
    if (!linearized) enq_linearize(R, s);
    k_lm_head(R->d_slots.p, sh.nslots, pass_flags, s);
    if (sh.red) kl_schur_reduce(R->d_slots.p, sh.nslots, sh.schur_runs, (long)sh.schur_items, s);
}
void enq_setup(LmRun* R, int which, hipStream_t s)
{
    kl_ml_numeric(R->d_slots.p, R->shape, which, s);
    kl_ml_trial(R->d_slots.p, R->shape, which, s);
}
const LmSlot* by_value_slot(LmRun* R) { return (R->shape.nslots == 1 && !R->shape.batch_geometry && !lm_slot_ptr) ? R->slots.data() : nullptr; }
void enq_init(LmRun* R, hipStream_t s) { UZL_HIP(kl_ml_init(R->d_slots.p, by_value_slot(R), R->shape, s)); }
void enq_pcg(LmRun* R, int first, int n, hipStream_t s, hipEvent_t* ev = nullptr) { UZL_HIP(kl_ml_pcg_its(R->d_slots.p, by_value_slot(R), R->shape, first, n, s, ev)); }
void enq_tail(LmRun* R, hipStream_t s)
{
    const LmShape& sh = R->shape;
    if (sh.red) kl_schur_backsub(R->d_slots.p, sh.nslots, sh.schur_backsub_grid, s);
    kl_eval(R->d_slots.p, sh.nslots, sh.g_edges, sh.g_oplus, s);
    k_lm_tail(R->d_slots.p, sh.nslots, s);
}
// a long launch sequence as a captured graph (a hipGraphLaunch costs ~10 us whatever it holds, a launch ~3 us of host time: short
// sequences are launched as they are)
template <class F>
void run_seg(LmRun::Seg& q, bool eager, hipStream_t s, F&& enqueue)
{
    if (eager) { enqueue(s); return; }
    if (!q.x) {
        UZL_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        try { enqueue(s); }
        catch (...) {                                           // a launch that fails must not leave the stream in capture mode: every later
            hipGraph_t part = nullptr;                          // call on the handle (synchronize, store, destroy) would fail with it
            (void)hipStreamEndCapture(s, &part);
            if (part) (void)hipGraphDestroy(part);
            (void)hipGetLastError();
            throw;
        }
        UZL_HIP(hipStreamEndCapture(s, &q.g));
        UZL_HIP(hipGraphInstantiate(&q.x, q.g, nullptr, nullptr, 0));
    }
    UZL_HIP(hipGraphLaunch(q.x, s));
}

// the slot of a handle's current structure, its LM state in lm / snapshot in pub
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
    S.g_edges = g_edges_for(h->e); S.g_asm = h->D.n_rb; S.g_oplus = g_oplus_for(h->n);
    S.g_rows = g_ml_rows(h->Dp.nb, h->ml_agg); S.g_spmv = g_ml_spmv(h->Dp.nb, h->ml_agg);
    S.copy_stride = (int64_t)h->ml_copy_stride;
    return S;
}

// launch geometry for the graphs `hs` (one, or a batch of one hierarchy shape from level 1 up): the largest extents
LmShape make_shape(const std::vector<uzl_pgo*>& hs, int nslots, bool batch_geometry)
{
    const uzl_pgo* h = hs[0];
    for (const uzl_pgo* g : hs) if (g->ml_levels >= 1 && g->ml_n[1] > h->ml_n[1]) h = g;       // the ml_cg variant that covers the largest level 1
    LmShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.nslots = nslots; sh.batch_geometry = batch_geometry ? 1 : 0;
    sh.levels = h->ml_levels; sh.cl = h->ml_comp ? h->ml_cl : 0; sh.agg = h->ml_agg;
    sh.mult = h->ml_mult ? 1 : 0; sh.ns_steps = h->ml_ns_steps; sh.upper_ns = kUpperNs;
    ml_cg_variant(h->mlb[0].hot, h->ml_agg, h->ml_lds, &sh.cg_variant, &sh.comp_u, &sh.cg_lds);
    for (const uzl_pgo* g : hs) {
        for (int l = 0; l <= g->ml_levels; l++) { sh.n_lv[l] = std::max(sh.n_lv[l], g->ml_n[l]); sh.work_t[l] = std::max(sh.work_t[l], g->ml_nslots[l] + g->ml_n[l]); sh.chunks[l] = std::max(sh.chunks[l], g->ml_chunks[l]); }
        sh.inner_aggs = std::max(sh.inner_aggs, g->ml_inner_aggs);
        sh.g_edges = std::max(sh.g_edges, g_edges_for(g->e)); sh.g_asm = std::max(sh.g_asm, g->D.n_rb); sh.g_oplus = std::max(sh.g_oplus, g_oplus_for(g->n));
        sh.g_rows = std::max(sh.g_rows, g_ml_rows(g->Dp.nb, g->ml_agg)); sh.g_spmv = std::max(sh.g_spmv, g_ml_spmv(g->Dp.nb, g->ml_agg));
        if (g->red.on) {
            const SchurDev& SD = g->red.S;
            sh.red = 1;
            sh.schur_runs = std::max(sh.schur_runs, SD.n_runs);
            sh.schur_items = std::max<int64_t>(sh.schur_items, (int64_t)(SD.nslots_r + SD.nbr) * 36);
            sh.schur_backsub_grid = std::max(sh.schur_backsub_grid, SD.n_runs + (SD.nbr * 6 + 63) / 64);
        }
    }
    return sh;
}

// the state a graph starts its loop in
LmDev initial_state(const uzl_pgo* h, int iterations)
{
    LmDev I;
    memset(&I, 0, sizeof(I));
    I.flags[0] = 1;                                          // PCG kernels are no-ops until a solve is initialised
    I.phase = kLmLin; I.cur = 0; I.ix = 0;
    I.init_pass = I.schur_pass = I.numeric_pass = I.trial_pass = I.build_pass = -1;
    I.iterations = iterations;
    I.max_it = h->cfg.pcg_max_iter > 0 ? h->cfg.pcg_max_iter : 6 * std::max(h->Dp.nb, 1);
    I.always_refresh = kAlwaysRefresh ? 1 : 0;
    // (do_optimize_host's async_ok: rebuilds run ahead for the small-graph class only - at 10k vertices the rebuild's GEMMs take more from
    //  the overlapped PCG than they give back)
    I.sync_rebuild = (h->ml_comp && ml_async_level(h)) ? 0 : 1;
    I.guarded = (h->ml_mult || h->ml_ns_steps > 0) ? 1 : 0;
    I.ni = 2.; I.last_rel = 1e300; I.rate_ref = -1.; I.rate_last = -1.;
    I.tol_f2 = pgo_tol_f2(h->cfg); I.eps_t = pgo_eps_t(h->cfg); I.eps_r = pgo_eps_r(h->cfg);
    I.refresh_rel = ml_async_level(h) ? kRefreshRel : kRefreshRelSync; I.rate_drop = ml_rate_drop(h); I.tol2 = h->cfg.pcg_tol * h->cfg.pcg_tol; I.lambda_retake = kLambdaRetake; I.delta = h->cfg.huber_delta;
    return I;
}
LmDev idle_state()
{
    LmDev I;
    memset(&I, 0, sizeof(I));
    I.flags[0] = 1; I.phase = kLmDone;
    I.init_pass = I.schur_pass = I.numeric_pass = I.trial_pass = I.build_pass = -1;
    return I;
}

// Waits until lm_tail launch number `seq` (or a later one) of slot `sl` has published and copies the snapshot; returns its number.
// lm_tail writes the fields and seq_begin (unordered among themselves), waits until they have been acknowledged, then seq
// (publish_wait_own_stores, uzl_common.hpp).  What keeps a copy whole is the DRIVER'S ORDER: a slot's snapshot is copied here before the
// pass whose tail writes the next one is enqueued; seq_begin == seq around the copy cross-checks it.
uint32_t wait_pub(hipStream_t s, LmRun* R, int sl, uint32_t seq, LmHost* out)
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
            UZL_HIP(hipStreamSynchronize(s));
            const uint32_t s2 = __atomic_load_n(&R->h_pub.p[sl].seq, __ATOMIC_ACQUIRE);
            if ((int32_t)(s2 - seq) < 0) throw HipError{hipErrorUnknown, "lm_tail_kernel did not publish", __FILE__, __LINE__};
            memcpy(out, const_cast<const LmHost*>(R->h_pub.p + sl), sizeof(LmHost));      // (the stream is idle: nothing writes the snapshot now)
            if (out->seq_begin != out->seq) throw HipError{hipErrorUnknown, "lm_tail_kernel: torn snapshot on an idle stream", __FILE__, __LINE__};
            return s2;
        }
    }
}

// one graph of a drive
struct LmJob {
    uzl_pgo* h = nullptr;
    size_t start_off = 0;                    // its start poses in LmRun::d_start
    LmHost last;                             // the snapshot it finished with
    bool finished = false, anomaly = false;
    int32_t passes = 0;
};

struct LmDriveOpts {
    hipStream_t s = nullptr, s2 = nullptr;
    int iterations = 0;
    bool eager = false;                      // no captured graphs (UZL_NO_GRAPH=1, profiling)
    bool verbose = false;
    KernelTimer* timer = nullptr;            // profiling: the two PCG kernels with dispatch timestamps
    const char* spmv_name = "pcg_spmv";
    const char* cg_name = "ml_cg";
};

// Solves `jobs` through the slots of R (shape and slot count set up by the caller; structures prepared).  Returns passes enqueued.
int lm_drive(LmRun* R, std::vector<LmJob>& jobs, const LmDriveOpts& o)
{
    hipStream_t s = o.s;
    const LmShape& sh = R->shape;
    const int nS = sh.nslots, Q = (int)jobs.size();
    const bool eager = o.eager;
    constexpr int kStep = 2 * kShortPairs;
    const int kLong = 2 * kGraphPairs;
    // ---- start poses of every job (anomaly fallback); its current estimate goes to pose buffer 0 (accepted steps flip LmDev::cur; an
    //      earlier solve may have left it in buffer 1)
    size_t tot = 0;
    for (LmJob& j : jobs) { j.start_off = tot; tot += (size_t)std::max(j.h->n, 1) * 8; }
    R->d_start.reserve(tot);
    for (LmJob& j : jobs) {
        uzl_pgo* h = j.h;
        if (h->cur != h->pose_a.p) {
            UZL_HIP(hipMemcpyAsync(h->pose_a.p, h->cur, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
            h->cur = h->pose_a.p; h->trial = h->pose_b.p;
        }
        UZL_HIP(hipMemcpyAsync(R->d_start.p + j.start_off, h->cur, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
    }
    std::vector<int> slot_job((size_t)nS, -1), solve_passes((size_t)nS, 0), prev_last((size_t)nS, 0), seen_trials((size_t)nS, 0), cur_last((size_t)nS, 0);
    std::vector<uint32_t> sent((size_t)nS, 0);               // per slot: lm_tail launches enqueued since its load (= the sequence word expected)
    std::vector<LmHost> snap((size_t)nS);
    int next_job = 0, n_active = 0;
    // The next job of the queue into slot sl (stream-ordered behind what the slot ran).  A REFILL keeps the slot's sequence word running
    // (LmDev::tails starts at what the slot has sent): lm_tail_kernel publishes for every slot in every pass and the look waits for the
    // unfinished ones only, so the last snapshot of a finished or idle slot may still be on its way to h_pub when the slot is reloaded -
    // with the sequence restarted at 0 that late write (an old, larger number with phase = done) would pass for the new graph's.  Only
    // the first load of a drive (everything drained by the previous drive's final synchronize) zeroes the snapshot.
    auto load_slot = [&](int sl, bool first) {
        LmDev I;
        if (next_job < Q) {
            const int j = next_job++;
            slot_job[sl] = j;
            R->slots[sl] = make_slot(jobs[j].h, R->d_lm.p + sl, R->d_pub + sl);
            UZL_HIP(hipMemcpyAsync(R->d_slots.p + sl, &R->slots[sl], sizeof(LmSlot), hipMemcpyHostToDevice, s));
            I = initial_state(jobs[j].h, o.iterations);
            n_active++;
        } else {
            slot_job[sl] = -1;
            I = idle_state();                                // (the slot keeps its last graph's arguments: every kernel no-ops on the state)
        }
        if (first) { memset(R->h_pub.p + sl, 0, sizeof(LmHost)); sent[sl] = 0; }
        I.tails = (int32_t)sent[sl];
        R->h_init.p[sl] = I;
        UZL_HIP(hipMemcpyAsync(R->d_lm.p + sl, R->h_init.p + sl, sizeof(LmDev), hipMemcpyHostToDevice, s));
        memset(&snap[sl], 0, sizeof(LmHost));
        snap[sl].lm = I;
        solve_passes[sl] = 0; prev_last[sl] = 0; cur_last[sl] = 0; seen_trials[sl] = 0;
    };
    for (int sl = 0; sl < nS; sl++) load_slot(sl, true);
    R->join_pending = false;
    {   // the last drive's counts are this drive's history where job j is the same structure again
        bool same = R->hist_gen.size() == (size_t)Q;
        for (int j = 0; same && j < Q; j++) same = R->hist_gen[(size_t)j] == jobs[(size_t)j].h->structure_gen;
        if (same) R->trial_its_prev.swap(R->trial_its_cur); else R->trial_its_prev.clear();
        R->trial_its_prev.resize((size_t)Q);
        R->trial_its_cur.assign((size_t)Q, std::vector<int>());
        R->hist_gen.resize((size_t)Q);
        for (int j = 0; j < Q; j++) R->hist_gen[(size_t)j] = jobs[(size_t)j].h->structure_gen;
    }
    struct Drain {                           // an exception must not leave a rebuild running on stream2 behind the caller's back
        LmRun* R; hipStream_t s2;
        ~Drain() { if (R->join_pending) { (void)hipStreamSynchronize(s2); R->join_pending = false; } }
    } drain{R, o.s2};
    // diagnostic build, UZL_PHASES=1: GPU time between the segment boundaries of every pass (events on the solver's stream)
    static const bool phases_on = diag_flag("UZL_PHASES");
    std::vector<hipEvent_t> ph_ev;
    std::vector<int> ph_tag;
    auto mark = [&](int tag) {
        if (!phases_on) return;
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s); ph_ev.push_back(e); ph_tag.push_back(tag);
    };
    int passes = 0;
    double enq_ms = 0., wait_ms = 0.;
    // The next pass's linearisation goes out right behind a pass's tail, BEFORE the look: hessian_lm_kernel takes nothing from the host
    // (it works on the graphs the tail has left in phase kLmLin and is a no-op for the others), so the GPU builds the Hessian while the
    // host reads the snapshot, chooses the next pass and launches it - otherwise ~12 us of idle GPU per pass (tail -> hessian in the
    // kernel trace).  Not while a rebuild on the second stream still reads H, and a refill sends the linearisation again.
    static const bool lin_ahead_on = diag_int("UZL_LM_LIN_AHEAD", 1) != 0;      // A/B switch
    bool lin_ahead = false;
    while (n_active > 0) {
        const auto tp0 = std::chrono::steady_clock::now();
        // ---- what this pass carries: predictions from the slots' last snapshots
        int pf = 0, want = 0;
        bool any_start = false;
        std::vector<int> wants;
        for (int sl = 0; sl < nS; sl++) {
            if (slot_job[sl] < 0 || jobs[slot_job[sl]].finished) continue;      // (a finished graph waits in its slot for its cohort: every kernel no-ops on its state)
            const LmDev& v = snap[sl].lm;
            int w;
            if (v.phase == kLmSolve) {
                // A solve that outlasted its pass.  Nothing says how much longer it takes, but a launch past the end is a 1.2-us no-op
                // and another pass is 50 - 100 us of tail, look and restart: a short batch, then half of what the solve has taken so far (round
                // 4: two batches of 4, then 16s - config 2's second trial, 48 iterations against 30 predicted, took four passes; now three)
                w = solve_passes[sl] == 0 ? kStep : std::min(std::max(kStep, (v.flags[1] / 2 + 1) & ~1), 4 * kLong);      // (most often it was one launch short: a short batch first)
                solve_passes[sl]++;
            } else {
                any_start = true; solve_passes[sl] = 0;
                if (v.phase == kLmNeedSetup) pf |= ((v.need & (kNeedNumeric | kNeedTrial)) ? kPassSetup : 0) | ((v.need & kNeedRebuild) ? kPassRebuild : 0);
                else if (v.phase == kLmLin) {
                    if (lm_refresh(v.it, v.iterations, kAlwaysRefresh, v.sync_rebuild != 0, v.last_rel, v.refresh_rel, v.rate_ref, v.rate_last, v.rate_drop))
                        pf |= (v.it == 0 || v.sync_rebuild) ? kPassSetup : kPassRebuild;
                    if (v.it > 0 && v.lambda > kLambdaRetake * v.lambda_setup[v.ix ^ (v.pending ? 1 : 0)]) pf |= kPassSetup;
                } else if (v.phase == kLmRetry) {
                    if (v.lambda > kLambdaRetake * v.lambda_setup[v.ix]) pf |= kPassSetup;
                }
                // The solve's length: the previous solve's count + 1 (a solve that the stop test ends after k iterations is declared done by
                // the ml_spmv of iteration k + 1): too many is a ~3.5-us no-op per launch, too few another pass.  The
                // first solve of an optimize has no predecessor: the first solve of the last optimize stands in (same structure or a grown
                // one: a re-optimisation), a fresh run starts with two long replays.
                // uzl_pgo_cfg::pass_history = 1: nothing an earlier optimize of this handle learned sizes a pass (the first solve starts with
                // two long replays, every later one follows its predecessor in the same optimize)
                static const bool no_history_env = diag_flag("UZL_LM_NO_HISTORY");           // A/B switch
                const bool no_history = no_history_env || jobs[(size_t)slot_job[sl]].h->cfg.pass_history == 1;
                // One graph: exactly count + 1 launches (odd or even; a continuation pass picks the parity up from the solve's own
                // iteration count).  A batch keeps whole pairs: its graphs share the launches' parity.
                static const bool odd_ok = diag_int("UZL_LM_ODD_K", 1) != 0;      // A/B switch
                auto up = [&](int count) { return (nS == 1 && odd_ok) ? count + 1 : ((count + 2) & ~1); };
                w = v.pcg_last > 0 ? up(v.pcg_last) : ((R->first_solve_its > 0 && !no_history) ? up(R->first_solve_its) : 2 * kLong);
                // counts that RISE from trial to trial (lambda falls after accepted steps, the system gets harder: chain-like graphs climb by 2
                // per trial for ten trials, each time one launch short of `last + 2`): extrapolate the last rise
                if (v.pcg_last > 0 && prev_last[sl] > 0 && v.pcg_last > prev_last[sl]) w = up(v.pcg_last + std::min(v.pcg_last - prev_last[sl], 16));
                {
                    const std::vector<int>& hist = R->trial_its_prev[(size_t)slot_job[sl]];
                    if (!no_history && (size_t)v.st_lm_trials < hist.size()) w = std::max(w, up(hist[(size_t)v.st_lm_trials]));
                }
                w = std::min(w, ((v.max_it + 1) & ~1));
            }
            want = std::max(want, w);
            wants.push_back(w);
        }
        static const int k_pct = diag_int("UZL_BATCH_K_PCT", 100);      // A/B switch: the pass's PCG count as a percentile of the slots' predictions (100 = the longest)
        if (k_pct < 100 && wants.size() > 1) { std::sort(wants.begin(), wants.end()); want = wants[std::min(wants.size() - 1, (wants.size() * (size_t)k_pct) / 100)]; }
        want = std::max(2, want);
        // iteration index of the pass's first PCG launch: a single graph's continuation goes on where its solve stands (its count may be
        // odd); in a batch every pass holds whole pairs, so every solve stands at an even count
        const int base = (nS == 1 && slot_job[0] >= 0 && snap[0].lm.phase == kLmSolve) ? snap[0].lm.flags[1] : 0;
        mark(0);
        if (any_start && R->join_pending) { UZL_HIP(hipStreamWaitEvent(s, R->ev_join, 0)); R->join_pending = false; }      // the rebuild of an earlier pass reads H and the poses
        mark(1);
        if (any_start) enq_head(R, pf, lin_ahead, s);
        lin_ahead = false;
        mark(2);
        if (pf & kPassSetup) {
            // The synchronous set-ups of the first kNsEarlyIts linearisations refine the dense operator with two Newton-Schulz steps where the
            // structure asks for four (large loopy graphs): the problem still changes wholesale there and a rougher operator costs no PCG
            // iterations (ml_ns_steps_at, uzl_pgo.hip).  Launched as they are - the captured segment holds the full sequence.
            const int steps = nS == 1 ? ml_ns_steps_at(R->shape.ns_steps, snap[0].lm.it) : R->shape.ns_steps;
            if (steps != R->shape.ns_steps) { const int keep = R->shape.ns_steps; R->shape.ns_steps = steps; enq_setup(R, 1, s); R->shape.ns_steps = keep; }
            else run_seg(R->setup, eager, s, [&](hipStream_t q) { enq_setup(R, 1, q); });
        }
        if (pf & kPassRebuild) {
            UZL_HIP(hipEventRecord(R->ev_fork, s));
            UZL_HIP(hipStreamWaitEvent(o.s2, R->ev_fork, 0));
            run_seg(R->reb, eager, o.s2, [&](hipStream_t q) { enq_setup(R, 0, q); });
            UZL_HIP(hipEventRecord(R->ev_join, o.s2));
            R->join_pending = true;
        }
        mark(3);
        if (any_start) enq_init(R, s);
        mark(4);
        // short solves are launched kernel by kernel, long ones as captured replays of 2 x kGraphPairs iterations plus a remainder
        if (o.timer && o.timer->on) {
            std::vector<hipEvent_t> ev((size_t)4 * want);
            for (int q = 0; q < want; q++) { o.timer->pair(o.spmv_name, &ev[4 * q], &ev[4 * q + 1]); o.timer->pair(o.cg_name, &ev[4 * q + 2], &ev[4 * q + 3]); }
            enq_pcg(R, base, want, s, ev.data());
        } else {
            int first = base, left = want;
            if (!eager && (first & 1)) { enq_pcg(R, first, 1, s); first++; left--; }      // (the captured replay starts at an even iteration)
            const int n_long = eager ? 0 : left / kLong, rem = left - n_long * kLong;
            for (int i = 0; i < n_long; i++) run_seg(R->pcg_long, false, s, [&](hipStream_t q) { enq_pcg(R, 0, kLong, q); });
            if (rem > 0) enq_pcg(R, first + n_long * kLong, rem, s);
        }
        mark(5);
        enq_tail(R, s);
        if (lin_ahead_on && !R->join_pending && !(o.timer && o.timer->on)) { enq_linearize(R, s); lin_ahead = true; }
        mark(6);
        passes++;
        for (int sl = 0; sl < nS; sl++) sent[sl]++;
        const auto tp1 = std::chrono::steady_clock::now();
        if (o.verbose) fprintf(stderr, "[uzl_pgo]   pass %d enqueued: segments %d, %d PCG iterations, %d graph(s) in the slots\n", passes - 1, pf, want, n_active);
        // ---- the look: every slot's snapshot of this pass
        bool refill = false;
        for (int sl = 0; sl < nS; sl++) {
            if (slot_job[sl] < 0 || jobs[slot_job[sl]].finished) continue;
            (void)wait_pub(s, R, sl, sent[sl], &snap[sl]);
            LmJob& J = jobs[slot_job[sl]];
            const LmDev& v = snap[sl].lm;
            J.passes++;
            if (o.verbose)
                fprintf(stderr, "[uzl_pgo] pass %d, graph %d -> it %d trial %d phase %d: pcg %d done %d (last solve %d) lambda %.3e chi2 %.9g |r|2/|b|2 %.3e need %d\n", passes - 1,
                        slot_job[sl], v.it, v.qmax, v.phase, v.flags[1], v.flags[0], v.pcg_last, v.lambda, v.chi_cur, snap[sl].scal[7], v.need);
            if (v.st_lm_trials > seen_trials[sl]) { seen_trials[sl] = v.st_lm_trials; prev_last[sl] = cur_last[sl]; cur_last[sl] = v.pcg_last; }      // a trial ended: its count and the one before
            if (v.st_lm_trials == 1 && v.pcg_last > 0 && slot_job[sl] == 0) R->first_solve_its = v.pcg_last;
            {
                std::vector<int>& cur = R->trial_its_cur[(size_t)slot_job[sl]];
                if (v.st_lm_trials > 0 && (size_t)v.st_lm_trials > cur.size()) cur.resize((size_t)v.st_lm_trials, v.pcg_last);
            }
            if (v.phase == kLmDone || v.phase == kLmAnomaly) {
                J.last = snap[sl]; J.finished = true; J.anomaly = v.phase == kLmAnomaly;
                n_active--; refill = true;
            }
        }
        if (o.timer && o.timer->on) { UZL_HIP(hipStreamSynchronize(s)); o.timer->resolve(); }
        // A queue longer than the slots is worked off in COHORTS: the slots are refilled when every resident graph is through, so that the
        // residents walk through their LM iterations together - their solves need similar numbers of PCG iterations, and a pass is as long
        // as its longest solve.  Refilling slot by slot (UZL_BATCH_FREE_RUNNING=1, diagnostic build) mixes converged graphs (4 iterations
        // per solve) with fresh ones (60): measured on 256 queued config-2 graphs through 16 / 64 slots 49 / 69 M edges/s against 60 / 74 M.
        static const bool free_running = diag_flag("UZL_BATCH_FREE_RUNNING");
        if (refill && (free_running || n_active == 0 || nS == 1)) {
            // (a rebuild of the outgoing graphs may still read the slot table and their LM state)
            if (R->join_pending && next_job < Q) { UZL_HIP(hipStreamWaitEvent(s, R->ev_join, 0)); R->join_pending = false; }
            for (int sl = 0; sl < nS; sl++)
                if (slot_job[sl] >= 0 && jobs[slot_job[sl]].finished) load_slot(sl, false);
            lin_ahead = false;                              // (the new graphs were not in their slots when the linearisation ran)
        }
        const auto tp2 = std::chrono::steady_clock::now();
        enq_ms += std::chrono::duration<double, std::milli>(tp1 - tp0).count(); wait_ms += std::chrono::duration<double, std::milli>(tp2 - tp1).count();
    }
    UZL_HIP(hipGetLastError());
    if (o.verbose) fprintf(stderr, "[uzl_pgo] device-resident loop: %d passes, host time enqueueing %.3f ms, waiting for snapshots %.3f ms\n", passes, enq_ms, wait_ms);
    if (R->join_pending) { UZL_HIP(hipStreamSynchronize(o.s2)); R->join_pending = false; }      // a rebuild nobody will use: let it drain
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
        fprintf(stderr, "[uzl_pgo] segments over %d passes (GPU event time, ms):", passes);
        for (int k = 0; k < 7; k++) fprintf(stderr, "  %s %.3f", nm[k], acc[k]);
        fprintf(stderr, "\n");
        for (hipEvent_t e : ph_ev) (void)hipEventDestroy(e);
    }
    return passes;
}

// what a finished job leaves in its handle and in the caller's stats
void finish_job(LmJob& J, uzl_pgo_stats* st, double wall_ms)
{
    uzl_pgo* h = J.h;
    const LmDev& v = J.last.lm;
    h->cur = v.cur ? h->pose_b.p : h->pose_a.p; h->trial = v.cur ? h->pose_a.p : h->pose_b.p;
    h->prev_pcg_iters = v.pcg_last;
    h->last_residual_ratio = J.last.scal[7];
    if (!st) return;
    uzl_pgo_stats S;
    memset(&S, 0, sizeof(S));
    S.structure_reused = h->last_structure_reused ? 1 : 0;
    S.n_vertices = h->n; S.n_edges = h->e; S.n_gauge_fixed = h->n_gauge; S.n_eliminated = h->red.on ? h->red.n_int : 0; S.reduced_strong = (h->red.on && h->red.strong) ? 1 : 0;
    S.iterations_done = v.st_iterations_done; S.lm_trials = v.st_lm_trials; S.pcg_iterations = v.st_pcg_iterations;
    S.terminated_early = v.st_terminated_early; S.precond_builds = v.st_precond_builds;
    S.chi2_initial = v.chi2_initial; S.chi2_final = v.chi_cur; S.lambda_final = v.lambda;
    S.lm_passes = J.passes;
    S.solve_ms = wall_ms;
    S.structure_ms = h->structure_ms;
    *st = S;
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
    if (!h->lm) h->lm = new_run();
    LmRun* R = h->lm;
    // ---- shape of this structure (captured segments belong to it)
    if (R->gen != h->structure_gen || R->nslots != 1) {
        const auto ts = std::chrono::steady_clock::now();
        UZL_HIP(hipStreamSynchronize(s));
        R->drop_all();
        reserve_slots(R, 1);
        R->shape = make_shape({h}, 1, false);
        R->gen = h->structure_gen;
        R->solves_of_gen = 0;
        R->hist_gen.clear();                                       // (a grown graph's solves are not the old graph's: measured on config 5, +5 % per solve with the old counts)
        h->structure_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
    }
    std::vector<LmJob> jobs(1);
    jobs[0].h = h;
    LmDriveOpts o;
    // The FIRST solve of a structure launches everything kernel by kernel: capturing and instantiating its segments costs ~0.5 ms and dropping
    // them again ~0.7 ms, which a structure that is solved once - every re-optimisation of a growing graph - never earns back (config 5:
    // 16.2 -> 15.7 ms per solve, structure 1.9 -> 1.2 ms).  A structure that comes back (uzl_pgo_reset, a timer-driven re-optimisation of an
    // unchanged graph) replays captured segments from its second solve on.  Same kernels, same results either way.
    static const bool capture_first = diag_flag("UZL_LM_CAPTURE_FIRST");       // A/B switch
    o.s = s; o.s2 = h->stream2; o.iterations = iterations; o.eager = h->no_graph || (R->solves_of_gen == 0 && !capture_first); o.verbose = h->cfg.verbose != 0;
    R->solves_of_gen++;
    lm_drive(R, jobs, o);
    LmJob& J = jobs[0];
    if (J.anomaly) {             // the host-driven loop knows the remedies (retake the inverses, additive operator): from the start poses
        if (h->cfg.verbose) fprintf(stderr, "[uzl_pgo] device-resident loop: anomaly %d at it %d trial %d -> host-driven loop\n", J.last.lm.anomaly_code, J.last.lm.it, J.last.lm.qmax);
        UZL_HIP(hipMemcpyAsync(h->pose_a.p, R->d_start.p + J.start_off, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
        h->cur = h->pose_a.p; h->trial = h->pose_b.p;
        return do_optimize_host(h, iterations, st);
    }
    finish_job(J, st, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->t_start).count());
    return UZL_OK;
}

// ---- many graphs through one launch sequence (uzl_pgo_batch_optimize) ---------------------------------------------------------
// Batched together are graphs of the small-graph class (dense level-1 operator, multiplicative cycle: the class whose kernels have a
// throughput geometry - four rows per wave in 128-lane workgroups, wave sums through the LDS crossbar; same bits as the single
// solve's geometry) with one hierarchy shape from level 1 up, Schur-reduced or not: the shape of the SYSTEM THE PCG SOLVES decides, so
// chain-like graphs (graph_slam_node.cpp:578-663: an odometry chain plus a few loop closures) batch on their reduced systems.
bool lm_batch_eligible(const std::vector<uzl_pgo*>& hs)
{
    if (lm_host_forced || hs.empty() || (int)hs.size() > kBatchMax) return false;
    const uzl_pgo* a = hs[0];
    for (const uzl_pgo* h : hs) {
        if (!(h->cfg.lm_loop != 1 && h->ml_levels > 0 && h->ml_agg == 1 && h->ml_comp && h->ml_mult && h->ml_cl == 1 && 6 * h->ml_n[1] <= 1536 && !h->sharded && h->nb > 0 &&
              h->e > 0 && h->Dp.nb > 0 && !h->timer.on)) return false;
        // one depth of the hierarchy (the launch sequence of a pass): sizes may differ - every launch takes the largest graph's grid and a
        // twin leaves past its own graph's extent
        if (h->ml_levels != a->ml_levels || h->ml_ns_steps != a->ml_ns_steps || h->cfg.device != a->cfg.device) return false;
    }
    return true;
}

// returns graphs solved by the batch (anomalies go through the single-graph path and are not counted); < 0: graph -1 - g failed with *rc_all
int batch_optimize_lm(LmRun*& Rp, const std::vector<uzl_pgo*>& hs, int resident, hipStream_t s, hipStream_t s2, int32_t iterations, bool eager, bool verbose, KernelTimer* timer,
                      uzl_pgo_stats* stats, int* rc_all)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (!Rp) Rp = new_run();
    LmRun* R = Rp;
    const int Q = (int)hs.size(), nS = std::max(1, std::min(resident > 0 ? resident : Q, Q));
    const LmShape sh = make_shape(hs, nS, true);
    if (R->nslots != nS || memcmp(&sh, &R->shape, sizeof(LmShape)) != 0) {      // captured segments hold the grids
        UZL_HIP(hipStreamSynchronize(s));
        R->drop_all();
        reserve_slots(R, nS);
        R->shape = sh;
    }
    std::vector<LmJob> jobs((size_t)Q);
    for (int g = 0; g < Q; g++) jobs[g].h = hs[g];
    LmDriveOpts o;
    o.s = s; o.s2 = s2; o.iterations = iterations; o.eager = eager || (timer && timer->on); o.verbose = verbose; o.timer = timer;
    o.spmv_name = "ml_spmv_batch"; o.cg_name = "ml_cg_comp_batch";
    lm_drive(R, jobs, o);
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int batched = 0;
    for (int g = 0; g < Q; g++) {
        LmJob& J = jobs[g];
        uzl_pgo* h = hs[g];
        // (diagnostic build: UZL_BATCH_FORCE_ANOMALY_N = n sends every graph of n nodes down the anomaly path - tests/test_batch_gpu.py)
        static const int force_n = diag_int("UZL_BATCH_FORCE_ANOMALY_N", -1);
        if (!J.anomaly && h->n != force_n) { finish_job(J, stats ? stats + g : nullptr, wall); batched++; continue; }
        if (verbose) fprintf(stderr, "[uzl_pgo_batch] graph %d: anomaly %d at it %d trial %d -> single-graph path\n", g, J.last.lm.anomaly_code, J.last.lm.it, J.last.lm.qmax);
        UZL_HIP(hipMemcpyAsync(h->pose_a.p, R->d_start.p + J.start_off, sizeof(double) * 8 * (size_t)h->n, hipMemcpyDeviceToDevice, s));
        UZL_HIP(hipStreamSynchronize(s));
        h->cur = h->pose_a.p; h->trial = h->pose_b.p;
        own_streams(h, false);                             // (sequence 0 may be capturing on the streams h borrowed: not synchronized, h has nothing on them)
        h->t_start = std::chrono::steady_clock::now();
        uzl_pgo_stats S;
        const int rc = do_optimize_host(h, iterations, &S);
        if (rc != UZL_OK && rc != UZL_ERR_NOT_CONVERGED) { *rc_all = rc; return -1 - g; }
        if (rc != UZL_OK) *rc_all = rc;
        if (stats) stats[g] = S;
    }
    return batched;
}

}  // namespace uzl
