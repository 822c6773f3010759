// pgo_lm_kernels.hip — the two decision kernels of the device-resident Levenberg-Marquardt loop (pgo_types.hpp: LmDev / LmSlot).
//
//   lm_head_kernel : in front of a trial's solve.  For a graph that has just been linearised: chi2 and the largest diagonal entry from
//                    the block partials (finalize_kernel's sums, same order), lambda_0 in the first iteration (computeLambdaInit),
//                    adoption of a hierarchy copy that was rebuilt ahead, the lazy-refresh decision (pgo_lm.hpp).  For every graph that
//                    starts a trial (fresh linearisation, rejected step, or a stalled start): setLambda, the stamps that switch this
//                    pass's Schur reduction / set-up / rebuild / PCG-init kernels on, phase -> solve.
//   lm_tail_kernel : behind the evaluation of a trial.  Convergence and residual-guard checks of the solve, chi2 of the trial and
//                    computeScale from the block partials, rho, accept / reject (pgo_lm.hpp: lm_step), next phase; then the snapshot
//                    the host polls (LmHost, pinned).
// One workgroup per graph (blockIdx.z = slot).  This is OptimizationAlgorithmLevenberg::solve [EXT] of the reference's
// optimizer_.optimize(iterations) (graph_optimization/src/g2o_optimizer.cpp:148), decision for decision what uzl_pgo.hip's host-driven
// loop takes - tests hold the two loops to identical poses.  Compiled with -ffp-contract=off like every file that shares arithmetic
// with the host.
#include "pgo_device.hpp"
#include "pgo_lm.hpp"

namespace uzl {

// lane 0 of lm_head_kernel, on the staged state
__device__ __forceinline__ void lm_head_decide(const LmSlot& S, LmDev* lm, int phase, double chi, double dmax, int pass_flags)
{
    const int p = lm->pass + 1;
    lm->pass = p;
    double* __restrict__ scal = S.D.scal;
    int need = 0;
    if (phase == kLmLin) {
        scal[4] = chi; scal[6] = dmax;
        if (lm->it == 0) {
            lm->chi_cur = chi; lm->chi2_initial = chi;
            lm->lambda = 1e-5 * dmax;                        // computeLambdaInit: tau * max diag
            lm->ni = 2.;
        }
        lm->adopted = 0;
        if (lm->pending) { lm->ix ^= 1; lm->pending = 0; lm->adopted = 1; }      // the copy built during the last iteration
        if (lm_refresh(lm->it, lm->iterations, lm->always_refresh != 0, lm->sync_rebuild != 0, lm->last_rel, lm->refresh_rel, lm->rate_ref, lm->rate_last, lm->rate_drop)) {
            lm->st_precond_builds++;
            need |= (lm->it == 0 || lm->sync_rebuild) ? (kNeedNumeric | kNeedTrial) : kNeedRebuild;
        }
        lm->qmax = 0;
        if (S.red) lm->schur_pass = p;                       // (H + lambda I) with the chain interiors eliminated: per lambda
    } else if (phase == kLmRetry) {
        if (S.red) lm->schur_pass = p;                       // a rejected step moved lambda: the Schur complement with it
    } else if (phase == kLmNeedSetup) {
        need = lm->need;                                     // (its Schur reduction ran in the pass that stalled)
    } else {
        return;                                              // solving (the pass goes on iterating), done, anomaly
    }
    // ---- start of a trial: setLambda, this trial's PCG floor and step accuracy
    const double lambda = lm->lambda;
    scal[3] = lambda; scal[8] = lm->tol_f2; scal[12] = lm->eps_t; scal[13] = lm->eps_r;
    // the lambda-dependent inverses of the hierarchy are kept across trials; after rejected steps lambda grows geometrically and
    // inverses taken at a much smaller lambda stop being a preconditioner at all
    if (lambda > lm->lambda_retake * lm->lambda_setup[lm->ix]) need |= kNeedTrial;
    if (((need & (kNeedNumeric | kNeedTrial)) && !(pass_flags & kPassSetup)) || ((need & kNeedRebuild) && !(pass_flags & kPassRebuild))) {
        lm->need = need; lm->phase = kLmNeedSetup; lm->flags[0] = 1;              // this pass lacks the segment: the PCG kernels stay no-ops
        return;
    }
    if (need & kNeedRebuild) {                               // into the copy the PCG does not use, with this iteration's lambda; adopted next iteration
        lm->build_pass = p; lm->build_ix = lm->ix ^ 1; lm->build_cur = lm->cur;
        lm->scal2[3] = lambda; lm->lambda_setup[lm->ix ^ 1] = lambda; lm->pending = 1;
    }
    if (need & kNeedNumeric) lm->numeric_pass = p;
    if (need & kNeedTrial) { lm->trial_pass = p; lm->lambda_setup[lm->ix] = lambda; }
    lm->fresh = ((need & kNeedTrial) || (lm->adopted && lm->qmax == 0)) ? 1 : 0;
    lm->need = 0;
    lm->init_pass = p;
    lm->phase = kLmSolve;                                    // (flags[0..3] are cleared by this pass's ml_init)
}


__global__ __launch_bounds__(kBlk) void lm_head_kernel(const LmSlot* __restrict__ slots, int pass_flags)
{
    __shared__ double s4[4];
    __shared__ LmDev L;                                      // the state, staged: lane 0's decisions are a chain of reads and writes of its
                                                             // fields, each a round trip to memory when taken from the global copy
    const LmSlot& S = slots[blockIdx.z];
    LmDev* lm = &L;
    constexpr int kLmWords = (int)(sizeof(LmDev) / 8);
    static_assert(sizeof(LmDev) % 8 == 0 && kLmWords <= kBlk, "LmDev is copied as 8-byte words");
    if ((int)threadIdx.x < kLmWords) reinterpret_cast<unsigned long long*>(&L)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(S.lm)[threadIdx.x];
    __syncthreads();
    const int phase = lm->phase;
    double chi = 0., dmax = 0.;
    if (phase == kLmLin) {                                   // (uniform) finalize_kernel(what = 2): chi2 + max diagonal
        chi = sum_partials(S.D.part_a, S.g_edges, s4);
        double v = 0.;
        for (int i = threadIdx.x; i < S.g_asm; i += kBlk) v = fmax(v, S.D.part_c[i]);
        dmax = block_max(v, s4);
    }
    if (threadIdx.x == 0) lm_head_decide(S, lm, phase, chi, dmax, pass_flags);
    __syncthreads();
    if ((int)threadIdx.x < kLmWords) reinterpret_cast<unsigned long long*>(S.lm)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(&L)[threadIdx.x];
}

// The tail runs 1024 lanes: the residual guard's sums (residual_guard_kernel's order), then the first 256 fold the chi2 / scale
// partials in finalize_kernel's order, lane 0 decides, and the first lanes copy the state to the host word by word.
constexpr int kTailBlk = 1024;
__global__ __launch_bounds__(kTailBlk) void lm_tail_kernel(const LmSlot* __restrict__ slots)
{
    __shared__ double s4[4], sr[16], sb[16], sscal[16];
    __shared__ LmDev L;                                      // the state and the scalars, staged (as in lm_head_kernel)
    const LmSlot& S = slots[blockIdx.z];
    LmDev* lm = &L;
    const int tid = threadIdx.x;
    constexpr int kLmWords = (int)(sizeof(LmDev) / 8);
    static_assert(sizeof(LmDev) % 8 == 0 && kLmWords + 16 <= kTailBlk, "LmDev is copied as 8-byte words");
    if (tid < kLmWords) reinterpret_cast<unsigned long long*>(&L)[tid] = reinterpret_cast<const unsigned long long*>(S.lm)[tid];
    else if (tid < kLmWords + 16) sscal[tid - kLmWords] = S.D.scal[tid - kLmWords];
    __syncthreads();
    const bool solving = lm->phase == kLmSolve;
    const bool done = lm->flags[0] != 0;
    const bool eval = solving && done && lm->flags[2] == 0;      // (uniform) the evaluation kernels of this pass ran for this graph
    double chi_t = 0., sc = 0., ratio = 0.;
    if (eval) {
        // |r|^2 / |b|^2 with the recurrence residual r (= b - (H + lambda) x up to rounding whatever the preconditioner did): what
        // residual_guard_kernel leaves in scal[7] for the host-driven loop
        const PgoDev& P = S.Dp;
        const int n = P.nb * 6;
        double rr = 0., bb = 0.;
        for (int i0 = tid; i0 < n; i0 += 8 * kTailBlk) {        // eight of the lane's entries in flight at a time (10k / 50k: 59 per lane - one
            double rv[8], bv[8];                                // round trip each made the tail 20 us there), added in the same order
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * kTailBlk; rv[u] = i < n ? P.r[i] : 0.; bv[u] = i < n ? P.b[i] : 0.; }
#pragma unroll
            for (int u = 0; u < 8; u++) { rr += rv[u] * rv[u]; bb += bv[u] * bv[u]; }
        }
        for (int o = 32; o; o >>= 1) { rr += __shfl_xor(rr, o); bb += __shfl_xor(bb, o); }
        if ((tid & 63) == 0) { sr[tid >> 6] = rr; sb[tid >> 6] = bb; }
        // finalize_kernel(what = 1): chi2 of the trial + computeScale, sum_partials' order (256 lanes)
        double va = 0., vb = 0.;
        if (tid < kBlk) {
            for (int i = tid; i < S.g_edges; i += kBlk) va += S.D.part_a[i];
            for (int i = tid; i < S.g_oplus; i += kBlk) vb += S.D.part_b[i];
            va = wave_sum(va);
        }
        __syncthreads();
        if (tid < kBlk && (tid & 63) == 0) s4[tid >> 6] = va;
        __syncthreads();
        chi_t = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        if (tid < kBlk) vb = wave_sum(vb);
        __syncthreads();
        if (tid < kBlk && (tid & 63) == 0) s4[tid >> 6] = vb;
        __syncthreads();
        sc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        rr = 0.; bb = 0.;
        for (int w = 0; w < 16; w++) { rr += sr[w]; bb += sb[w]; }
        ratio = bb > 0. ? rr / bb : 0.;
    }
    __shared__ uint32_t s_seq;
    if (tid == 0) {
        const uint32_t seq = (uint32_t)(lm->tails + 1);
        lm->tails = (int32_t)seq;
        s_seq = seq;
        double* __restrict__ scal = sscal;                   // (0, 1 read; 4, 5, 7 written: stored below)
        if (solving && !done) {
            if (lm->flags[1] >= lm->max_it) { lm->phase = kLmAnomaly; lm->anomaly_code = 1; lm->flags[0] = 1; }      // PCG hit its cap
        } else if (solving) {
            const int its = lm->flags[1];
            bool conv = lm->flags[2] == 0;
            if (conv) scal[7] = ratio;
            if (conv && lm->guarded && !(ratio <= kResidualGuard)) conv = false;      // (DESIGN.md "Safeguards": not SPD by construction)
            if (!conv) { lm->phase = kLmAnomaly; lm->anomaly_code = lm->flags[2] ? 2 : 3; }
            else {
                scal[4] = chi_t; scal[5] = sc;
                lm->st_pcg_iterations += its; lm->st_lm_trials++; lm->pcg_last = its;
                const double rate = lm_pcg_rate(scal[1], scal[0], its, lm->tol2, lm->tol_f2);
                if (rate > 0.) { lm->rate_last = rate; if (lm->fresh || lm->rate_ref < 0.) lm->rate_ref = rate; }
                double lambda = lm->lambda, ni = lm->ni;
                const LmStep st = lm_step(lm->chi_cur, chi_t, sc, lambda, ni);
                lm->lambda = lambda; lm->ni = ni;
                if (st.accepted) { lm->last_rel = st.last_rel; lm->chi_cur = chi_t; lm->cur ^= 1; }      // discardTop
                const int qmax = lm->qmax + 1;
                lm->qmax = qmax;
                if (st.rho < 0 && qmax < 10) lm->phase = kLmRetry;                    // another trial on the same linearisation
                else {
                    lm->st_iterations_done = lm->it + 1;
                    if (qmax == 10 || st.rho == 0) { lm->st_terminated_early = 1; lm->phase = kLmDone; }      // Terminate
                    else { lm->it += 1; lm->qmax = 0; lm->phase = (lm->it >= lm->iterations) ? kLmDone : kLmLin; }
                }
            }
        }
    }
    __syncthreads();
    // ---- the state back to its place, and the snapshot the host polls: one word per lane.  The snapshot lives in pinned, coherent
    //      (uncached) host memory: its stores go out as system-scope stores, the lanes wait for their acknowledgement (vmcnt) and the
    //      sequence word follows behind the barrier.  No fence: a system-scope release fence also writes back every dirty line of the L2
    //      - the whole solve's working set, 16 waves each - which was most of this kernel's 11 us, and nothing the host reads is there.
    unsigned long long* __restrict__ dst = reinterpret_cast<unsigned long long*>(S.pub);
    if (tid < kLmWords) {
        const unsigned long long w = reinterpret_cast<const unsigned long long*>(&L)[tid];
        reinterpret_cast<unsigned long long*>(S.lm)[tid] = w;
        __hip_atomic_store(dst + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (tid < kLmWords + 8) {
        const double v = sscal[tid - kLmWords];
        if (tid - kLmWords == 4 || tid - kLmWords == 5 || tid - kLmWords == 7) S.D.scal[tid - kLmWords] = v;
        __hip_atomic_store(reinterpret_cast<double*>(dst) + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (tid == kLmWords + 8) {
        // seq_begin travels with the fields (unordered among them) and seq follows behind them all.  HOST-SIDE INVARIANT this rests on
        // (wait_pub / load_slot, uzl_pgo_lm.hip): the host copies a slot's snapshot BEFORE it enqueues the pass whose tail writes the next
        // one, so no tail can overwrite the fields under its copy; seq_begin == seq around a copy is a cross-check, not the protection.
        // (Round 4 sent seq_begin ahead behind a fence of its own.)
        __hip_atomic_store(&S.pub->seq_begin, s_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    publish_wait_own_stores();                              // this lane's stores have been acknowledged (uzl_common.hpp: gfx9 only without a fence)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&S.pub->seq, s_seq, UZL_PUBLISH_SEQ_ORDER, __HIP_MEMORY_SCOPE_SYSTEM);
}

void k_lm_head(const LmSlot* slots, int nslots, int pass_flags, hipStream_t s)
{
    hipLaunchKernelGGL(lm_head_kernel, dim3(1, 1, nslots), dim3(kBlk), 0, s, slots, pass_flags);
}
void k_lm_tail(const LmSlot* slots, int nslots, hipStream_t s)
{
    hipLaunchKernelGGL(lm_tail_kernel, dim3(1, 1, nslots), dim3(kTailBlk), 0, s, slots);
}

}  // namespace uzl
