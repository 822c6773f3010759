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

__global__ __launch_bounds__(kBlk) void lm_head_kernel(const LmSlot* __restrict__ slots, int pass_flags)
{
    __shared__ double s4[4];
    const LmSlot& S = slots[blockIdx.z];
    LmDev* lm = S.lm;
    const int phase = lm->phase;
    double chi = 0., dmax = 0.;
    if (phase == kLmLin) {                                   // (uniform) finalize_kernel(what = 2): chi2 + max diagonal
        chi = sum_partials(S.D.part_a, S.g_edges, s4);
        double v = 0.;
        for (int i = threadIdx.x; i < S.g_asm; i += kBlk) v = fmax(v, S.D.part_c[i]);
        dmax = block_max(v, s4);
    }
    if (threadIdx.x != 0) return;
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
        if (lm_refresh(lm->it, lm->iterations, lm->always_refresh != 0, lm->sync_rebuild != 0, lm->last_rel, lm->refresh_rel, lm->rate_ref, lm->rate_last)) {
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

__device__ __forceinline__ void lm_publish(const LmSlot& S, const LmDev* lm, uint32_t seq)
{
    LmHost* __restrict__ out = S.pub;
    __hip_atomic_store(&out->seq_begin, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    out->phase = lm->phase; out->cur = lm->cur; out->ix = lm->ix; out->need = lm->need;
    out->it = lm->it; out->qmax = lm->qmax; out->pending = lm->pending; out->pcg_last = lm->pcg_last;
    for (int i = 0; i < 4; i++) out->flags[i] = lm->flags[i];
    out->st_pcg_iterations = lm->st_pcg_iterations; out->st_lm_trials = lm->st_lm_trials; out->st_precond_builds = lm->st_precond_builds;
    out->st_iterations_done = lm->st_iterations_done; out->st_terminated_early = lm->st_terminated_early; out->anomaly_code = lm->anomaly_code;
    out->lambda = lm->lambda; out->chi_cur = lm->chi_cur; out->last_rel = lm->last_rel; out->rate_ref = lm->rate_ref; out->rate_last = lm->rate_last;
    out->chi2_initial = lm->chi2_initial; out->lambda_setup[0] = lm->lambda_setup[0]; out->lambda_setup[1] = lm->lambda_setup[1];
    for (int i = 0; i < 8; i++) out->scal[i] = S.D.scal[i];
    __threadfence_system();
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlk) void lm_tail_kernel(const LmSlot* __restrict__ slots)
{
    __shared__ double s4[4];
    const LmSlot& S = slots[blockIdx.z];
    LmDev* lm = S.lm;
    const bool solving = lm->phase == kLmSolve;
    const bool done = lm->flags[0] != 0;
    const bool eval = solving && done && lm->flags[2] == 0;      // (uniform) the evaluation kernels of this pass ran for this graph
    double chi_t = 0., sc = 0.;
    if (eval) {                                              // finalize_kernel(what = 1): chi2 of the trial + computeScale
        chi_t = sum_partials(S.D.part_a, S.g_edges, s4);
        sc = sum_partials(S.D.part_b, S.g_oplus, s4);
    }
    if (threadIdx.x != 0) return;
    const uint32_t seq = (uint32_t)(lm->tails + 1);
    lm->tails = (int32_t)seq;
    double* __restrict__ scal = S.D.scal;
    if (solving && !done) {
        if (lm->flags[1] >= lm->max_it) { lm->phase = kLmAnomaly; lm->anomaly_code = 1; lm->flags[0] = 1; }      // PCG hit its cap
    } else if (solving) {
        const int its = lm->flags[1];
        bool conv = lm->flags[2] == 0;
        if (conv && lm->guarded && !(scal[7] <= kResidualGuard)) conv = false;      // (DESIGN.md "Safeguards": not SPD by construction)
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
    lm_publish(S, lm, seq);
}

void k_lm_head(const LmSlot* slots, int nslots, int pass_flags, hipStream_t s)
{
    hipLaunchKernelGGL(lm_head_kernel, dim3(1, 1, nslots), dim3(kBlk), 0, s, slots, pass_flags);
}
void k_lm_tail(const LmSlot* slots, int nslots, hipStream_t s)
{
    hipLaunchKernelGGL(lm_tail_kernel, dim3(1, 1, nslots), dim3(kBlk), 0, s, slots);
}

}  // namespace uzl
