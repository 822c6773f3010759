// pgo_lm.hpp — the scalar decisions of the Levenberg-Marquardt loop, written once for the host-driven loop (uzl_pgo.hip, do_optimize)
// and the device-resident one (pgo_lm_kernels.hip): g2o's OptimizationAlgorithmLevenberg::solve [EXT], which the reference runs through
// optimizer_.optimize(iterations) (graph_optimization/src/g2o_optimizer.cpp:148).  Both loops must take the SAME decisions from the same
// numbers - tests hold their poses to array_equal - so nothing here may depend on who compiles it: no contraction of a * b + c, no
// library pow / log whose last bit differs between libm and the device library.
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

#include "pgo_types.hpp"

namespace uzl {

#define UZL_HD __host__ __device__ __forceinline__

// |r|^2 / |b|^2 a solve under the multiplicative operator must reach (DESIGN.md "Safeguards"); the stop test lets a solve go at 0.2
constexpr double kResidualGuard = 0.25;
// a kept preconditioner is rebuilt when its contraction per PCG iteration has fallen below this share of what it delivered when fresh
constexpr double kRateDrop = 0.6;
// factor on pcg_tol^2 of the relative floor under the step-error stop test (scal[8])
constexpr double kTolFloor2 = 1e-4;
// step accuracy asked of a solve, per unit of cfg.pcg_tol: metres, and quaternion-vector units (~ half radians)
constexpr double kStepT = 1.0, kStepR = 0.1;

// t^3 rounded once (g2o: pow(2 rho - 1, 3)): the product is carried exactly in two doubles and summed at the end.  Correctly rounded
// up to a second-order term far below half an ulp; the same bits from x86 and gfx950, which libm's and the device library's pow are not.
UZL_HD double lm_cube(double t)
{
#pragma clang fp contract(off)
    const double p = t * t, pe = fma(t, t, -p);          // t^2 = p + pe
    const double q = p * t, qe = fma(p, t, -q);          // p t = q + qe
    return q + (qe + pe * t);
}

// ln(x) for x > 0 from exponent + atanh series: plain arithmetic, identical on both sides.  Only ratios of such logs are compared
// (pcg_rate), so 1e-15 relative accuracy is ample.
UZL_HD double lm_log(double x)
{
#pragma clang fp contract(off)
    if (!(x < 1e300)) return 690.77552789821368;         // (ln 1e300: an overflowed ratio, or not a number; nothing compares that finely)
    if (!(x > 1e-300)) return -690.77552789821368;
    int e = 0;
    double m = x;
    // frexp by hand: scale into [sqrt(1/2), sqrt(2))
    while (m >= 1.4142135623730951) { m *= 0.5; e++; }
    while (m < 0.70710678118654757) { m *= 2.0; e--; }
    const double s = (m - 1.) / (m + 1.), s2 = s * s;
    double term = s, sum = 0.;
    for (int k = 1; k < 60; k += 2) {
        sum += term / k;
        term *= s2;
        if (term < 1e-20 && term > -1e-20) break;
    }
    return 2. * sum + e * 0.69314718055994529;
}

// How stale is a kept preconditioner?  The CONTRACTION it delivers, nats of r.M^-1 r per PCG iteration (rz_stop = scal[1] is
// pcg_tol^2 * tol_f2 * (r_0.M^-1 r_0)); -1 = no estimate (too few iterations).
UZL_HD double lm_pcg_rate(double rz_stop, double rz_end, int its, double tol2, double tol_f2)
{
#pragma clang fp contract(off)
    const double rz0 = rz_stop / (tol2 * tol_f2);
    return (its >= 16 && rz0 > 0. && rz_end > 0. && rz_end < rz0) ? lm_log(rz0 / rz_end) / its : -1.;
}

// does this linearisation rebuild the multilevel preconditioner?  (lazy refresh)
UZL_HD bool lm_refresh(int it, int iterations, bool always_refresh, bool may_run_last, double last_rel, double refresh_rel, double rate_ref, double rate_last,
                       double rate_drop = kRateDrop)
{
#pragma clang fp contract(off)
    return it == 0 || ((always_refresh || last_rel > refresh_rel || (rate_ref > 0. && rate_last > 0. && rate_last < rate_drop * rate_ref)) &&
                       (may_run_last || it + 1 < iterations));
}

// the step control of one evaluated trial: rho from chi2 before / after and computeScale() + 1e-3; updates lambda, ni; returns rho
struct LmStep { double rho; bool accepted; double last_rel; };
UZL_HD LmStep lm_step(double current_chi, double temp_chi, double scale_sum, double& lambda, double& ni)
{
#pragma clang fp contract(off)
    LmStep r;
    const double scale = scale_sum + 1e-3;                                       // computeScale + 1e-3
    r.rho = (current_chi - temp_chi) / scale;
    r.accepted = r.rho > 0 && std::isfinite(temp_chi);
    r.last_rel = 0.;
    if (r.accepted) {                                                            // good step
        double alpha = 1. - lm_cube(2 * r.rho - 1);
        alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
        const double scaleFactor = (1. / 3. > alpha) ? 1. / 3. : alpha;
        lambda *= scaleFactor;
        ni = 2.;
        const double at = temp_chi < 0 ? -temp_chi : temp_chi, d = current_chi - temp_chi;
        r.last_rel = (d < 0 ? -d : d) / (at > 1e-300 ? at : 1e-300);
    } else {
        lambda *= ni;
        ni *= 2.;
    }
    return r;
}

}  // namespace uzl
