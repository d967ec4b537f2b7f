// lm_decide.h - the accept / reject / terminate decision of one Levenberg-Marquardt iteration as ONE function that the host
// loop AND the kernels evaluate (round 4).
//
// What it restates: the tests ceres' TrustRegionMinimizer applies to a candidate step (Ceres 1.8, SURVEY.md section 3.4;
// reached from the reference through ceres::Solve, src/base3d/bundle_adjustment.cc:554-569) and the radius update of its
// Levenberg-Marquardt strategy - the body of mavba_session::iterate of rounds 1-3.
//
// Why a shared function: the host used to read the candidate's scalars, decide, and only then enqueue the evaluation at
// the accepted point - 40-45 us of idle device per iteration (a third of a local-window iteration). Now that evaluation is
// enqueued BEFORE the scalars are read: a one-lane kernel behind the candidate (k_lm_snapshot) runs this function on the
// device, leaves {code, radius} for the kernels of the speculative evaluation - they return at once unless the step is
// accepted, the front end takes the new radius from there - and publishes the scalars to host-mapped memory (no copy
// operation, no event: the host polls a sequence number). The host evaluates the same function on the same scalars, so
// both sides always agree: IEEE +, -, *, /, sqrt and comparisons only, contraction off (the device build fuses a * b + c,
// the host build does not).
// One deliberate deviation from Ceres' text: levenberg_marquardt_strategy.cc writes the radius update as
// pow(2 rho - 1, 3); here the cube is (t * t) * t - two correctly rounded products, which can differ from a (nearly
// correctly rounded) pow in the last bit of the new radius. pow has no bit-exact device counterpart, and host and device
// MUST agree with each other. The CPU restatement under oracle/ keeps Ceres' pow: a last-bit difference of the radius
// scales the LM diagonal by 1 +- 2^-52, far inside the 1e-6 parity bar; the full-solve tests compare iteration counts,
// terminations and final radii (1e-7: the radius follows the gain ratios, which agree to ~1e-9) against it.
#ifndef MAVBA_LM_DECIDE_H_
#define MAVBA_LM_DECIDE_H_
#include "internal.h"
#include "ba_math.h"

namespace mavba {

MAVBA_HD bool lm_finite(double x) { return (x - x) == 0.0; }

MAVBA_HD LmDecision lm_decide(const double* sc, const LmSpec& P) {
#pragma clang fp contract(off)
  LmDecision d;
  d.code = LM_REJECTED; d.radius = P.radius; d.decrease_factor = P.decrease_factor; d.rel = 0.0; d.step_norm = 0.0; d.cost_change = 0.0;
  // the evaluation's own test comes first (ceres tests the gradient right after evaluating an accepted point)
  if (P.pending_eval && sc[SC_GRAD_MAX] <= P.abs_gtol) { d.code = LM_TERM_GTOL; return d; }
  const double cost = sc[SC_COST], x_norm = sqrt(sc[SC_XNORM2]);
  const double mcc = sc[SC_MODEL_CHANGE];
  const bool solved = sc[SC_FAIL] == 0.0 && sc[SC_FAIL_FRONT] == 0.0 && lm_finite(mcc) && lm_finite(sc[SC_STEP_NORM2]);
  const bool valid = solved && !(mcc < 0.0);
  if (!valid) {
    d.code = LM_INVALID;
  } else {
    d.step_norm = sqrt(sc[SC_STEP_NORM2]);
    if (d.step_norm <= P.ptol * (x_norm + P.ptol)) { d.code = LM_TERM_PTOL; return d; }
    d.cost_change = cost - sc[SC_NEW_COST];
    if (fabs(d.cost_change) < P.ftol * cost) { d.code = LM_TERM_FTOL; return d; }
    d.rel = d.cost_change / mcc;
    if (d.rel > P.min_rel_dec) d.code = LM_ACCEPTED;
  }
  if (d.code == LM_ACCEPTED) {
    const double t = 2.0 * d.rel - 1.0;
    const double t3 = (t * t) * t;
    const double q = 1.0 - t3;
    const double third = 1.0 / 3.0;
    double r = P.radius / (q > third ? q : third);
    if (r > P.max_radius) r = P.max_radius;
    d.radius = r;
    d.decrease_factor = 2.0;
  } else {  // rejected or invalid
    d.radius = P.radius / P.decrease_factor;
    d.decrease_factor = P.decrease_factor * 2.0;
  }
  return d;
}

#if defined(__HIPCC__)
// Top of a kernel of the speculative evaluation: false = the step was not accepted, nothing of this launch may run.
// radius (may be null) receives the trust-region radius the accepted step leaves.
__device__ __forceinline__ bool lm_spec_go(const LmSpec& s, double* radius) {
  if (s.dec == nullptr) return true;
  if (s.dec[0] != (double)LM_ACCEPTED) return false;
  if (radius) *radius = s.dec[1];
  return true;
}
#endif

}  // namespace mavba
#endif
