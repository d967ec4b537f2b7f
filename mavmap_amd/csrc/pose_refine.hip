// pose_refine.hip - batched single-camera pose refinement: ONE work-group runs the whole Levenberg-Marquardt
// loop of one problem on the device, a launch carries any number of problems.
//
// Replaces pose_refinement() (reference src/base3d/bundle_adjustment.cc:139-225: one image, six pose parameters,
// points and intrinsics constant, Cauchy loss, ceres LM with DENSE_QR), which MAVMAP calls once per processed image
// pair on the RANSAC inlier set (src/sfm/sequential_mapper.cc:709-720). The general session pays ~1 ms of launch
// latencies and host round trips per call for a 6 x 6 system; here the trust-region loop itself - evaluation,
// normal equations, 6 x 6 solve, candidate cost, accept / reject, radius update, termination tests in ceres'
// order - lives in the kernel, so a call is one upload, one launch, one download, and N inlier sets (RANSAC
// hypotheses, or the pairs of several images) cost the same launch.
//
// LM semantics are those of session_lm.hip (Ceres 1.8 TrustRegionMinimizer + LevenbergMarquardtStrategy,
// SURVEY.md 3.4): Jacobi scaling 1 / (1 + |J_j|) from the first Jacobian, D^2 = clamp(diag) / radius, model cost
// change 1/2 y (g + D^2 y), step quality, radius update 1 / max(1/3, 1 - (2 rho - 1)^3). Every thread carries the
// loop's scalars redundantly (they are computed from block-reduced sums every thread receives), so the control
// flow is uniform and needs no broadcast. Reductions have a fixed shape: results are bit-reproducible.
#include "session.h"
#include "dev_reduce.h"

namespace mavba {

namespace {
constexpr int kPoseSums = 28;  // H (21, upper triangle of J^T J) | g (6) | cost

struct PoseEval { double H[21], g[6], cost; };

// Sum the per-thread partials over the block; every thread gets the totals. scratch: [4][kPoseSums].
__device__ __forceinline__ void pose_block_sum(double* v, int n, double* scratch, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  __syncthreads();  // scratch may still be read from the previous reduction
  for (int k = 0; k < n; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) scratch[wv * kPoseSums + k] = s;
  }
  __syncthreads();
  for (int k = 0; k < n; ++k)
    v[k] = (scratch[k] + scratch[kPoseSums + k]) + (scratch[2 * kPoseSums + k] + scratch[3 * kPoseSums + k]);
}

// Residuals + Jacobian w.r.t. the pose at x: H = J^T J, g = J^T r (loss-corrected rows), cost = 1/2 sum rho.
__device__ __forceinline__ void pose_evaluate(const double* x, int model, const double* kin, const double2* uv, const double* xyz,
                                              long long n, double loss_b, double loss_inv_b, double* scratch, int tid, PoseEval& E) {
  double rec[9];
  cam_prepare(x, rec);
  double acc[kPoseSums];
#pragma unroll
  for (int k = 0; k < kPoseSums; ++k) acc[k] = 0.0;
  for (long long o = tid; o < n; o += 256) {
    const double2 m = uv[o];
    const double X[3] = {xyz[3 * o], xyz[3 * o + 1], xyz[3 * o + 2]};
    double r[2], Jc[12], Jp[6], Jk[18];
    obs_jacobian(model, rec, kin, X, m.x, m.y, r, Jc, Jp, Jk);
    double w, half_rho;
    cauchy_weight(r[0] * r[0] + r[1] * r[1], loss_b, loss_inv_b, w, half_rho);
    const double w2 = w * w;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = a; b < 6; ++b) acc[sym_idx(a, b, 6)] += w2 * (Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
      acc[21 + a] += w2 * (Jc[a] * r[0] + Jc[6 + a] * r[1]);
    }
    acc[27] += half_rho;
  }
  pose_block_sum(acc, kPoseSums, scratch, tid);
#pragma unroll
  for (int k = 0; k < 21; ++k) E.H[k] = acc[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) E.g[k] = acc[21 + k];
  E.cost = acc[27];
}

__device__ __forceinline__ double pose_cost(const double* x, int model, const double* kin, const double2* uv, const double* xyz,
                                            long long n, double loss_b, double loss_inv_b, double* scratch, int tid) {
  double rec[9];
  cam_prepare(x, rec);
  double c[1] = {0.0};
  for (long long o = tid; o < n; o += 256) {
    const double2 m = uv[o];
    const double X[3] = {xyz[3 * o], xyz[3 * o + 1], xyz[3 * o + 2]};
    double r[2], w, half_rho;
    obs_residual(model, rec, kin, X, m.x, m.y, r);
    cauchy_weight(r[0] * r[0] + r[1] * r[1], loss_b, loss_inv_b, w, half_rho);
    c[0] += half_rho;
  }
  pose_block_sum(c, 1, scratch, tid);
  return c[0];
}

// Solve the SPD 6 x 6 system A y = b (A full, row-major) by Cholesky; false if a pivot is not positive / finite.
__device__ __forceinline__ bool solve6(const double* A, const double* b, double* y) {
  double L[36];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
    ok = ok && (d > 0.0) && isfinite(d);
    const double inv = 1.0 / sqrt(d);
    L[j * 6 + j] = d * inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = s * inv;
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * z[k];
    z[i] = s / L[i * 6 + i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * y[k];
    y[i] = s / L[i * 6 + i];
  }
  return ok;
}
}  // namespace

struct PoseItemDev { long long begin, end; int model, pad; };

__global__ void __launch_bounds__(256) k_pose_refine_batch(const PoseItemDev* __restrict__ items, double* __restrict__ poses,
                                                           const double* __restrict__ intr, const double2* __restrict__ uv,
                                                           const double* __restrict__ xyz, mavba_options opt,
                                                           mavba_result* __restrict__ results) {
  __shared__ double scratch[4 * kPoseSums];
  const int tid = threadIdx.x;
  const PoseItemDev it = items[blockIdx.x];
  const long long n = it.end - it.begin;
  const double2* u = uv + it.begin;
  const double* X = xyz + 3 * it.begin;
  double kin[9], x[6];
#pragma unroll
  for (int k = 0; k < 9; ++k) kin[k] = intr[9 * (size_t)blockIdx.x + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = poses[6 * (size_t)blockIdx.x + k];
  const double loss_b = opt.loss_scale_factor * opt.loss_scale_factor, loss_inv_b = 1.0 / loss_b;
  const double dmin = opt.min_lm_diagonal, dmax = opt.max_lm_diagonal;

  PoseEval E;
  pose_evaluate(x, it.model, kin, u, X, n, loss_b, loss_inv_b, scratch, tid, E);
  double scale[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) scale[e] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(E.H[sym_idx(e, e, 6)])) : 1.0;
  auto grad_max_of = [&](const PoseEval& V) { double m = 0.0; for (int e = 0; e < 6; ++e) m = fmax(m, fabs(V.g[e])); return m; };
  auto norm_of = [&](const double* v) { double s = 0.0; for (int e = 0; e < 6; ++e) s += v[e] * v[e]; return sqrt(s); };
  double cost = E.cost, grad_max = grad_max_of(E), x_norm = norm_of(x);
  const double initial_cost = cost;
  const double abs_gtol = opt.gradient_tolerance * fmax(grad_max, 2.220446049250313e-16);
  double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
  int iteration = 0, invalid_steps = 0, n_success = 0, n_fail = 0;
  int termination = MAVBA_TERM_RUNNING;
  if (n == 0) termination = MAVBA_TERM_FUNCTION_TOLERANCE;  // no residual block: nothing to do (session start())
  else if (grad_max <= abs_gtol) termination = MAVBA_TERM_GRADIENT_TOLERANCE;

  while (termination == MAVBA_TERM_RUNNING) {
    if (iteration >= opt.max_num_iterations) { termination = MAVBA_TERM_NO_CONVERGENCE; break; }
    ++iteration;
    // (J_s^T J_s + D^2) y = J_s^T r, J_s = J diag(scale)
    double A[36], b[6], D2[6], y[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c < 6; ++c) A[r * 6 + c] = scale[r] * scale[c] * E.H[sym_idx(r < c ? r : c, r < c ? c : r, 6)];
      D2[r] = clampd(A[r * 6 + r], dmin, dmax) / radius;
      A[r * 6 + r] += D2[r];
      b[r] = scale[r] * E.g[r];
    }
    bool solved = solve6(A, b, y);
    double cand[6], step2 = 0.0, mcc = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const double d = -y[e] * scale[e];
      cand[e] = x[e] + d;
      step2 += d * d;
      mcc += 0.5 * y[e] * (b[e] + D2[e] * y[e]);
    }
    solved = solved && isfinite(mcc) && isfinite(step2);
    const bool valid = solved && !(mcc < 0.0);
    bool successful = false;
    double rel = 0.0;
    if (!valid) {
      if (++invalid_steps >= opt.max_num_consecutive_invalid_steps) { termination = MAVBA_TERM_NUMERICAL_FAILURE; ++n_fail; break; }
    } else {
      invalid_steps = 0;
      const double new_cost = pose_cost(cand, it.model, kin, u, X, n, loss_b, loss_inv_b, scratch, tid);
      const double step_norm = sqrt(step2);
      if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { termination = MAVBA_TERM_PARAMETER_TOLERANCE; break; }
      const double cost_change = cost - new_cost;
      if (fabs(cost_change) < opt.function_tolerance * cost) { termination = MAVBA_TERM_FUNCTION_TOLERANCE; break; }
      rel = cost_change / mcc;
      successful = rel > opt.min_relative_decrease;
    }
    if (successful) {
      ++n_success;
      const double t = 2.0 * rel - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
      radius = fmin(opt.max_trust_region_radius, radius);
      decrease_factor = 2.0;
#pragma unroll
      for (int e = 0; e < 6; ++e) x[e] = cand[e];
      pose_evaluate(x, it.model, kin, u, X, n, loss_b, loss_inv_b, scratch, tid, E);
      cost = E.cost; grad_max = grad_max_of(E); x_norm = norm_of(x);
      if (grad_max <= abs_gtol) termination = MAVBA_TERM_GRADIENT_TOLERANCE;
    } else {
      ++n_fail;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
    if (termination == MAVBA_TERM_RUNNING && radius < opt.min_trust_region_radius) termination = MAVBA_TERM_PARAMETER_TOLERANCE;
  }
  if (tid == 0) {
    // ceres leaves the user's parameter blocks untouched after NUMERICAL_FAILURE
    if (termination != MAVBA_TERM_NUMERICAL_FAILURE)
      for (int e = 0; e < 6; ++e) poses[6 * (size_t)blockIdx.x + e] = x[e];
    mavba_result R;
    R.initial_cost = initial_cost; R.final_cost = cost; R.fixed_cost = 0.0;
    R.num_residuals = 2 * n; R.num_residuals_reduced = 2 * n; R.num_parameters_reduced = n > 0 ? 6 : 0;
    R.num_successful_steps = n_success; R.num_unsuccessful_steps = n_fail; R.termination = termination;
    R.final_gradient_max_norm = grad_max; R.final_trust_region_radius = radius;
    R.setup_seconds = 0.0; R.solve_seconds = 0.0;
    results[blockIdx.x] = R;
  }
}

// Host side: pack the inliers of every item, one upload, one launch, one download.
void pose_refine_batch(int count, mavba_pose_refine_item* items, const mavba_options& opt, mavba_result* results) {
  const double t0 = now_s();
  std::vector<PoseItemDev> hd((size_t)count);
  std::vector<double> poses((size_t)count * 6), intr((size_t)count * 9, 0.0);
  long long total = 0;
  for (int q = 0; q < count; ++q) {
    const mavba_pose_refine_item& it = items[q];
    if (it.n < 0 || (it.n > 0 && (!it.uv || !it.xyz)) || !it.intrinsics) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument in pose-refinement item");
    if (it.camera_model < 1 || it.camera_model > 3) throw Failure(MAVBA_ERR_BAD_MODEL, "bad camera model");
    long long m = 0;
    if (it.inlier_mask) { for (long long i = 0; i < it.n; ++i) m += it.inlier_mask[i] != 0; } else m = it.n;
    hd[q].begin = total; hd[q].end = total + m; hd[q].model = it.camera_model; hd[q].pad = 0;
    total += m;
    for (int k = 0; k < 3; ++k) { poses[(size_t)q * 6 + k] = it.rvec[k]; poses[(size_t)q * 6 + 3 + k] = it.tvec[k]; }
    for (int k = 0; k < model_k(it.camera_model); ++k) intr[(size_t)q * 9 + k] = it.intrinsics[k];
  }
  std::vector<double2> uv((size_t)std::max<long long>(total, 1));
  std::vector<double> xyz((size_t)std::max<long long>(total, 1) * 3);
  for (int q = 0; q < count; ++q) {
    const mavba_pose_refine_item& it = items[q];
    long long at = hd[q].begin;
    for (long long i = 0; i < it.n; ++i) {
      if (it.inlier_mask && !it.inlier_mask[i]) continue;
      uv[(size_t)at] = make_double2(it.uv[2 * i], it.uv[2 * i + 1]);
      xyz[(size_t)at * 3] = it.xyz[3 * i]; xyz[(size_t)at * 3 + 1] = it.xyz[3 * i + 1]; xyz[(size_t)at * 3 + 2] = it.xyz[3 * i + 2];
      ++at;
    }
  }
  hipStream_t st = nullptr;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(stream_acquire(&st));
  struct Release { hipStream_t st; int dev; ~Release() { (void)hipStreamSynchronize(st); release_staged(st); stream_release(st, dev); } } rel{st, dev};
  DevBuf<PoseItemDev> d_items; DevBuf<double> d_poses, d_intr, d_xyz; DevBuf<double2> d_uv; DevBuf<mavba_result> d_res;
  d_items.upload(hd, st); d_poses.upload(poses, st); d_intr.upload(intr, st); d_uv.upload(uv, st); d_xyz.upload(xyz, st);
  d_res.alloc((size_t)count);
  const double t1 = now_s();
  hipLaunchKernelGGL(k_pose_refine_batch, dim3(count), dim3(256), 0, st, d_items.p, d_poses.p, d_intr.p, d_uv.p, d_xyz.p, opt, d_res.p);
  std::vector<mavba_result> hres((size_t)count);
  HIP_OK(copy_d2h_staged_sync(poses.data(), d_poses.p, poses.size() * 8, st));
  HIP_OK(copy_d2h_staged_sync(hres.data(), d_res.p, hres.size() * sizeof(mavba_result), st));
  release_staged(st);
  HIP_OK(hipGetLastError());
  const double t2 = now_s();
  for (int q = 0; q < count; ++q) {
    for (int k = 0; k < 3; ++k) { items[q].rvec[k] = poses[(size_t)q * 6 + k]; items[q].tvec[k] = poses[(size_t)q * 6 + 3 + k]; }
    if (results) {
      results[q] = hres[q];
      results[q].setup_seconds = (t1 - t0) / count;
      results[q].solve_seconds = (t2 - t1) / count;
    }
  }
}

}  // namespace mavba
