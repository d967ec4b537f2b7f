// ceres_check.cc - the reference's OWN solver on a replay file.   TEST INFRASTRUCTURE (oracle/): never linked into,
// loaded by or executed from the product; only tests/ and bench.py's cpu_baseline leg run it, as the checker.
//
//   mavba_ceres_check <problem.bin> <result.bin> [num_threads]
//
// Builds the ceres::Problem the way reference src/base3d/bundle_adjustment.cc:473-551 does from the flat problem
// of include/mavba.h (the shim's replay file, shim/base3d/bundle_adjustment.cc dump_problem / BAProblem.save):
//   * one AutoDiffCostFunction<.., 2, 3, 1, 1, 1, 3, K> per observation on the blocks rvec(3), tx, ty, tz, point(3),
//     intrinsics(K) (bundle_adjustment.h:124-159), model by camera code, shared CauchyLoss (:477-478), in the file's
//     observation order (= the reference's residual-block order, :511-533);
//   * constant blocks from the per-image mask / intr_const / point_const (:361-385, :545-549);
//   * one rotation-prior residual per listed image, NULL loss, with the reference's index pattern (:72-111);
//   * Solver::Options of :553-566; point3D errors by Problem::Evaluate with apply_loss_function = false (:575-598).
// Camera models are restated from src/base3d/camera_models.h:111-130, 170-193, 225-242, 277-302 as templates on T.
// Output (little endian): "MAVBAR1\0", double initial_cost, final_cost, int32 successful, unsuccessful, termination
// (ceres enum value), num_residuals, double solve_seconds, then poses[NI*6], intrinsics[NC*9], points[NP*3],
// point_error[NP] (NaN = not in the problem).
//
// This file cannot be compiled in the build image (no Ceres, no Eigen); it is written against the public Ceres 1.8+
// API and is exercised wherever CMake finds Ceres (oracle/ceres_check/CMakeLists.txt).
#include <ceres/ceres.h>
#include <ceres/rotation.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#ifdef MAVBA_HAVE_OPENMP
#include <omp.h>
#endif

namespace {

struct Pinhole {
  static const int K = 4;
  template <typename T>
  static void project(const T& x, const T& y, const T& z, const T* k, T& u, T& v) {
    u = k[0] * (x / z) + k[2];
    v = k[1] * (y / z) + k[3];
  }
};
template <typename T>
void brown(const T& un, const T& vn, const T* k, T& du, T& dv) {
  const T u2 = un * un, v2 = vn * vn, uv = un * vn, r2 = u2 + v2;
  const T radial = k[4] * r2 + k[5] * r2 * r2;
  du = un * radial + T(2) * k[6] * uv + k[7] * (r2 + T(2) * u2);
  dv = vn * radial + T(2) * k[7] * uv + k[6] * (r2 + T(2) * v2);
}
struct OpenCV {
  static const int K = 8;
  template <typename T>
  static void project(const T& x, const T& y, const T& z, const T* k, T& u, T& v) {
    const T un = x / z, vn = y / z;
    T du, dv;
    brown(un, vn, k, du, dv);
    u = k[0] * (un + du) + k[2];
    v = k[1] * (vn + dv) + k[3];
  }
};
struct Cata {
  static const int K = 9;
  template <typename T>
  static void project(const T& x, const T& y, const T& z, const T* k, T& u, T& v) {
    const T zz = z + k[8] * ceres::sqrt(x * x + y * y + z * z);
    const T un = x / zz, vn = y / zz;
    T du, dv;
    brown(un, vn, k, du, dv);
    u = k[0] * (un + du) + k[2];
    v = k[1] * (vn + dv) + k[3];
  }
};

template <typename Model>
struct Reprojection {
  Reprojection(double u, double v) : u_(u), v_(v) {}
  template <typename T>
  bool operator()(const T* const rvec, const T* const tx, const T* const ty, const T* const tz, const T* const X,
                  const T* const k, T* r) const {
    T Xc[3];
    ceres::AngleAxisRotatePoint(rvec, X, Xc);
    Xc[0] += tx[0]; Xc[1] += ty[0]; Xc[2] += tz[0];
    T u, v;
    Model::project(Xc[0], Xc[1], Xc[2], k, u, v);
    r[0] = u - T(u_);
    r[1] = v - T(v_);
    return true;
  }
  static ceres::CostFunction* create(double u, double v) {
    return new ceres::AutoDiffCostFunction<Reprojection<Model>, 2, 3, 1, 1, 1, 3, Model::K>(new Reprojection<Model>(u, v));
  }
  double u_, v_;
};

// w * sqrt(sum (R(rvec)^T - R0)^2) with the reference's element pairing, whose eighth term reads rotmat[6] where the
// transpose pattern would read rotmat[5] (bundle_adjustment.cc:88-105) - kept, it is part of what Ceres minimises there.
struct RotationPrior {
  RotationPrior(double w, const double* rvec0) : w_(w) { ceres::AngleAxisToRotationMatrix(rvec0, R0_); }
  template <typename T>
  bool operator()(const T* const rvec, T* r) const {
    T R[9];
    ceres::AngleAxisToRotationMatrix(rvec, R);  // column-major
    static const int ia[9] = {0, 3, 6, 1, 4, 7, 2, 6, 8};
    T s = T(0);
    for (int q = 0; q < 9; ++q) { const T d = R[ia[q]] - T(R0_[q]); s += d * d; }
    r[0] = T(w_) * ceres::sqrt(s);
    return true;
  }
  double w_, R0_[9];
};

template <typename T>
bool rd(std::FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin result.bin [threads]\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  char magic[8];
  int32_t dims[4];
  int64_t NO;
  double prior_weight, opts[4];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "MAVBA1", 6) != 0 || std::fread(dims, 4, 4, f) != 4 ||
      std::fread(&NO, 8, 1, f) != 1 || std::fread(&prior_weight, 8, 1, f) != 1 || std::fread(opts, 8, 4, f) != 4) {
    std::fprintf(stderr, "not a MAVBA1 replay file\n"); return 2;
  }
  const size_t NI = dims[0], NC = dims[1], NP = dims[2], NR = dims[3];
  std::vector<double> poses, intr, points, uv, prior_rvec;
  std::vector<uint8_t> pose_const, intr_const, point_const;
  std::vector<int32_t> image_camera, camera_model, obs_image, obs_point, prior_image;
  bool ok = rd(f, poses, NI * 6) && rd(f, pose_const, NI) && rd(f, image_camera, NI) && rd(f, intr, NC * 9) &&
            rd(f, camera_model, NC) && rd(f, intr_const, NC) && rd(f, points, NP * 3) && rd(f, point_const, NP) &&
            rd(f, uv, (size_t)NO * 2) && rd(f, obs_image, (size_t)NO) && rd(f, obs_point, (size_t)NO) &&
            rd(f, prior_image, NR) && rd(f, prior_rvec, NR * 3);
  std::fclose(f);
  if (!ok) { std::fprintf(stderr, "truncated replay file\n"); return 2; }

  ceres::Problem problem;
  ceres::LossFunction* loss = new ceres::CauchyLoss(opts[3]);
  std::vector<int64_t> first_block_of_point(NP, -1);
  std::vector<int> point_of_block;
  std::vector<size_t> residuals_of_image(NI, 0);
  std::vector<char> point_in_problem(NP, 0);
  for (int64_t o = 0; o < NO; ++o) {
    const int i = obs_image[o], p = obs_point[o], c = image_camera[i];
    ceres::CostFunction* cost = nullptr;
    switch (camera_model[c]) {
      case 1: cost = Reprojection<Pinhole>::create(uv[2 * o], uv[2 * o + 1]); break;
      case 2: cost = Reprojection<OpenCV>::create(uv[2 * o], uv[2 * o + 1]); break;
      case 3: cost = Reprojection<Cata>::create(uv[2 * o], uv[2 * o + 1]); break;
      default: std::fprintf(stderr, "bad camera model\n"); return 2;
    }
    double* ps = &poses[(size_t)i * 6];
    problem.AddResidualBlock(cost, loss, ps, ps + 3, ps + 4, ps + 5, &points[(size_t)p * 3], &intr[(size_t)c * 9]);
    point_of_block.push_back(p);
    point_in_problem[p] = 1;
    residuals_of_image[i]++;
  }
  // constancy (the flat problem already encodes the reference's "> 1 residual" rule in the masks)
  for (size_t i = 0; i < NI; ++i) {
    if (residuals_of_image[i] == 0) continue;
    double* ps = &poses[i * 6];
    if (pose_const[i] & 1u) problem.SetParameterBlockConstant(ps);
    if (pose_const[i] & 2u) problem.SetParameterBlockConstant(ps + 3);
    if (pose_const[i] & 4u) problem.SetParameterBlockConstant(ps + 4);
    if (pose_const[i] & 8u) problem.SetParameterBlockConstant(ps + 5);
    if (intr_const[image_camera[i]]) problem.SetParameterBlockConstant(&intr[(size_t)image_camera[i] * 9]);
  }
  for (size_t p = 0; p < NP; ++p)
    if (point_const[p] && point_in_problem[p]) problem.SetParameterBlockConstant(&points[p * 3]);
  for (size_t q = 0; q < NR; ++q) {
    ceres::CostFunction* cost = new ceres::AutoDiffCostFunction<RotationPrior, 1, 3>(new RotationPrior(prior_weight, &prior_rvec[q * 3]));
    problem.AddResidualBlock(cost, nullptr, &poses[(size_t)prior_image[q] * 6]);
    point_of_block.push_back(-1);
  }

  ceres::Solver::Options so;  // bundle_adjustment.cc:553-566
  so.linear_solver_type = ceres::SPARSE_SCHUR;
  so.max_num_iterations = (int)opts[0];
  so.function_tolerance = opts[1];
  so.gradient_tolerance = opts[2];
  so.max_num_consecutive_invalid_steps = 10;
  so.max_consecutive_nonmonotonic_steps = 10;
  so.minimizer_progress_to_stdout = false;
  int threads = argc > 3 ? std::atoi(argv[3]) : 0;
#ifdef MAVBA_HAVE_OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#endif
  if (threads > 0) so.num_threads = threads;
  ceres::Solver::Summary summary;
  const auto t0 = std::chrono::steady_clock::now();
  ceres::Solve(so, &problem, &summary);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%s\n", summary.BriefReport().c_str());
  std::printf("threads %d, solve %.3f s, %.3f iterations/s\n", threads, secs,
              (summary.num_successful_steps + summary.num_unsuccessful_steps) / secs);

  // point3D errors: un-robustified residual norms, mean per point (bundle_adjustment.cc:575-598)
  std::vector<double> perr(NP, std::numeric_limits<double>::quiet_NaN());
  {
    ceres::Problem::EvaluateOptions eo;
    eo.apply_loss_function = false;
    std::vector<double> res;
    problem.Evaluate(eo, nullptr, &res, nullptr, nullptr);
    std::vector<int> count(NP, 0);
    for (int64_t o = 0; o < NO; ++o) count[obs_point[o]]++;
    for (size_t p = 0; p < NP; ++p) if (count[p] > 0) perr[p] = 0.0;
    for (int64_t o = 0; o < NO; ++o) {  // residual blocks are in insertion order; the priors follow the observations
      const int p = obs_point[o];
      perr[p] += std::sqrt(res[2 * o] * res[2 * o] + res[2 * o + 1] * res[2 * o + 1]) / count[p];
    }
  }
  std::FILE* g = std::fopen(argv[2], "wb");
  if (!g) { std::perror(argv[2]); return 2; }
  const char omagic[8] = {'M', 'A', 'V', 'B', 'A', 'R', '1', 0};
  const int32_t ints[4] = {summary.num_successful_steps, summary.num_unsuccessful_steps, (int32_t)summary.termination_type,
                           (int32_t)summary.num_residuals};
  std::fwrite(omagic, 1, 8, g);
  std::fwrite(&summary.initial_cost, 8, 1, g); std::fwrite(&summary.final_cost, 8, 1, g);
  std::fwrite(ints, 4, 4, g); std::fwrite(&secs, 8, 1, g);
  std::fwrite(poses.data(), 8, poses.size(), g); std::fwrite(intr.data(), 8, intr.size(), g);
  std::fwrite(points.data(), 8, points.size(), g); std::fwrite(perr.data(), 8, perr.size(), g);
  std::fclose(g);
  return 0;
}
