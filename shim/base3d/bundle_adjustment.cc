// bundle_adjustment.cc — replacement translation unit for MAVMAP's
// src/base3d/bundle_adjustment.cc: same two functions, no Ceres. It restates the reference's
// problem construction (which images / observations / constant blocks enter the problem) on
// the host and hands the flattened problem to the MI355X library through include/mavba.h.
//
// Restated (reference file:line, /root/reference):
//   validation                     src/base3d/bundle_adjustment.cc:459-471
//   observation selection + order  :228-286 (extract), :289-387 (fill), :493-533 (call order)
//   constancy per pose state       :361-385  (only when the image contributed > 1 residual)
//   rotation priors + pre-rotation :390-446  (+ src/base3d/projection.cc:12-23,
//                                   src/base3d/similarity_transform.cc:90-122)
//   GCP points                     :545-549
//   point3D_errors                 :575-598
//   report / return value          :114-136, :600-612
//   pose_refinement                :139-225
#include "base3d/bundle_adjustment.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <mutex>
#include <string>
#include <thread>

#include "mavba.h"

namespace {

// ---- a few host threads for the read-only walks over the FeatureManager's hash maps ----------------------------------
int host_threads() {
  // (MAVBA_SHIM_THREADS overrides; the default is a measurement: see INTEGRATION.md)
  static const int n = [] {
    if (const char* e = std::getenv("MAVBA_SHIM_THREADS")) return std::max(1, std::atoi(e));
    return (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  }();
  return n;
}
// The threads are started once and kept (a bundle_adjustment() call has six parallel passes; starting and joining 31 threads
// for each of them was milliseconds of a 40 ms call). job(t) runs on worker t = 1 .. T-1 while the caller does t = 0; calls
// are serialised. The pool is never destroyed: its threads may outlive static destruction.
class Workers {
  std::mutex m, run_m;
  std::condition_variable cv_start, cv_done;
  const std::function<void(int)>* job = nullptr;
  int job_T = 0, remaining = 0;
  unsigned long long generation = 0;
  int n = 0;
  std::exception_ptr error;  // the first exception a worker's job threw in the current run()
  void loop(int id) {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_start.wait(lk, [&] { return generation != seen; });
      seen = generation;
      if (id < job_T) {
        const std::function<void(int)>* j = job;
        lk.unlock();
        std::exception_ptr e;
        try { (*j)(id); } catch (...) { e = std::current_exception(); }  // (handed to the caller of run(); the thread lives on)
        lk.lock();
        if (e && !error) error = e;
        if (--remaining == 0) cv_done.notify_one();
      }
    }
  }

 public:
  explicit Workers(int count) : n(count) {
    for (int i = 1; i < count; ++i) { std::thread t([this, i] { loop(i); }); t.detach(); }
  }
  int size() const { return n; }
  void run(int T, const std::function<void(int)>& body) {  // body(0 .. T-1), T <= size()
    std::lock_guard<std::mutex> one(run_m);
    {
      std::unique_lock<std::mutex> lk(m);
      job = &body; job_T = T; remaining = T - 1; ++generation;
    }
    cv_start.notify_all();
    // Exception safe: whatever body(0) throws, the workers are waited for before `body` - which they run through
    // a pointer - and the caller's captures leave scope; the first exception of any thread is rethrown here.
    std::exception_ptr mine;
    try { body(0); } catch (...) { mine = std::current_exception(); }
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return remaining == 0; });
    job = nullptr; job_T = 0;
    std::exception_ptr e = mine ? mine : error;
    error = nullptr;
    lk.unlock();
    if (e) std::rethrow_exception(e);
  }
};
Workers& workers() {
  static Workers* W = new Workers(host_threads());
  return *W;
}
// body(begin, end, thread) over [0, n) in contiguous ranges; small jobs stay on the calling thread
template <class F>
void parallel_for(size_t n, F body) {
  const int T = n < 64 ? 1 : host_threads();
  if (T == 1) { body((size_t)0, n, 0); return; }
  // ranges of 1/(4T) of the work, dealt round-robin: images differ a lot in their number of 2-D points
  const size_t chunks = (size_t)4 * T;
  const std::function<void(int)> job = [&](int t) { for (size_t c = (size_t)t; c < chunks; c += (size_t)T) body(n * c / chunks, n * (c + 1) / chunks, t); };
  workers().run(T, job);
}

const double kEps = std::numeric_limits<double>::epsilon();

// ---- small SO(3) helpers on plain doubles (row-major 3x3) -------------------------------
// angle_axis_from_rvec(rvec).toRotationMatrix()  (projection.cc:12-23)
void rotation_from_rvec(const double* w, double* R) {
  double angle = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double ax[3] = {0, 0, 1};
  if (angle < kEps) {
    angle = 0;
  } else {
    for (int i = 0; i < 3; ++i) ax[i] = w[i] / angle;
  }
  const double c = std::cos(angle), s = std::sin(angle), t = 1 - c;
  R[0] = c + t * ax[0] * ax[0];         R[1] = t * ax[0] * ax[1] - s * ax[2]; R[2] = t * ax[0] * ax[2] + s * ax[1];
  R[3] = t * ax[0] * ax[1] + s * ax[2]; R[4] = c + t * ax[1] * ax[1];         R[5] = t * ax[1] * ax[2] - s * ax[0];
  R[6] = t * ax[0] * ax[2] - s * ax[1]; R[7] = t * ax[1] * ax[2] + s * ax[0]; R[8] = c + t * ax[2] * ax[2];
}

// Eigen::AngleAxisd(matrix): matrix -> quaternion -> (angle in [0, pi], axis); returns angle*axis.
void rvec_from_rotation(const double* R, double* w) {
  double q[4];  // w, x, y, z
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[1 + i] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  double n = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n != 0.0) {
    const double angle = 2.0 * std::atan2(n, std::fabs(q[0]));
    if (q[0] < 0) n = -n;
    for (int i = 0; i < 3; ++i) w[i] = angle * q[1 + i] / n;
  } else {
    w[0] = w[1] = w[2] = 0.0;
  }
}

void matmul3(const double* A, const double* B, double* C) {  // C = A B
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

int model_num_params(int code) {
  switch (code) {
    case MAVBA_MODEL_PINHOLE: return 4;
    case MAVBA_MODEL_OPENCV: return 8;
    case MAVBA_MODEL_CATA: return 9;
  }
  return -1;
}

void fill_options(const BundleAdjustmentOptions& o, mavba_options* m) {
  mavba_options_init(m);
  m->max_num_iterations = (int32_t)o.max_num_iterations;
  m->function_tolerance = o.function_tolerance;
  m->gradient_tolerance = o.gradient_tolerance;
  m->loss_scale_factor = o.loss_scale_factor;
  m->update_point_errors = o.update_point3D_errors ? 1 : 0;
  m->print_progress = o.print_progress ? 1 : 0;
}

// Error policy at the boundary. The reference never fails inside ceres::Solve, so mapper.cc has no handler beyond the
// two std::invalid_argument checks it can trigger itself. Argument errors of the flat problem (a bug in this file or a
// corrupt FeatureManager) become std::invalid_argument like those; a device-side failure (HIP error, out of device
// memory) is retried ONCE by the callers below - the session is rebuilt from the untouched host data - and only then
// surfaces as std::runtime_error, which ends the run like any other uncaught exception there.
bool transient(int code) { return code == MAVBA_ERR_HIP || code == MAVBA_ERR_OUT_OF_MEMORY; }

[[noreturn]] void raise(int code) {
  const std::string msg = std::string("mavba: ") + mavba_last_error();
  if (code == MAVBA_ERR_INVALID_ARGUMENT || code == MAVBA_ERR_BAD_INDEX || code == MAVBA_ERR_BAD_MODEL)
    throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// _print_report (bundle_adjustment.cc:114-136)
void print_report(const mavba_result& s) {
  std::cout << std::right << std::setw(18) << "Residuals : " << std::left << s.num_residuals_reduced << std::endl;
  std::cout << std::right << std::setw(18) << "Parameters : " << std::left << s.num_parameters_reduced << std::endl;
  std::cout << std::right << std::setw(18) << "Iterations : " << std::left
            << s.num_successful_steps + s.num_unsuccessful_steps << std::endl;
  std::cout << std::right << std::setw(18) << "Initial cost : " << std::right << std::setprecision(6)
            << std::sqrt(s.initial_cost / s.num_residuals) << " [px]" << std::endl;
  std::cout << std::right << std::setw(18) << "Final cost : " << std::right << std::setprecision(6)
            << std::sqrt(s.final_cost / s.num_residuals) << " [px]" << std::endl;
  std::cout << std::endl;
}

}  // namespace

double pose_refinement(Eigen::Vector3d& rvec, Eigen::Vector3d& tvec, std::vector<double>& camera_params,
                       const std::vector<Eigen::Vector2d>& points2D, std::vector<Eigen::Vector3d>& points3D,
                       const std::vector<bool>& inlier_mask, const BundleAdjustmentOptions& options) {
  const int model = (int)camera_params.back();
  const size_t n = points2D.size();
  std::vector<double> uv(2 * n), xyz(3 * n);
  std::vector<uint8_t> mask(n);
  for (size_t i = 0; i < n; ++i) {
    uv[2 * i] = points2D[i](0); uv[2 * i + 1] = points2D[i](1);
    xyz[3 * i] = points3D[i](0); xyz[3 * i + 1] = points3D[i](1); xyz[3 * i + 2] = points3D[i](2);
    mask[i] = inlier_mask[i] ? 1 : 0;
  }
  mavba_options mo;
  fill_options(options, &mo);
  mavba_result res;
  int rc = mavba_pose_refine(rvec.data(), tvec.data(), camera_params.data(), model, uv.data(), xyz.data(),
                             mask.data(), (int64_t)n, &mo, &res);
  if (transient(rc))  // (inputs are only written on success)
    rc = mavba_pose_refine(rvec.data(), tvec.data(), camera_params.data(), model, uv.data(), xyz.data(), mask.data(), (int64_t)n, &mo, &res);
  if (rc != MAVBA_OK) raise(rc);
  if (options.print_progress) std::cout << std::endl;
  if (options.print_summary) {
    std::cout << "Pose Refinement Report" << std::endl;
    std::cout << "----------------------" << std::endl;
    print_report(res);
  }
  return std::sqrt(res.final_cost / res.num_residuals);
}

// Optional replay file of the flattened problem (set MAVBA_DUMP_DIR): lets a maintainer benchmark the backend on the
// problems a real MAVMAP run produces (`python bench.py --problem <file>`), without the rest of the pipeline.
// Layout (little endian): "MAVBA1\0\0", int32 NI NC NP NPRI, int64 NO, double prior_weight, double opts[4]
// {max_num_iterations, function_tolerance, gradient_tolerance, loss_scale_factor}, then the arrays of
// mavba_problem in declaration order.
static void dump_problem(const mavba_problem& P, const mavba_options& mo) {
  const char* dir = std::getenv("MAVBA_DUMP_DIR");
  if (!dir || !*dir) return;
  static int counter = 0;
  char path[1024];
  std::snprintf(path, sizeof(path), "%s/mavba_problem_%05d_%dimg.bin", dir, counter++, (int)P.num_images);
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return;
  const char magic[8] = {'M', 'A', 'V', 'B', 'A', '1', 0, 0};
  const int32_t dims[4] = {P.num_images, P.num_cameras, P.num_points, P.num_rot_priors};
  const int64_t no = P.num_obs;
  const double opts[4] = {(double)mo.max_num_iterations, mo.function_tolerance, mo.gradient_tolerance, mo.loss_scale_factor};
  bool ok = true;
  auto put = [&](const void* p, size_t size, size_t count) { if (count && std::fwrite(p, size, count, f) != count) ok = false; };
  put(magic, 1, 8); put(dims, 4, 4); put(&no, 8, 1);
  put(&P.rot_prior_weight, 8, 1); put(opts, 8, 4);
  const size_t NI = (size_t)P.num_images, NC = (size_t)P.num_cameras, NP = (size_t)P.num_points, NO = (size_t)P.num_obs, NR = (size_t)P.num_rot_priors;
  put(P.poses, 8, NI * 6); put(P.pose_const, 1, NI); put(P.image_camera, 4, NI);
  put(P.intrinsics, 8, NC * MAVBA_MAX_INTR); put(P.camera_model, 4, NC); put(P.intr_const, 1, NC);
  put(P.points, 8, NP * 3); put(P.point_const, 1, NP);
  put(P.obs_uv, 8, NO * 2); put(P.obs_image, 4, NO); put(P.obs_point, 4, NO);
  put(P.rot_prior_image, 4, NR); put(P.rot_prior_rvec, 8, NR * 3);
  if (std::fclose(f) != 0) ok = false;
  if (!ok) {  // a truncated replay file would silently benchmark a different problem
    std::remove(path);
    std::cerr << "mavba: could not write the replay file " << path << " (disk full?) - removed" << std::endl;
  }
}


double bundle_adjustment(FeatureManager& fm, const std::vector<size_t>& free_image_ids,
                         const std::vector<size_t>& fixed_image_ids, const std::vector<size_t>& fixed_x_image_ids,
                         const BundleAdjustmentOptions& options, std::unordered_map<size_t, double>& point3D_errors,
                         const std::unordered_map<size_t, Eigen::Vector3d>& rotation_constraints,
                         const std::set<size_t>& gcp_ids) {
  const size_t num_fixed_params = fixed_image_ids.size() * 6 + fixed_x_image_ids.size() + gcp_ids.size() * 3;
  if (num_fixed_params < 7) {
    throw std::invalid_argument("At least 7 parameters should be set as fixed to avoid datum defects resulting in a "
                                "singular Jacobian.");
  }
  if (options.min_track_len < 2) {
    throw std::invalid_argument("Minimum track length must be >= 2 in order build valid bundle adjustment problem.");
  }

  // Rotation priors: the reference first rotates EVERY pose and point of the feature manager so
  // that the first fixed image agrees with its prior (bundle_adjustment.cc:399-425).
  if (options.constrain_rotation) {
    if (fixed_image_ids.empty())
      throw std::out_of_range("constrain_rotation needs a fixed image (the reference reads fixed_image_ids[0])");
    const size_t ref_id = fixed_image_ids[0];
    double R_fm[9], R_c[9], S[9], R_fm_t[9];
    rotation_from_rvec(fm.rvecs.at(ref_id).data(), R_fm);
    rotation_from_rvec(rotation_constraints.at(ref_id).data(), R_c);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R_fm_t[i * 3 + j] = R_fm[j * 3 + i];
    matmul3(R_fm_t, R_c, S);  // rotation from the SfM to the constraint frame
    double St[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) St[i * 3 + j] = S[j * 3 + i];
    for (auto it = fm.rvecs.begin(); it != fm.rvecs.end(); ++it) {
      // transform_pose: [R | t] S^-1; S is a pure rotation, so t is unchanged
      double R[9], Rn[9];
      rotation_from_rvec(it->second.data(), R);
      matmul3(R, St, Rn);
      rvec_from_rotation(Rn, it->second.data());
    }
    for (auto it = fm.points3D.begin(); it != fm.points3D.end(); ++it) {
      double* X = it->second.data();
      const double x = X[0], y = X[1], z = X[2];
      X[0] = S[0] * x + S[1] * y + S[2] * z;
      X[1] = S[3] * x + S[4] * y + S[5] * z;
      X[2] = S[6] * x + S[7] * y + S[8] * z;
    }
  }

  // ---- which observations enter (extract_data, :228-286) and the flat problem in the reference's residual-block
  // order FREE, FIXED, FIXED_X (:511-533).
  // The reference (and the first version of this file) walks the FeatureManager's hash maps observation by observation
  // on one thread: ~3 look-ups per 2-D point, 0.4 s for a 2 M-observation global BA - several times the device solve.
  // Here the look-ups run on a few threads (the maps are only read), everything else works on dense arrays:
  //   1. per listed image (parallel): its 2-D points that have a 3-D point               (point2D_to_point3D)
  //   2. per 3-D point: observations inside the selected image set                         (dense counter)
  //   3. per listed image (parallel): keep count >= min_track_len, fetch the pixels       (points2D)
  //   4. serial, no hashing: concatenate in list order, number points by first appearance
  //   5. per point (parallel): coordinates                                                 (points3D)
  // The result is the same flat problem, element for element, as the serial walk produced.
  const bool timing = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now();
  auto lap = [&](const char* what) { if (timing) { const double t = now(); std::fprintf(stderr, "[shim] %-30s %8.2f ms\n", what, 1e3 * (t - t_lap)); t_lap = t; } };
  struct Entry { size_t image_id; int state; std::vector<size_t> p2d, p3d; std::vector<double> xy; size_t kept = 0; };
  std::vector<Entry> entries;
  {
    const std::vector<size_t>* fill_order[3] = {&free_image_ids, &fixed_image_ids, &fixed_x_image_ids};
    const int fill_state[3] = {BA_POSE_FREE, BA_POSE_FIXED, BA_POSE_FIXED_X};
    for (int l = 0; l < 3; ++l)
      for (size_t image_id : *fill_order[l]) { Entry e; e.image_id = image_id; e.state = fill_state[l]; entries.push_back(std::move(e)); }
  }
  size_t max_p3d = 0;
  {
    std::vector<size_t> tmax(host_threads(), 0);
    parallel_for(entries.size(), [&](size_t e0, size_t e1, int t) {
      for (size_t e = e0; e < e1; ++e) {
        Entry& E = entries[e];
        auto it = fm.image_to_points2D.find(E.image_id);
        if (it == fm.image_to_points2D.end()) continue;
        for (size_t point2D_id : it->second) {
          auto it3 = fm.point2D_to_point3D.find(point2D_id);
          if (it3 == fm.point2D_to_point3D.end()) continue;
          E.p2d.push_back(point2D_id); E.p3d.push_back(it3->second);
          if (it3->second > tmax[t]) tmax[t] = it3->second;
        }
      }
    });
    for (size_t m : tmax) max_p3d = std::max(max_p3d, m);
  }
  lap("2-D points with a 3-D point");
  std::vector<uint32_t> point3D_num_points2D(max_p3d + 1, 0);  // an image listed twice counts twice, as in the reference
  size_t total_p3d = 0;
  for (const Entry& E : entries) total_p3d += E.p3d.size();
  // (MAVBA_SHIM_ASSEMBLY=serial | blocks forces one form: tests compare them on small scenes)
  const char* asm_env = std::getenv("MAVBA_SHIM_ASSEMBLY");
  const bool asm_serial = asm_env && std::string(asm_env) == "serial", asm_blocks = asm_env && std::string(asm_env) == "blocks";
  const bool big_call = host_threads() > 1 && !asm_serial && (asm_blocks || total_p3d >= 200000);
  // ONE contiguous block of entries per worker, cut at equal weights: neighbouring images see the same 3-D points, so a point's
  // owner slot is shared by the threads of adjacent blocks only
  auto block_bounds = [&](const std::function<size_t(const Entry&)>& weight) {
    const size_t nblk = std::min(entries.size(), (size_t)host_threads());
    std::vector<size_t> bound(nblk + 1, entries.size());
    size_t total = 0, run = 0, b = 1;
    for (const Entry& E : entries) total += weight(E);
    bound[0] = 0;
    for (size_t e = 0; e < entries.size() && b < nblk; ++e) {
      run += weight(entries[e]);
      while (b < nblk && run * nblk >= total * b) bound[b++] = e + 1;
    }
    return bound;
  };
  auto over_blocks = [&](const std::vector<size_t>& bound, const std::function<void(size_t, size_t, size_t)>& body) {
    const size_t nblk = bound.size() - 1;
    const std::function<void(int)> job = [&](int t) { if ((size_t)t < nblk) body((size_t)t, bound[t], bound[t + 1]); };
    workers().run((int)nblk, job);
  };
  // (the counting stays serial: with relaxed atomic increments on the worker threads it took 9 ms instead of 3.5 on the GPU box's
  // host, contiguous blocks or not - every increment is a locked read-modify-write on a line other cores also write)
  for (const Entry& E : entries) for (size_t id : E.p3d) point3D_num_points2D[id] += 1;
  parallel_for(entries.size(), [&](size_t e0, size_t e1, int) {
    for (size_t e = e0; e < e1; ++e) {
      Entry& E = entries[e];
      size_t k = 0;
      for (size_t i = 0; i < E.p3d.size(); ++i) {
        if (point3D_num_points2D[E.p3d[i]] < options.min_track_len) continue;   // :330
        E.p2d[k] = E.p2d[i]; E.p3d[k] = E.p3d[i]; ++k;
      }
      E.kept = k;
      E.xy.resize(2 * k);
      for (size_t i = 0; i < k; ++i) {
        auto ixy = fm.points2D.find(E.p2d[i]);  // (a missing entry reads as (0, 0): what operator[] gives the reference)
        if (ixy == fm.points2D.end()) continue;
        E.xy[2 * i] = ixy->second(0); E.xy[2 * i + 1] = ixy->second(1);
      }
    }
  });

  lap("counts, filter, pixels");
  std::vector<size_t> image_ids, camera_ids, point_ids;           // flat index -> feature-manager id
  std::unordered_map<size_t, int32_t> image_index, camera_index;
  std::vector<int32_t> point_index(max_p3d + 1, -1);
  std::vector<double> poses, intrinsics, points, obs_uv;
  std::vector<uint8_t> pose_const, intr_const, point_const;
  std::vector<int32_t> image_camera, camera_model, obs_image, obs_point;
  // Per entry (a few hundred, serial): images and cameras by first appearance, constancy, where its observations go.
  std::vector<int32_t> entry_img(entries.size(), -1);
  std::vector<size_t> entry_off(entries.size() + 1, 0);
  for (size_t e = 0; e < entries.size(); ++e) {
    Entry& E = entries[e];
    entry_off[e + 1] = entry_off[e] + E.kept;
    if (E.kept == 0) continue;  // no residual block: the image does not enter the problem here
    const size_t image_id = E.image_id;
    // An image id listed twice (in one list or in two) is ONE set of parameter blocks in the reference: its
    // residual blocks are added again on the same blocks and the constancy settings accumulate.
    int32_t img;
    auto known = image_index.find(image_id);
    if (known != image_index.end()) {
      img = known->second;
    } else {
      const size_t camera_id = fm.image_to_camera[image_id];
      std::vector<double>& cam = fm.camera_params[camera_id];
      const int model = (int)cam.back();
      const int K = model_num_params(model);
      if (K < 0) throw std::invalid_argument("unknown camera model code");
      auto ic = camera_index.find(camera_id);
      if (ic == camera_index.end()) {
        ic = camera_index.emplace(camera_id, (int32_t)camera_ids.size()).first;
        camera_ids.push_back(camera_id);
        camera_model.push_back(model);
        intr_const.push_back(0);
        for (int k = 0; k < MAVBA_MAX_INTR; ++k) intrinsics.push_back(k < K ? cam[k] : 0.0);
      }
      img = (int32_t)image_ids.size();
      image_index[image_id] = img;
      image_ids.push_back(image_id);
      image_camera.push_back(ic->second);
      pose_const.push_back(0);
      const double* r = fm.rvecs[image_id].data();
      const double* t = fm.tvecs[image_id].data();
      poses.insert(poses.end(), r, r + 3);
      poses.insert(poses.end(), t, t + 3);
    }
    entry_img[e] = img;
    // Constancy is applied only if the image contributed more than one residual (:361).
    if (E.kept > 1) {
      if (E.state == BA_POSE_FIXED) pose_const[img] |= MAVBA_CONST_POSE;
      if (E.state == BA_POSE_FIXED_X) pose_const[img] |= MAVBA_CONST_TX;
      if (!options.refine_camera_params) intr_const[image_camera[img]] = 1;
    }
  }
  const size_t total_obs = entry_off.back();
  obs_uv.resize(2 * total_obs); obs_image.resize(total_obs); obs_point.resize(total_obs);
  if (!big_call) {
    // the observations concatenated in list order, points numbered by first appearance
    for (size_t e = 0; e < entries.size(); ++e) {
      Entry& E = entries[e];
      for (size_t i = 0; i < E.kept; ++i) {
        int32_t& ip = point_index[E.p3d[i]];
        if (ip < 0) { ip = (int32_t)point_ids.size(); point_ids.push_back(E.p3d[i]); }
        const size_t o = entry_off[e] + i;
        obs_uv[2 * o] = E.xy[2 * i]; obs_uv[2 * o + 1] = E.xy[2 * i + 1];
        obs_image[o] = entry_img[e];
        obs_point[o] = ip;
      }
    }
  } else {
    // Round 6, the same arrays without the serial walk over 2 M observations (7 of the call's 35 ms at C3). The entries are
    // cut into BLOCKS that are contiguous in list order. A point's first appearance lies in the lowest block that sees it
    // (atomic minimum per point, pass A); the block that owns a point numbers it locally in the order it first meets it
    // (pass B: only the owner writes the point's slot); a prefix sum over the blocks' counts turns local numbers into the
    // serial walk's numbers (points first met in block b follow all points first met in earlier blocks, in b's own order of
    // first appearance), and pass C writes the observations. No per-thread tables, no zero-fills beyond the two dense arrays.
    const std::vector<size_t> bound = block_bounds([](const Entry& E) { return E.kept; });
    const size_t nblk = bound.size() - 1;
    std::vector<uint32_t> first_blk(max_p3d + 1, 0xFFFFFFFFu);
    over_blocks(bound, [&](size_t b, size_t e0, size_t e1) {
      for (size_t e = e0; e < e1; ++e) {
        const Entry& E = entries[e];
        for (size_t i = 0; i < E.kept; ++i) {
          uint32_t* f = &first_blk[E.p3d[i]];
          uint32_t cur = __atomic_load_n(f, __ATOMIC_RELAXED);
          while (cur > (uint32_t)b && !__atomic_compare_exchange_n(f, &cur, (uint32_t)b, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
      }
    });
    std::vector<std::vector<size_t>> blk_points(nblk);
    over_blocks(bound, [&](size_t b, size_t e0, size_t e1) {
      std::vector<size_t>& mine = blk_points[b];
      for (size_t e = e0; e < e1; ++e) {
        const Entry& E = entries[e];
        for (size_t i = 0; i < E.kept; ++i) {
          const size_t id = E.p3d[i];
          if (first_blk[id] != (uint32_t)b || point_index[id] >= 0) continue;
          point_index[id] = (int32_t)mine.size();
          mine.push_back(id);
        }
      }
    });
    std::vector<size_t> blk_base(nblk + 1, 0);
    for (size_t b = 0; b < nblk; ++b) blk_base[b + 1] = blk_base[b] + blk_points[b].size();
    point_ids.resize(blk_base[nblk]);
    over_blocks(bound, [&](size_t b, size_t e0, size_t e1) {
      std::copy(blk_points[b].begin(), blk_points[b].end(), point_ids.begin() + blk_base[b]);
      for (size_t e = e0; e < e1; ++e) {
        const Entry& E = entries[e];
        for (size_t i = 0; i < E.kept; ++i) {
          const size_t id = E.p3d[i], o = entry_off[e] + i;
          obs_uv[2 * o] = E.xy[2 * i]; obs_uv[2 * o + 1] = E.xy[2 * i + 1];
          obs_image[o] = entry_img[e];
          obs_point[o] = (int32_t)(blk_base[first_blk[id]] + (size_t)point_index[id]);
        }
      }
    });
  }
  for (Entry& E : entries) { std::vector<size_t>().swap(E.p2d); std::vector<size_t>().swap(E.p3d); std::vector<double>().swap(E.xy); }
  lap("assembly in list order");
  points.resize(3 * point_ids.size());
  point_const.resize(point_ids.size());
  parallel_for(point_ids.size(), [&](size_t p0, size_t p1, int) {
    for (size_t p = p0; p < p1; ++p) {
      auto iX = fm.points3D.find(point_ids[p]);
      if (iX != fm.points3D.end()) { points[3 * p] = iX->second(0); points[3 * p + 1] = iX->second(1); points[3 * p + 2] = iX->second(2); }
      point_const[p] = gcp_ids.count(point_ids[p]) ? 1 : 0;  // :545-549
    }
  });

  lap("point coordinates");
  // rotation-prior residuals for the FREE images (:428-444)
  std::vector<int32_t> prior_image;
  std::vector<double> prior_rvec;
  if (options.constrain_rotation) {
    for (size_t image_id : free_image_ids) {
      const Eigen::Vector3d& rvec0 = rotation_constraints.at(image_id);
      auto ii = image_index.find(image_id);
      if (ii == image_index.end()) {
        // an image without residual blocks still gets its prior in the reference
        const size_t camera_id = fm.image_to_camera[image_id];
        std::vector<double>& cam = fm.camera_params[camera_id];
        const int model = (int)cam.back();
        const int K = model_num_params(model);
        if (K < 0) throw std::invalid_argument("unknown camera model code");
        auto ic = camera_index.find(camera_id);
        if (ic == camera_index.end()) {
          ic = camera_index.emplace(camera_id, (int32_t)camera_ids.size()).first;
          camera_ids.push_back(camera_id);
          camera_model.push_back(model);
          intr_const.push_back(1);  // no residual touches it through this image
          for (int k = 0; k < MAVBA_MAX_INTR; ++k) intrinsics.push_back(k < K ? cam[k] : 0.0);
        }
        ii = image_index.emplace(image_id, (int32_t)image_ids.size()).first;
        image_ids.push_back(image_id);
        image_camera.push_back(ic->second);
        pose_const.push_back(MAVBA_CONST_TX | MAVBA_CONST_TY | MAVBA_CONST_TZ);  // only rvec is in the problem
        const double* r = fm.rvecs[image_id].data();
        const double* t = fm.tvecs[image_id].data();
        poses.insert(poses.end(), r, r + 3);
        poses.insert(poses.end(), t, t + 3);
      }
      prior_image.push_back(ii->second);
      prior_rvec.push_back(rvec0(0)); prior_rvec.push_back(rvec0(1)); prior_rvec.push_back(rvec0(2));
    }
  }

  mavba_problem P;
  P.num_images = (int32_t)image_ids.size();
  P.num_cameras = (int32_t)camera_ids.size();
  P.num_points = (int32_t)point_ids.size();
  P.num_obs = (int64_t)obs_image.size();
  P.poses = poses.data(); P.pose_const = pose_const.data(); P.image_camera = image_camera.data();
  P.intrinsics = intrinsics.data(); P.camera_model = camera_model.data(); P.intr_const = intr_const.data();
  P.points = points.data(); P.point_const = point_const.data();
  P.obs_uv = obs_uv.data(); P.obs_image = obs_image.data(); P.obs_point = obs_point.data();
  P.num_rot_priors = (int32_t)prior_image.size();
  P.rot_prior_image = prior_image.data(); P.rot_prior_rvec = prior_rvec.data();
  P.rot_prior_weight = options.constrain_rotation_weight;

  mavba_options mo;
  fill_options(options, &mo);
  std::vector<double> perr(point_ids.size(), 0.0);
  mavba_result res;
  dump_problem(P, mo);
  int rc = mavba_solve(&P, &mo, &res, options.update_point3D_errors ? perr.data() : nullptr);
  if (transient(rc))  // (the flat arrays are only written back after a completed solve)
    rc = mavba_solve(&P, &mo, &res, options.update_point3D_errors ? perr.data() : nullptr);
  if (rc != MAVBA_OK) raise(rc);

  lap("mavba_solve");
  // ---- write back in place (the reference lets Ceres write through raw pointers)
  for (size_t i = 0; i < image_ids.size(); ++i) {
    double* r = fm.rvecs[image_ids[i]].data();
    double* t = fm.tvecs[image_ids[i]].data();
    for (int k = 0; k < 3; ++k) { r[k] = poses[6 * i + k]; t[k] = poses[6 * i + 3 + k]; }
  }
  for (size_t c = 0; c < camera_ids.size(); ++c) {
    std::vector<double>& cam = fm.camera_params[camera_ids[c]];
    const int K = model_num_params(camera_model[c]);
    for (int k = 0; k < K; ++k) cam[k] = intrinsics[MAVBA_MAX_INTR * c + k];  // the model code (last slot) is never written
  }
  parallel_for(point_ids.size(), [&](size_t p0, size_t p1, int) {  // (values of existing nodes: no rehash, safe in parallel)
    for (size_t p = p0; p < p1; ++p) {
      auto iX = fm.points3D.find(point_ids[p]);
      if (iX == fm.points3D.end()) continue;
      for (int k = 0; k < 3; ++k) iX->second(k) = points[3 * p + k];
    }
  });

  if (obs_image.empty()) {
    std::cout << "No observations in bundle adjustment. Consider relaxing the constraints." << std::endl;
  }
  if (options.update_point3D_errors) {
    point3D_errors.reserve(point3D_errors.size() + point_ids.size());
    for (size_t p = 0; p < point_ids.size(); ++p) point3D_errors[point_ids[p]] = perr[p];
  }
  lap("write-back + point3D_errors");
  if (options.print_progress) std::cout << std::endl;
  if (options.print_summary) {
    std::cout << "Bundle Adjustment Report" << std::endl;
    std::cout << "------------------------" << std::endl;
    print_report(res);
  }
  return std::sqrt(res.final_cost / res.num_residuals);
}
