// Test driver for the shim: builds a FeatureManager (test double) from flat arrays, calls the
// reference-signature bundle_adjustment() / pose_refinement() of shim/base3d/bundle_adjustment.cc,
// and copies the mutated scene back. Linked either against the recording mock (CPU tests) or
// against the real libmavba.so (GPU tests).
#include <chrono>
#include <cmath>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>

#include "base3d/bundle_adjustment.h"

static std::string g_what;
static double g_ba_seconds = 0.0;

extern "C" {

// wall time of the last bundle_adjustment() call itself (without building / reading back the FeatureManager)
double shim_last_ba_seconds(void) { return g_ba_seconds; }

const char* shim_last_exception(void) { return g_what.c_str(); }

// ids are 1-based in the feature manager: image i -> i+1, camera c -> c+1, point p -> p+1,
// observation o -> o+1 (added in the given order, which fixes image_to_points2D order).
// returns 0 ok, 1 std::invalid_argument, 2 std::out_of_range, 3 other exception
int shim_bundle_adjustment(
    int n_cam, double* cam_params /*[n_cam][10]: 9 params + model code*/, int n_img, const int* img_cam,
    double* poses /*[n_img][6]*/, int n_pt, double* points /*[n_pt][3]*/, long long n_obs, const int* obs_img,
    const int* obs_pt /* -1: no 3-D point */, const double* obs_uv, int n_free, const int* free_imgs, int n_fixed,
    const int* fixed_imgs, int n_fixed_x, const int* fixed_x_imgs, int n_gcp, const int* gcp_pts,
    const double* rot_constraints /*[n_img][3] or NULL*/,
    // options
    int max_num_iterations, double function_tolerance, double gradient_tolerance, int update_point3D_errors,
    int min_track_len, double loss_scale_factor, int constrain_rotation, double constrain_rotation_weight,
    int refine_camera_params, int print_summary,
    // outputs
    double* point_errors /*[n_pt], NaN = not in the map*/, double* ret) {
  try {
    FeatureManager fm;
    for (int c = 0; c < n_cam; ++c) {
      const int model = (int)cam_params[c * 10 + 9];
      const int K = model == 1 ? 4 : model == 2 ? 8 : 9;
      std::vector<double> v(cam_params + c * 10, cam_params + c * 10 + K);
      v.push_back((double)model);
      fm.camera_params[c + 1] = v;
    }
    for (int i = 0; i < n_img; ++i) {
      fm.image_to_camera[i + 1] = img_cam[i] + 1;
      fm.rvecs[i + 1] = Eigen::Vector3d(poses[i * 6], poses[i * 6 + 1], poses[i * 6 + 2]);
      fm.tvecs[i + 1] = Eigen::Vector3d(poses[i * 6 + 3], poses[i * 6 + 4], poses[i * 6 + 5]);
      fm.image_to_points2D[i + 1];
    }
    for (int p = 0; p < n_pt; ++p) fm.points3D[p + 1] = Eigen::Vector3d(points[p * 3], points[p * 3 + 1], points[p * 3 + 2]);
    for (long long o = 0; o < n_obs; ++o) {
      fm.points2D[o + 1] = Eigen::Vector2d(obs_uv[2 * o], obs_uv[2 * o + 1]);
      fm.image_to_points2D[obs_img[o] + 1].push_back(o + 1);
      fm.point2D_to_image[o + 1] = obs_img[o] + 1;
      if (obs_pt[o] >= 0) { fm.point2D_to_point3D[o + 1] = obs_pt[o] + 1; fm.point3D_to_points2D[obs_pt[o] + 1].push_back(o + 1); }
    }
    std::vector<size_t> fr, fx, fxx;
    for (int i = 0; i < n_free; ++i) fr.push_back(free_imgs[i] + 1);
    for (int i = 0; i < n_fixed; ++i) fx.push_back(fixed_imgs[i] + 1);
    for (int i = 0; i < n_fixed_x; ++i) fxx.push_back(fixed_x_imgs[i] + 1);
    std::set<size_t> gcp;
    for (int i = 0; i < n_gcp; ++i) gcp.insert(gcp_pts[i] + 1);
    std::unordered_map<size_t, Eigen::Vector3d> rc;
    if (rot_constraints)
      for (int i = 0; i < n_img; ++i)
        if (!std::isnan(rot_constraints[3 * i]))
          rc[i + 1] = Eigen::Vector3d(rot_constraints[3 * i], rot_constraints[3 * i + 1], rot_constraints[3 * i + 2]);
    BundleAdjustmentOptions o;
    o.max_num_iterations = max_num_iterations; o.function_tolerance = function_tolerance;
    o.gradient_tolerance = gradient_tolerance; o.update_point3D_errors = update_point3D_errors != 0;
    o.min_track_len = min_track_len; o.loss_scale_factor = loss_scale_factor;
    o.constrain_rotation = constrain_rotation != 0; o.constrain_rotation_weight = constrain_rotation_weight;
    o.refine_camera_params = refine_camera_params != 0; o.print_progress = false; o.print_summary = print_summary != 0;
    std::unordered_map<size_t, double> perr;
    perr[999999] = -1.0;  // an unrelated entry must survive untouched
    const auto t0 = std::chrono::steady_clock::now();
    *ret = bundle_adjustment(fm, fr, fx, fxx, o, perr, rc, gcp);
    g_ba_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (perr.at(999999) != -1.0) throw std::runtime_error("unrelated point3D_errors entry was modified");
    for (int c = 0; c < n_cam; ++c) {
      const std::vector<double>& v = fm.camera_params[c + 1];
      for (size_t k = 0; k + 1 < v.size(); ++k) cam_params[c * 10 + k] = v[k];
      cam_params[c * 10 + 9] = v.back();
    }
    for (int i = 0; i < n_img; ++i)
      for (int k = 0; k < 3; ++k) { poses[i * 6 + k] = fm.rvecs[i + 1](k); poses[i * 6 + 3 + k] = fm.tvecs[i + 1](k); }
    for (int p = 0; p < n_pt; ++p) {
      for (int k = 0; k < 3; ++k) points[p * 3 + k] = fm.points3D[p + 1](k);
      point_errors[p] = perr.count(p + 1) ? perr[p + 1] : std::nan("");
    }
    return 0;
  } catch (const std::invalid_argument& e) { g_what = e.what(); return 1;
  } catch (const std::out_of_range& e) { g_what = e.what(); return 2;
  } catch (const std::exception& e) { g_what = e.what(); return 3; }
}

int shim_pose_refinement(double* rvec, double* tvec, const double* cam_params10, long long n, const double* uv,
                         const double* xyz, const unsigned char* mask, double loss_scale_factor, double* ret) {
  try {
    Eigen::Vector3d r(rvec[0], rvec[1], rvec[2]), t(tvec[0], tvec[1], tvec[2]);
    const int model = (int)cam_params10[9];
    const int K = model == 1 ? 4 : model == 2 ? 8 : 9;
    std::vector<double> cam(cam_params10, cam_params10 + K);
    cam.push_back((double)model);
    std::vector<Eigen::Vector2d> p2(n);
    std::vector<Eigen::Vector3d> p3(n);
    std::vector<bool> m(n);
    for (long long i = 0; i < n; ++i) {
      p2[i] = Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]);
      p3[i] = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      m[i] = mask[i] != 0;
    }
    BundleAdjustmentOptions o;
    o.loss_scale_factor = loss_scale_factor; o.print_summary = false;
    *ret = pose_refinement(r, t, cam, p2, p3, m, o);
    for (int k = 0; k < 3; ++k) { rvec[k] = r(k); tvec[k] = t(k); }
    return 0;
  } catch (const std::exception& e) { g_what = e.what(); return 3; }
}
}
