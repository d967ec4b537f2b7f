// RECORDING MOCK of the C ABI (include/mavba.h) for the CPU tests of the shim: mavba_solve()
// stores a deep copy of the problem it was handed and returns without touching it. It computes
// nothing — it exists so that the flattening done by shim/base3d/bundle_adjustment.cc can be
// inspected on a machine without a GPU.
#include <cstring>
#include <vector>

#include "mavba.h"

namespace {
struct Recorded {
  mavba_problem P;
  mavba_options O;
  std::vector<double> poses, intr, points, uv, prior_rvec;
  std::vector<uint8_t> pose_const, intr_const, point_const;
  std::vector<int32_t> image_camera, camera_model, obs_image, obs_point, prior_image;
  int calls = 0;
  int want_point_error = 0;
} g;
}  // namespace

extern "C" {
void mavba_options_init(mavba_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 100; o->function_tolerance = 1e-4; o->gradient_tolerance = 1e-8;
  o->loss_scale_factor = 1.0; o->parameter_tolerance = 1e-8; o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32; o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->max_num_consecutive_invalid_steps = 10;
  o->jacobi_scaling = 1; o->device = -1;
}
const char* mavba_last_error(void) { return "mock"; }
int mavba_solve(const mavba_problem* P, const mavba_options* O, mavba_result* R, double* point_error) {
  g.calls++;
  g.P = *P; g.O = *O;
  g.want_point_error = point_error != nullptr;
  g.poses.assign(P->poses, P->poses + 6 * (size_t)P->num_images);
  g.pose_const.assign(P->pose_const, P->pose_const + P->num_images);
  g.image_camera.assign(P->image_camera, P->image_camera + P->num_images);
  g.intr.assign(P->intrinsics, P->intrinsics + 9 * (size_t)P->num_cameras);
  g.camera_model.assign(P->camera_model, P->camera_model + P->num_cameras);
  g.intr_const.assign(P->intr_const, P->intr_const + P->num_cameras);
  g.points.assign(P->points, P->points + 3 * (size_t)P->num_points);
  g.point_const.assign(P->point_const, P->point_const + P->num_points);
  g.uv.assign(P->obs_uv, P->obs_uv + 2 * (size_t)P->num_obs);
  g.obs_image.assign(P->obs_image, P->obs_image + P->num_obs);
  g.obs_point.assign(P->obs_point, P->obs_point + P->num_obs);
  g.prior_image.assign(P->rot_prior_image, P->rot_prior_image + P->num_rot_priors);
  g.prior_rvec.assign(P->rot_prior_rvec, P->rot_prior_rvec + 3 * (size_t)P->num_rot_priors);
  std::memset(R, 0, sizeof(*R));
  R->num_residuals = 2 * P->num_obs + P->num_rot_priors;
  R->initial_cost = R->final_cost = 2.0 * (double)R->num_residuals;  // -> returned "RMSE" == sqrt(2)
  if (point_error) for (int32_t p = 0; p < P->num_points; ++p) point_error[p] = 100.0 + p;
  return MAVBA_OK;
}
int mavba_pose_refine(double*, double*, const double*, int32_t, const double*, const double*, const uint8_t*, int64_t,
                      const mavba_options*, mavba_result* R) {
  std::memset(R, 0, sizeof(*R)); R->num_residuals = 2; R->final_cost = 4.0; return MAVBA_OK;
}
// ---- inspection -----------------------------------------------------------------------
int mock_calls(void) { return g.calls; }
void mock_sizes(int64_t* out) {
  out[0] = g.P.num_images; out[1] = g.P.num_cameras; out[2] = g.P.num_points; out[3] = g.P.num_obs;
  out[4] = g.P.num_rot_priors; out[5] = g.want_point_error;
}
const mavba_options* mock_options(void) { return &g.O; }
double mock_prior_weight(void) { return g.P.rot_prior_weight; }
const double* mock_poses(void) { return g.poses.data(); }
const uint8_t* mock_pose_const(void) { return g.pose_const.data(); }
const int32_t* mock_image_camera(void) { return g.image_camera.data(); }
const double* mock_intr(void) { return g.intr.data(); }
const int32_t* mock_camera_model(void) { return g.camera_model.data(); }
const uint8_t* mock_intr_const(void) { return g.intr_const.data(); }
const double* mock_points(void) { return g.points.data(); }
const uint8_t* mock_point_const(void) { return g.point_const.data(); }
const double* mock_uv(void) { return g.uv.data(); }
const int32_t* mock_obs_image(void) { return g.obs_image.data(); }
const int32_t* mock_obs_point(void) { return g.obs_point.data(); }
const int32_t* mock_prior_image(void) { return g.prior_image.data(); }
const double* mock_prior_rvec(void) { return g.prior_rvec.data(); }
}
