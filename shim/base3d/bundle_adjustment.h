// bundle_adjustment.h — drop-in replacement for MAVMAP's src/base3d/bundle_adjustment.h.
//
// Source-compatible with everything outside the bundle-adjustment translation unit:
//   * src/sfm/sequential_mapper.h:29 includes this header and src/mapper.cc:588-590 instantiates
//     BundleAdjustmentOptions;
//   * SequentialMapper calls bundle_adjustment() (src/sfm/sequential_mapper.cc:1074-1080,
//     :1150-1157) and pose_refinement() (:715-720).
// The Ceres cost-functor classes the original header also declares
// (reference src/base3d/bundle_adjustment.h:117-209) are used only inside the original .cc
// and are intentionally gone: this backend needs neither Ceres nor its headers.
//
// The implementation (bundle_adjustment.cc next to this file) flattens FeatureManager into
// the plain arrays of include/mavba.h and calls the MI355X library through its C ABI.
#ifndef MAVMAP_SRC_BASE3D_BUNDLE_ADJUSTMENT_H_
#define MAVMAP_SRC_BASE3D_BUNDLE_ADJUSTMENT_H_

#include <set>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include <Eigen/Core>

#include "fm/feature_management.h"

#define BA_POSE_FREE       0
#define BA_POSE_FIXED      1
#define BA_POSE_FIXED_X    2

// Same members, same defaults, same order as the reference struct
// (reference src/base3d/bundle_adjustment.h:38-114).
struct BundleAdjustmentOptions {
  BundleAdjustmentOptions()
      : max_num_iterations(100), function_tolerance(1e-4), gradient_tolerance(1e-8),
        update_point3D_errors(false), min_track_len(2), loss_scale_factor(1),
        constrain_rotation(false), constrain_rotation_weight(0), refine_camera_params(false),
        print_progress(false), print_summary(true) {}

  size_t max_num_iterations;         // maximum number of LM iterations
  double function_tolerance;         // |cost change| < function_tolerance * cost
  double gradient_tolerance;         // max|g| < gradient_tolerance * max|g_initial|
  bool update_point3D_errors;        // fill the per-point mean reprojection error map
  size_t min_track_len;              // minimum #observations (in the selected images) per point
  double loss_scale_factor;          // Cauchy loss scale
  bool constrain_rotation;           // add one rotation-prior residual per free image
  double constrain_rotation_weight;  // weight of those residuals
  bool refine_camera_params;         // intrinsics variable
  bool print_progress;               // per-iteration table
  bool print_summary;                // final report
};

double pose_refinement(Eigen::Vector3d& rvec, Eigen::Vector3d& tvec,
                       std::vector<double>& camera_params,
                       const std::vector<Eigen::Vector2d>& points2D,
                       std::vector<Eigen::Vector3d>& points3D,
                       const std::vector<bool>& inlier_mask,
                       const BundleAdjustmentOptions& options);

double bundle_adjustment(
    FeatureManager& feature_manager, const std::vector<size_t>& free_image_ids,
    const std::vector<size_t>& fixed_image_ids, const std::vector<size_t>& fixed_x_image_ids,
    const BundleAdjustmentOptions& options, std::unordered_map<size_t, double>& point3D_errors,
    const std::unordered_map<size_t, Eigen::Vector3d>& rotation_constraints =
        std::unordered_map<size_t, Eigen::Vector3d>(),
    const std::set<size_t>& gcp_ids = std::set<size_t>());

#endif  // MAVMAP_SRC_BASE3D_BUNDLE_ADJUSTMENT_H_
