// TEST DOUBLE of MAVMAP's FeatureManager: only the public containers the bundle-adjustment
// translation unit reads and writes (reference src/fm/feature_management.h:189-230), with the
// same names and key conventions (1-based size_t ids; camera_params carries the model code in
// its last slot). Track management etc. is not reproduced — tests fill the maps directly.
#ifndef MAVBA_TEST_FEATURE_MANAGEMENT_H_
#define MAVBA_TEST_FEATURE_MANAGEMENT_H_
#include <set>
#include <unordered_map>
#include <vector>

#include <Eigen/Core>

class FeatureManager {
 public:
  std::unordered_map<size_t, Eigen::Vector3d> points3D;
  std::unordered_map<size_t, Eigen::Vector2d> points2D;
  std::unordered_map<size_t, size_t> point2D_to_point3D;
  std::unordered_map<size_t, size_t> point2D_to_image;
  std::unordered_map<size_t, std::vector<size_t> > image_to_points2D;
  std::unordered_map<size_t, std::vector<size_t> > point3D_to_points2D;
  std::unordered_map<size_t, Eigen::Vector3d> rvecs;
  std::unordered_map<size_t, Eigen::Vector3d> tvecs;
  std::unordered_map<size_t, size_t> image_to_camera;
  std::unordered_map<size_t, std::vector<double> > camera_params;
};
#endif
