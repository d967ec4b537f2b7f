// scene.hip - an incremental, flat mirror of what bundle adjustment needs from MAVMAP's FeatureManager.
//
// The reference re-walks the FeatureManager's hash maps on every bundle_adjustment() call
// (src/base3d/bundle_adjustment.cc:228-387: ~3 look-ups per 2-D point, 0.4 s for a 2 M-observation global BA, and local
// BA runs after every image, src/mapper.cc:1120-1135). A mavba_scene receives the same information as DELTAS, at the
// moment the FeatureManager changes (one call per add_point2D / add_point3D / correspondence / delete_point3D /
// set_pose), keeps it in dense arrays indexed by the FeatureManager's own ids, and builds the
// flat problem of a BA call from those arrays without hashing: which observations enter, in which order, which
// blocks are constant - the rules of bundle_adjustment.cc:228-549, identical to shim/base3d/bundle_adjustment.cc
// (tests compare the two element for element).
//
// Host code only: the solve itself is mavba_solve() on the flat problem. Ids are the caller's (1-based size_t ids of
// the FeatureManager are fine); arrays grow to the largest id seen.
#include "session.h"

using namespace mavba;

struct mavba_scene {
  struct Camera { int model = 0; double p[MAVBA_MAX_INTR] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; bool set = false; };
  struct Image { long long camera = -1; double pose[6] = {0, 0, 0, 0, 0, 0}; std::vector<long long> p2d; bool set = false; };
  std::vector<Camera> cameras;
  std::vector<Image> images;
  std::vector<double> xy;          // [point2D id][2]
  std::vector<long long> link;     // point2D id -> point3D id, -1 = none
  std::vector<unsigned char> p2d_set;
  std::vector<double> xyz;         // [point3D id][3]
  std::vector<unsigned char> p3d_alive;

  // ---- the flat problem of the last flatten() (views handed out through mavba_problem) ----
  std::vector<long long> image_ids, camera_ids, point_ids;  // flat index -> caller's id
  std::vector<double> poses, intrinsics, points, obs_uv, prior_rvec;
  std::vector<uint8_t> pose_const, intr_const, point_const;
  std::vector<int32_t> image_camera, camera_model, obs_image, obs_point, prior_image;
  std::vector<uint32_t> count;     // scratch: observations of a point inside the selected image set
  std::vector<int32_t> point_index, image_index, camera_index;  // scratch: caller's id -> flat index, -1

  template <class V> static void grow(V& v, long long id, size_t width = 1) {
    if ((size_t)(id + 1) * width > v.size()) v.resize((size_t)(id + 1) * width + (size_t)(id + 1) * width / 2);
  }
  bool has_point3D(long long id) const { return id >= 0 && (size_t)id < p3d_alive.size() && p3d_alive[(size_t)id]; }

  void flatten(const long long* const lists[3], const int64_t counts[3], const long long* gcp, int64_t n_gcp,
               const long long* rot_images, const double* rot_rvecs, int64_t n_rot, const mavba_scene_options& o,
               mavba_problem* P);
};

namespace {
void check_id(long long id) { if (id < 0 || id > (1ll << 40)) throw Failure(MAVBA_ERR_BAD_INDEX, "id out of range"); }

// angle-axis <-> rotation matrix (row-major), as the shim's helpers (reference src/base3d/projection.cc:12-23 and Eigen's
// AngleAxisd(matrix)): only used by the rotation-prior pre-rotation
void rot_from_rvec(const double* w, double* R) {
  double angle = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double ax[3] = {0, 0, 1};
  if (angle < std::numeric_limits<double>::epsilon()) angle = 0; else for (int i = 0; i < 3; ++i) ax[i] = w[i] / angle;
  const double c = std::cos(angle), s = std::sin(angle), t = 1 - c;
  R[0] = c + t * ax[0] * ax[0];         R[1] = t * ax[0] * ax[1] - s * ax[2]; R[2] = t * ax[0] * ax[2] + s * ax[1];
  R[3] = t * ax[0] * ax[1] + s * ax[2]; R[4] = c + t * ax[1] * ax[1];         R[5] = t * ax[1] * ax[2] - s * ax[0];
  R[6] = t * ax[0] * ax[2] - s * ax[1]; R[7] = t * ax[1] * ax[2] + s * ax[0]; R[8] = c + t * ax[2] * ax[2];
}
void rvec_from_rot(const double* R, double* w) {
  double q[4];
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
  const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 < std::numeric_limits<double>::min()) { w[0] = w[1] = w[2] = 0.0; return; }
  double angle = 2.0 * std::acos(std::min(std::max(q[0], -1.0), 1.0));
  const double inv = 1.0 / std::sqrt(n2);
  for (int a = 0; a < 3; ++a) w[a] = angle * q[1 + a] * inv;
}
void mul3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
}  // namespace

// The flat problem of one bundle_adjustment() call: lists[0..2] = free, fixed, fixed_x image ids.
void mavba_scene::flatten(const long long* const lists[3], const int64_t counts[3], const long long* gcp, int64_t n_gcp,
                          const long long* rot_images, const double* rot_rvecs, int64_t n_rot, const mavba_scene_options& o,
                          mavba_problem* P) {
  const int64_t num_fixed_params = counts[1] * 6 + counts[2] + n_gcp * 3;
  if (num_fixed_params < 7)  // bundle_adjustment.cc:459-466
    throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "At least 7 parameters should be set as fixed to avoid datum defects resulting in a singular Jacobian.");
  if (o.min_track_len < 2)   // :468-471
    throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "Minimum track length must be >= 2 in order build valid bundle adjustment problem.");
  auto image_of = [&](long long id) -> Image& {
    if (id < 0 || (size_t)id >= images.size() || !images[(size_t)id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown image id");
    return images[(size_t)id];
  };
  auto rot_of = [&](long long image_id) -> const double* {
    for (int64_t q = 0; q < n_rot; ++q) if (rot_images[q] == image_id) return rot_rvecs + 3 * q;
    throw Failure(MAVBA_ERR_BAD_INDEX, "no rotation constraint for an image that needs one");  // (.at() in the reference)
  };
  // rotation priors: first rotate EVERY pose and point so that the first fixed image agrees with its prior (:399-425)
  if (o.constrain_rotation) {
    if (counts[1] == 0) throw Failure(MAVBA_ERR_BAD_INDEX, "constrain_rotation needs a fixed image");
    const long long ref_id = lists[1][0];
    double R_fm[9], R_c[9], R_fm_t[9], S[9], St[9];
    rot_from_rvec(image_of(ref_id).pose, R_fm);
    rot_from_rvec(rot_of(ref_id), R_c);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R_fm_t[i * 3 + j] = R_fm[j * 3 + i];
    mul3(R_fm_t, R_c, S);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) St[i * 3 + j] = S[j * 3 + i];
    for (Image& im : images) {
      if (!im.set) continue;
      double R[9], Rn[9];
      rot_from_rvec(im.pose, R);
      mul3(R, St, Rn);
      rvec_from_rot(Rn, im.pose);
    }
    for (size_t p = 0; p < p3d_alive.size(); ++p) {
      if (!p3d_alive[p]) continue;
      double* X = &xyz[3 * p];
      const double x = X[0], y = X[1], z = X[2];
      X[0] = S[0] * x + S[1] * y + S[2] * z; X[1] = S[3] * x + S[4] * y + S[5] * z; X[2] = S[6] * x + S[7] * y + S[8] * z;
    }
  }
  // observations of every 3-D point inside the selected image set (:228-286; an image listed twice counts twice)
  count.assign(p3d_alive.size(), 0);
  for (int l = 0; l < 3; ++l)
    for (int64_t e = 0; e < counts[l]; ++e)
      for (long long id2 : image_of(lists[l][e]).p2d) { const long long id3 = link[(size_t)id2]; if (has_point3D(id3)) count[(size_t)id3]++; }
  image_ids.clear(); camera_ids.clear(); point_ids.clear();
  poses.clear(); intrinsics.clear(); points.clear(); obs_uv.clear(); prior_rvec.clear();
  pose_const.clear(); intr_const.clear(); point_const.clear();
  image_camera.clear(); camera_model.clear(); obs_image.clear(); obs_point.clear(); prior_image.clear();
  point_index.assign(p3d_alive.size(), -1);
  image_index.assign(images.size(), -1);
  camera_index.assign(cameras.size(), -1);
  auto register_image = [&](long long image_id, uint8_t initial_const, bool intr_const_if_new) -> int32_t {
    Image& im = image_of(image_id);
    if (im.camera < 0 || (size_t)im.camera >= cameras.size() || !cameras[(size_t)im.camera].set) throw Failure(MAVBA_ERR_BAD_INDEX, "image without a camera");
    const Camera& cam = cameras[(size_t)im.camera];
    int32_t& ic = camera_index[(size_t)im.camera];
    if (ic < 0) {
      ic = (int32_t)camera_ids.size();
      camera_ids.push_back(im.camera);
      camera_model.push_back(cam.model);
      intr_const.push_back(intr_const_if_new ? 1 : 0);
      intrinsics.insert(intrinsics.end(), cam.p, cam.p + MAVBA_MAX_INTR);
    }
    const int32_t img = (int32_t)image_ids.size();
    image_index[(size_t)image_id] = img;
    image_ids.push_back(image_id);
    image_camera.push_back(ic);
    pose_const.push_back(initial_const);
    poses.insert(poses.end(), im.pose, im.pose + 6);
    return img;
  };
  // residual-block order FREE, FIXED, FIXED_X (:511-533); inside an image its 2-D points in insertion order
  const uint8_t state_mask[3] = {0, (uint8_t)MAVBA_CONST_POSE, (uint8_t)MAVBA_CONST_TX};
  for (int l = 0; l < 3; ++l)
    for (int64_t e = 0; e < counts[l]; ++e) {
      const long long image_id = lists[l][e];
      const Image& im = image_of(image_id);
      size_t num_residuals = 0;
      int32_t img = image_index[(size_t)image_id];  // (an id listed twice: one set of blocks, observations added again)
      for (long long id2 : im.p2d) {
        const long long id3 = link[(size_t)id2];
        if (!has_point3D(id3) || count[(size_t)id3] < (uint32_t)o.min_track_len) continue;  // :330
        if (img < 0) img = register_image(image_id, 0, false);
        int32_t& ip = point_index[(size_t)id3];
        if (ip < 0) {
          ip = (int32_t)point_ids.size();
          point_ids.push_back(id3);
          points.insert(points.end(), &xyz[3 * (size_t)id3], &xyz[3 * (size_t)id3] + 3);
          point_const.push_back(0);
        }
        obs_uv.push_back(xy[2 * (size_t)id2]); obs_uv.push_back(xy[2 * (size_t)id2 + 1]);
        obs_image.push_back(img);
        obs_point.push_back(ip);
        ++num_residuals;
      }
      if (num_residuals > 1) {  // :361
        pose_const[(size_t)img] |= state_mask[l];
        if (!o.refine_camera_params) intr_const[(size_t)image_camera[(size_t)img]] = 1;
      }
    }
  for (int64_t g = 0; g < n_gcp; ++g)  // :545-549
    if (gcp[g] >= 0 && (size_t)gcp[g] < point_index.size() && point_index[(size_t)gcp[g]] >= 0) point_const[(size_t)point_index[(size_t)gcp[g]]] = 1;
  // one rotation prior per FREE image (:428-444); an image without residual blocks still gets its prior
  if (o.constrain_rotation)
    for (int64_t e = 0; e < counts[0]; ++e) {
      const long long image_id = lists[0][e];
      const double* rv = rot_of(image_id);
      int32_t img = image_index[(size_t)image_id];
      if (img < 0) img = register_image(image_id, (uint8_t)(MAVBA_CONST_TX | MAVBA_CONST_TY | MAVBA_CONST_TZ), true);
      prior_image.push_back(img);
      prior_rvec.insert(prior_rvec.end(), rv, rv + 3);
    }
  std::memset(P, 0, sizeof(*P));
  P->num_images = (int32_t)image_ids.size(); P->num_cameras = (int32_t)camera_ids.size();
  P->num_points = (int32_t)point_ids.size(); P->num_obs = (int64_t)obs_image.size();
  P->poses = poses.data(); P->pose_const = pose_const.data(); P->image_camera = image_camera.data();
  P->intrinsics = intrinsics.data(); P->camera_model = camera_model.data(); P->intr_const = intr_const.data();
  P->points = points.data(); P->point_const = point_const.data();
  P->obs_uv = obs_uv.data(); P->obs_image = obs_image.data(); P->obs_point = obs_point.data();
  P->num_rot_priors = (int32_t)prior_image.size(); P->rot_prior_image = prior_image.data(); P->rot_prior_rvec = prior_rvec.data();
  P->rot_prior_weight = o.constrain_rotation_weight;
}

#define SCENE_TRY try {
#define SCENE_CATCH                                                                                      \
  }                                                                                                      \
  catch (const Failure& f) { g_last_error = f.what(); return f.code; }                                   \
  catch (const std::bad_alloc&) { g_last_error = "host out of memory"; return MAVBA_ERR_OUT_OF_MEMORY; } \
  catch (const std::exception& e) { g_last_error = e.what(); return MAVBA_ERR_HIP; }

extern "C" {

int mavba_scene_create(mavba_scene** out) {
  if (!out) { g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT; }
  SCENE_TRY
  *out = new mavba_scene();
  return MAVBA_OK;
  SCENE_CATCH
}
void mavba_scene_destroy(mavba_scene* s) { delete s; }

int mavba_scene_set_camera(mavba_scene* s, int64_t camera_id, int32_t model, const double* params) {
  SCENE_TRY
  if (!s || !params) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (model < 1 || model > 3) throw Failure(MAVBA_ERR_BAD_MODEL, "camera model code not in {1,2,3}");
  check_id(camera_id);
  mavba_scene::grow(s->cameras, camera_id);
  mavba_scene::Camera& c = s->cameras[(size_t)camera_id];
  c.model = model; c.set = true;
  for (int k = 0; k < MAVBA_MAX_INTR; ++k) c.p[k] = k < model_k(model) ? params[k] : 0.0;
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_set_image(mavba_scene* s, int64_t image_id, int64_t camera_id, const double* rvec, const double* tvec) {
  SCENE_TRY
  if (!s) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  check_id(image_id);
  mavba_scene::grow(s->images, image_id);
  mavba_scene::Image& im = s->images[(size_t)image_id];
  if (camera_id >= 0) im.camera = camera_id;
  if (rvec) for (int k = 0; k < 3; ++k) im.pose[k] = rvec[k];
  if (tvec) for (int k = 0; k < 3; ++k) im.pose[3 + k] = tvec[k];
  im.set = true;
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_add_point2d(mavba_scene* s, int64_t image_id, int64_t point2D_id, const double* xy) {
  SCENE_TRY
  if (!s || !xy) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  check_id(image_id); check_id(point2D_id);
  if ((size_t)image_id >= s->images.size() || !s->images[(size_t)image_id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown image id");
  mavba_scene::grow(s->xy, point2D_id, 2); mavba_scene::grow(s->p2d_set, point2D_id);
  if (s->link.size() < (size_t)point2D_id + 1) s->link.resize(((size_t)point2D_id + 1) * 3 / 2 + 1, -1);
  if (s->p2d_set[(size_t)point2D_id]) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "2-D point id added twice");
  s->p2d_set[(size_t)point2D_id] = 1;
  s->xy[2 * (size_t)point2D_id] = xy[0]; s->xy[2 * (size_t)point2D_id + 1] = xy[1];
  s->link[(size_t)point2D_id] = -1;
  s->images[(size_t)image_id].p2d.push_back(point2D_id);
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_add_points2d(mavba_scene* s, int64_t image_id, int64_t count, const int64_t* point2D_ids, const double* xy,
                             const int64_t* point3D_ids) {
  SCENE_TRY
  if (!s || count < 0 || (count > 0 && (!point2D_ids || !xy))) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  check_id(image_id);
  if ((size_t)image_id >= s->images.size() || !s->images[(size_t)image_id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown image id");
  long long top = -1;
  for (int64_t i = 0; i < count; ++i) { check_id(point2D_ids[i]); top = std::max<long long>(top, point2D_ids[i]); }
  if (top >= 0) { mavba_scene::grow(s->xy, top, 2); mavba_scene::grow(s->link, top); mavba_scene::grow(s->p2d_set, top); }
  // (nothing is changed before the whole call is known to be valid)
  for (int64_t i = 0; i < count; ++i)
    if (s->p2d_set[(size_t)point2D_ids[i]]) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "2-D point id added twice");
  {
    std::vector<int64_t> sorted(point2D_ids, point2D_ids + count);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "2-D point id twice in one call");
  }
  std::vector<long long>& list = s->images[(size_t)image_id].p2d;
  list.reserve(list.size() + (size_t)count);
  for (int64_t i = 0; i < count; ++i) {
    const size_t id = (size_t)point2D_ids[i];
    s->p2d_set[id] = 1;
    s->xy[2 * id] = xy[2 * i]; s->xy[2 * id + 1] = xy[2 * i + 1];
    s->link[id] = point3D_ids && point3D_ids[i] >= 0 ? (long long)point3D_ids[i] : -1;
    list.push_back((long long)id);
  }
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_set_point3d(mavba_scene* s, int64_t point3D_id, const double* xyz) {
  SCENE_TRY
  if (!s || !xyz) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  check_id(point3D_id);
  mavba_scene::grow(s->xyz, point3D_id, 3); mavba_scene::grow(s->p3d_alive, point3D_id);
  for (int k = 0; k < 3; ++k) s->xyz[3 * (size_t)point3D_id + k] = xyz[k];
  s->p3d_alive[(size_t)point3D_id] = 1;
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_link(mavba_scene* s, int64_t point2D_id, int64_t point3D_id) {
  SCENE_TRY
  if (!s) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (point2D_id < 0 || (size_t)point2D_id >= s->p2d_set.size() || !s->p2d_set[(size_t)point2D_id]) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown 2-D point id");
  s->link[(size_t)point2D_id] = point3D_id < 0 ? -1 : point3D_id;
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_delete_point3d(mavba_scene* s, int64_t point3D_id) {
  SCENE_TRY
  if (!s) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (s->has_point3D(point3D_id)) s->p3d_alive[(size_t)point3D_id] = 0;  // its 2-D points read as unmatched from now on
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_get_image(mavba_scene* s, int64_t image_id, double* rvec, double* tvec) {
  SCENE_TRY
  if (!s || image_id < 0 || (size_t)image_id >= s->images.size() || !s->images[(size_t)image_id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown image id");
  const mavba_scene::Image& im = s->images[(size_t)image_id];
  if (rvec) for (int k = 0; k < 3; ++k) rvec[k] = im.pose[k];
  if (tvec) for (int k = 0; k < 3; ++k) tvec[k] = im.pose[3 + k];
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_get_point3d(mavba_scene* s, int64_t point3D_id, double* xyz) {
  SCENE_TRY
  if (!s || !xyz || !s->has_point3D(point3D_id)) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown 3-D point id");
  for (int k = 0; k < 3; ++k) xyz[k] = s->xyz[3 * (size_t)point3D_id + k];
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_get_camera(mavba_scene* s, int64_t camera_id, int32_t* model, double* params) {
  SCENE_TRY
  if (!s || camera_id < 0 || (size_t)camera_id >= s->cameras.size() || !s->cameras[(size_t)camera_id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown camera id");
  const mavba_scene::Camera& c = s->cameras[(size_t)camera_id];
  if (model) *model = c.model;
  if (params) for (int k = 0; k < model_k(c.model); ++k) params[k] = c.p[k];
  return MAVBA_OK;
  SCENE_CATCH
}

int mavba_scene_flatten(mavba_scene* s, const int64_t* free_ids, int64_t n_free, const int64_t* fixed_ids, int64_t n_fixed,
                        const int64_t* fixed_x_ids, int64_t n_fixed_x, const int64_t* gcp_ids, int64_t n_gcp,
                        const int64_t* rot_image_ids, const double* rot_rvecs, int64_t n_rot, const mavba_scene_options* o,
                        mavba_problem* problem, const int64_t** image_ids, const int64_t** camera_ids, const int64_t** point_ids) {
  SCENE_TRY
  if (!s || !o || !problem) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  static_assert(sizeof(long long) == sizeof(int64_t), "ids are 64-bit");
  const long long* lists[3] = {reinterpret_cast<const long long*>(free_ids), reinterpret_cast<const long long*>(fixed_ids),
                               reinterpret_cast<const long long*>(fixed_x_ids)};
  const int64_t counts[3] = {n_free, n_fixed, n_fixed_x};
  s->flatten(lists, counts, reinterpret_cast<const long long*>(gcp_ids), n_gcp, reinterpret_cast<const long long*>(rot_image_ids),
             rot_rvecs, n_rot, *o, problem);
  if (image_ids) *image_ids = reinterpret_cast<const int64_t*>(s->image_ids.data());
  if (camera_ids) *camera_ids = reinterpret_cast<const int64_t*>(s->camera_ids.data());
  if (point_ids) *point_ids = reinterpret_cast<const int64_t*>(s->point_ids.data());
  return MAVBA_OK;
  SCENE_CATCH
}

int mavba_scene_bundle_adjust(mavba_scene* s, const int64_t* free_ids, int64_t n_free, const int64_t* fixed_ids, int64_t n_fixed,
                              const int64_t* fixed_x_ids, int64_t n_fixed_x, const int64_t* gcp_ids, int64_t n_gcp,
                              const int64_t* rot_image_ids, const double* rot_rvecs, int64_t n_rot, const mavba_scene_options* so,
                              const mavba_options* options, mavba_result* result, double* final_cost_px,
                              const int64_t** error_point_ids, const double** error_values, int64_t* num_errors) {
  if (!options) { g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT; }
  mavba_problem P;
  int rc = mavba_scene_flatten(s, free_ids, n_free, fixed_ids, n_fixed, fixed_x_ids, n_fixed_x, gcp_ids, n_gcp, rot_image_ids, rot_rvecs,
                               n_rot, so, &P, nullptr, nullptr, nullptr);
  if (rc != MAVBA_OK) return rc;
  static thread_local std::vector<double> perr;
  perr.assign((size_t)std::max(P.num_points, 1), 0.0);
  mavba_result local;
  mavba_result* res = result ? result : &local;
  rc = mavba_solve(&P, options, res, options->update_point_errors ? perr.data() : nullptr);
  if (rc != MAVBA_OK) return rc;
  // write the solution back into the mirror (the caller reads what it needs with the get calls / the flat views)
  for (size_t i = 0; i < s->image_ids.size(); ++i)
    for (int k = 0; k < 6; ++k) s->images[(size_t)s->image_ids[i]].pose[k] = s->poses[6 * i + k];
  for (size_t c = 0; c < s->camera_ids.size(); ++c) {
    mavba_scene::Camera& cam = s->cameras[(size_t)s->camera_ids[c]];
    for (int k = 0; k < model_k(cam.model); ++k) cam.p[k] = s->intrinsics[MAVBA_MAX_INTR * c + k];
  }
  for (size_t p = 0; p < s->point_ids.size(); ++p)
    for (int k = 0; k < 3; ++k) s->xyz[3 * (size_t)s->point_ids[p] + k] = s->points[3 * p + k];
  if (final_cost_px) *final_cost_px = std::sqrt(res->final_cost / (double)res->num_residuals);  // bundle_adjustment.cc:610 (NaN for no residuals)
  if (error_point_ids) *error_point_ids = reinterpret_cast<const int64_t*>(s->point_ids.data());
  if (error_values) *error_values = options->update_point_errors ? perr.data() : nullptr;
  if (num_errors) *num_errors = options->update_point_errors ? (int64_t)s->point_ids.size() : 0;
  return MAVBA_OK;
}

}  // extern "C"
