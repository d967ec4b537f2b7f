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
// Ids are the caller's (1-based size_t ids of the FeatureManager are fine); arrays grow to the largest id seen.
//
// Two routes to the solve:
//  * host flatten (small calls: a local-BA window is cheaper to flatten on the host than to launch kernels for) ->
//    mavba_solve() on the flat problem;
//  * DEVICE-RESIDENT scene (big calls): the 2-D points (pixel, link), the 3-D points and their alive flags live in HBM and
//    follow the host mirror by dirty ranges; a call uploads 4 bytes per candidate 2-D point of the selected images, the
//    selection (track length inside the image set, min_track_len, first-appearance point numbering - the rules of
//    bundle_adjustment.cc:228-387) runs in kernels, the session is built from the arrays where they are
//    (device_setup.hip), and the refined points are scattered back into the resident arrays. Same flat problem, element
//    for element, as the host route (tests compare the two solves bit for bit).
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

  // ---- device residency ----
  struct Dev {
    int device = -1;
    hipStream_t st = nullptr;
    DevBuf<double2> xy;            // [2-D point id]
    DevBuf<int> link;              // [2-D point id] 3-D point id or -1
    DevBuf<double> xyz;            // [3-D point id][3]
    DevBuf<unsigned char> alive;   // [3-D point id]
    size_t cap2 = 0, cap3 = 0;     // allocated ids
    ~Dev() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
  } dev;
  // host ranges changed since the device last saw them ([lo, hi), lo >= hi: clean)
  size_t dirty2_lo = 0, dirty2_hi = 0, dirty3_lo = 0, dirty3_hi = 0;
  bool dirty2_any = false, dirty3_any = false;
  void touch2(size_t id) { if (!dirty2_any) { dirty2_lo = id; dirty2_hi = id + 1; dirty2_any = true; } else { dirty2_lo = std::min(dirty2_lo, id); dirty2_hi = std::max(dirty2_hi, id + 1); } }
  void touch3(size_t id) { if (!dirty3_any) { dirty3_lo = id; dirty3_hi = id + 1; dirty3_any = true; } else { dirty3_lo = std::min(dirty3_lo, id); dirty3_hi = std::max(dirty3_hi, id + 1); } }
  void sync_device();
  int device_bundle_adjust(const long long* const lists[3], const int64_t counts[3], const long long* rot_images, const double* rot_rvecs,
                           int64_t n_rot, const mavba_scene_options& o, const mavba_options& options, mavba_result* res, std::vector<double>& perr);

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


// =====================================================================================================================
// Device-resident route
// =====================================================================================================================
namespace {
__global__ void k_sel_count(int T, const int* __restrict__ cand, const int* __restrict__ link, const unsigned char* __restrict__ alive,
                            int n2, int n3, unsigned* __restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int id2 = cand[t];
  const int id3 = id2 < n2 ? link[id2] : -1;
  if (id3 >= 0 && id3 < n3 && alive[id3]) atomicAdd(&count[id3], 1u);  // (integer count: order-independent)
}
__global__ void k_sel_flag(int T, const int* __restrict__ cand, const int* __restrict__ link, const unsigned char* __restrict__ alive,
                           int n2, int n3, const unsigned* __restrict__ count, unsigned min_track, unsigned* __restrict__ keep) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > T) return;
  unsigned k = 0u;
  if (t < T) {
    const int id2 = cand[t];
    const int id3 = id2 < n2 ? link[id2] : -1;
    k = (id3 >= 0 && id3 < n3 && alive[id3] && count[id3] >= min_track) ? 1u : 0u;
  }
  keep[t] = k;  // (keep[T] = 0: after the scan it holds the number of observations)
}
// kept candidate t -> observation scan[t]: pixel, slot (position of its image in the call's lists), 3-D point id; the
// 3-D point's first observation (lowest position) numbers the points by first appearance
__global__ void k_sel_compact(int T, const int* __restrict__ cand, const int* __restrict__ link, const unsigned char* __restrict__ alive,
                              int n2, int n3, const unsigned* __restrict__ count, unsigned min_track, const unsigned* __restrict__ scan,
                              const int* __restrict__ cand_off, int S, const double2* __restrict__ xy, double2* __restrict__ o_uv,
                              int* __restrict__ o_slot, int* __restrict__ o_id3, unsigned* __restrict__ first_pos) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int id2 = cand[t];
  const int id3 = id2 < n2 ? link[id2] : -1;
  if (!(id3 >= 0 && id3 < n3 && alive[id3] && count[id3] >= min_track)) return;
  const unsigned pos = scan[t];
  int lo = 0, hi = S;  // slot: cand_off[slot] <= t < cand_off[slot + 1]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cand_off[mid] <= t) lo = mid; else hi = mid; }
  o_uv[pos] = xy[id2];
  o_slot[pos] = lo;
  o_id3[pos] = id3;
  atomicMin(&first_pos[id3], pos);  // (a minimum: order-independent)
}
__global__ void k_sel_first(int NO, const int* __restrict__ o_id3, const unsigned* __restrict__ first_pos, unsigned* __restrict__ isfirst) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a > NO) return;
  isfirst[a] = a < NO && first_pos[o_id3[a]] == (unsigned)a ? 1u : 0u;
}
__global__ void k_sel_points(int NO, const int* __restrict__ o_id3, const unsigned* __restrict__ first_pos, const unsigned* __restrict__ pscan,
                             const double* __restrict__ xyz, int* __restrict__ o_pt, int* __restrict__ point_id3, double* __restrict__ pts) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= NO) return;
  const int id3 = o_id3[a];
  const unsigned fp = first_pos[id3];
  const int pt = (int)pscan[fp];
  o_pt[a] = pt;
  if (fp == (unsigned)a) {
    point_id3[pt] = id3;
    pts[3 * (size_t)pt] = xyz[3 * (size_t)id3]; pts[3 * (size_t)pt + 1] = xyz[3 * (size_t)id3 + 1]; pts[3 * (size_t)pt + 2] = xyz[3 * (size_t)id3 + 2];
  }
}
__global__ void k_sel_gather(int n, const unsigned* __restrict__ scan, const int* __restrict__ at, unsigned* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = scan[at[i]];
}
__global__ void k_sel_images(int NO, const int* __restrict__ o_slot, const int* __restrict__ slot_img, int* __restrict__ o_img) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < NO) o_img[a] = slot_img[o_slot[a]];
}
// refined points (internal order of the session) back into the resident 3-D points
__global__ void k_scene_scatter_points(int NP, const int* __restrict__ pt_orig, const int* __restrict__ point_id3, const double* __restrict__ pts,
                                       double* __restrict__ xyz) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= NP) return;
  const int id3 = point_id3[pt_orig[q]];
  xyz[3 * (size_t)id3] = pts[3 * (size_t)q]; xyz[3 * (size_t)id3 + 1] = pts[3 * (size_t)q + 1]; xyz[3 * (size_t)id3 + 2] = pts[3 * (size_t)q + 2];
}
}  // namespace

// Host mirror -> HBM: whole arrays when they grew past the device's capacity, else the range touched since the last call.
void mavba_scene::sync_device() {
  if (dev.device < 0) {
    HIP_OK(hipGetDevice(&dev.device));
    HIP_OK(hipStreamCreate(&dev.st));
  }
  hipStream_t st = dev.st;
  const size_t n2 = p2d_set.size(), n3 = p3d_alive.size();
  if (n2 >= ((size_t)1 << 31) || n3 >= ((size_t)1 << 31)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "ids beyond 2^31 on the device-resident route");
  std::vector<int> l32;
  auto push2 = [&](size_t lo, size_t hi) {
    if (hi <= lo) return;
    HIP_OK(copy_h2d_staged(dev.xy.p + lo, xy.data() + 2 * lo, (hi - lo) * 16, st));  // (the scene's vectors are pageable and grow)
    l32.resize(hi - lo);
    for (size_t i = lo; i < hi; ++i) l32[i - lo] = link[i] < 0 || link[i] >= ((long long)1 << 31) ? -1 : (int)link[i];
    HIP_OK(copy_h2d_staged(dev.link.p + lo, l32.data(), (hi - lo) * 4, st));
    HIP_OK(hipStreamSynchronize(st));  // (l32 is a temporary)
    release_staged(st);
  };
  auto push3 = [&](size_t lo, size_t hi) {
    if (hi <= lo) return;
    HIP_OK(copy_h2d_staged(dev.xyz.p + 3 * lo, xyz.data() + 3 * lo, (hi - lo) * 24, st));
    HIP_OK(copy_h2d_staged(dev.alive.p + lo, p3d_alive.data() + lo, hi - lo, st));
  };
  if (n2 > dev.cap2) {
    dev.cap2 = n2 + n2 / 2 + 1024;
    dev.xy.alloc(dev.cap2); dev.link.alloc(dev.cap2);
    push2(0, n2);
  } else if (dirty2_any) {
    push2(dirty2_lo, std::min(dirty2_hi, n2));
  }
  if (n3 > dev.cap3) {
    dev.cap3 = n3 + n3 / 2 + 1024;
    dev.xyz.alloc(dev.cap3 * 3); dev.alive.alloc(dev.cap3);
    push3(0, n3);
  } else if (dirty3_any) {
    push3(dirty3_lo, std::min(dirty3_hi, n3));
  }
  dirty2_any = dirty3_any = false;
  HIP_OK(hipStreamSynchronize(st));
  release_staged(st);
}

// One bundle_adjustment() call on the device-resident scene (no GCPs: the caller takes the host route for those).
int mavba_scene::device_bundle_adjust(const long long* const lists[3], const int64_t counts[3], const long long* rot_images,
                                      const double* rot_rvecs, int64_t n_rot, const mavba_scene_options& o, const mavba_options& options,
                                      mavba_result* res, std::vector<double>& perr) {
  const double t_begin = now_s();
  auto image_of = [&](long long id) -> Image& {
    if (id < 0 || (size_t)id >= images.size() || !images[(size_t)id].set) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown image id");
    return images[(size_t)id];
  };
  auto rot_of = [&](long long image_id) -> const double* {
    for (int64_t q = 0; q < n_rot; ++q) if (rot_images[q] == image_id) return rot_rvecs + 3 * q;
    throw Failure(MAVBA_ERR_BAD_INDEX, "no rotation constraint for an image that needs one");
  };
  // the scene lives on the device of its first big call; later calls run there whatever the calling thread's device is
  struct Restore { int d = -1; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore;
  if (dev.device >= 0) {
    int cur = -1;
    HIP_OK(hipGetDevice(&cur));
    if (cur != dev.device) { HIP_OK(hipSetDevice(dev.device)); restore.d = cur; }
  }
  sync_device();
  hipStream_t st = dev.st;
  const int n2 = (int)p2d_set.size(), n3 = (int)p3d_alive.size();
  // candidates: the 2-D points of the listed images, list order FREE, FIXED, FIXED_X (:511-533), insertion order inside
  std::vector<int> cand_off;
  std::vector<long long> slot_image;
  std::vector<int> slot_list;
  long long T = 0;
  for (int l = 0; l < 3; ++l)
    for (int64_t e = 0; e < counts[l]; ++e) {
      cand_off.push_back((int)T);
      slot_image.push_back(lists[l][e]); slot_list.push_back(l);
      T += (long long)image_of(lists[l][e]).p2d.size();
    }
  cand_off.push_back((int)T);
  const int S = (int)slot_image.size();
  if (T >= ((long long)1 << 31) - 4096) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "more than 2^31 candidate 2-D points");
  PinnedBuf<int> cand((size_t)std::max<long long>(T, 1));
  parallel_ranges(S, [&](long long s0, long long s1) {
    for (long long sl = s0; sl < s1; ++sl) {
      const std::vector<long long>& p2d = images[(size_t)slot_image[sl]].p2d;
      int* out = cand.data() + cand_off[sl];
      for (size_t i = 0; i < p2d.size(); ++i) out[i] = (int)p2d[i];
    }
  }, 8);
  DevBuf<int> d_cand, d_cand_off, o_slot, o_id3, o_pt, o_img, point_id3, d_slot_img, d_at;
  DevBuf<unsigned> count, first_pos, keep, isfirst, scratch, gathered;
  DevBuf<double2> o_uv;
  DevBuf<double> pts;
  d_cand.upload_pinned(cand.data(), (size_t)T, st);
  d_cand_off.upload(cand_off, st);
  count.alloc((size_t)std::max(n3, 1)); first_pos.alloc((size_t)std::max(n3, 1));
  count.zero(st);
  HIP_OK(hipMemsetAsync(first_pos.p, 0xFF, (size_t)std::max(n3, 1) * 4, st));
  keep.alloc((size_t)T + 1);
  scratch.alloc((size_t)device_scan_scratch(T + 1) + 8);
  const unsigned min_track = (unsigned)o.min_track_len;
  const int Ti = (int)T;
  if (Ti > 0) hipLaunchKernelGGL(k_sel_count, dim3((Ti + 255) / 256), dim3(256), 0, st, Ti, d_cand.p, dev.link.p, dev.alive.p, n2, n3, count.p);
  hipLaunchKernelGGL(k_sel_flag, dim3((Ti + 256) / 256), dim3(256), 0, st, Ti, d_cand.p, dev.link.p, dev.alive.p, n2, n3, count.p, min_track, keep.p);
  device_scan_exclusive(st, keep.p, T + 1, scratch.p);
  // observations per slot (the "> 1 residual" rule and the registration order need them) and in total
  gathered.alloc((size_t)S + 1);
  hipLaunchKernelGGL(k_sel_gather, dim3((S + 1 + 255) / 256), dim3(256), 0, st, S + 1, keep.p, d_cand_off.p, gathered.p);
  std::vector<unsigned> slot_start((size_t)S + 1);
  HIP_OK(copy_d2h_staged_sync(slot_start.data(), gathered.p, ((size_t)S + 1) * 4, st));
  release_staged(st);
  const int NO = (int)slot_start[S];
  if (NO == 0) return MAVBA_ERR_NEEDS_REBUILD;  // (nothing to optimise: the host route produces the reference's NaN / warnings)
  o_uv.alloc((size_t)NO); o_slot.alloc((size_t)NO); o_id3.alloc((size_t)NO); o_pt.alloc((size_t)NO); o_img.alloc((size_t)NO);
  isfirst.alloc((size_t)NO + 1);
  DevBuf<unsigned> scratch2;
  scratch2.alloc((size_t)device_scan_scratch(NO + 1) + 8);
  hipLaunchKernelGGL(k_sel_compact, dim3((Ti + 255) / 256), dim3(256), 0, st, Ti, d_cand.p, dev.link.p, dev.alive.p, n2, n3, count.p, min_track, keep.p,
                     d_cand_off.p, S, dev.xy.p, o_uv.p, o_slot.p, o_id3.p, first_pos.p);
  hipLaunchKernelGGL(k_sel_first, dim3((NO + 256) / 256), dim3(256), 0, st, NO, o_id3.p, first_pos.p, isfirst.p);
  device_scan_exclusive(st, isfirst.p, (long long)NO + 1, scratch2.p);
  unsigned np_u = 0;
  HIP_OK(hipMemcpyAsync(&np_u, isfirst.p + NO, 4, hipMemcpyDeviceToHost, st));
  // ---- the small, per-image part of the flat problem on the host, exactly as flatten() registers it ----
  image_ids.clear(); camera_ids.clear(); point_ids.clear();
  poses.clear(); intrinsics.clear(); prior_rvec.clear();
  pose_const.clear(); intr_const.clear(); point_const.clear();
  image_camera.clear(); camera_model.clear(); prior_image.clear();
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
  const uint8_t state_mask[3] = {0, (uint8_t)MAVBA_CONST_POSE, (uint8_t)MAVBA_CONST_TX};
  std::vector<int> slot_img((size_t)S, 0);
  for (int sl = 0; sl < S; ++sl) {
    const long long image_id = slot_image[sl];
    const size_t num_residuals = slot_start[sl + 1] - slot_start[sl];
    int32_t img = image_index[(size_t)image_id];
    if (num_residuals > 0 && img < 0) img = register_image(image_id, 0, false);
    slot_img[sl] = std::max(img, 0);
    if (num_residuals > 1) {  // :361
      pose_const[(size_t)img] |= state_mask[slot_list[sl]];
      if (!o.refine_camera_params) intr_const[(size_t)image_camera[(size_t)img]] = 1;
    }
  }
  if (o.constrain_rotation)
    for (int64_t e = 0; e < counts[0]; ++e) {
      const long long image_id = lists[0][e];
      const double* rv = rot_of(image_id);
      int32_t img = image_index[(size_t)image_id];
      if (img < 0) img = register_image(image_id, (uint8_t)(MAVBA_CONST_TX | MAVBA_CONST_TY | MAVBA_CONST_TZ), true);
      prior_image.push_back(img);
      prior_rvec.insert(prior_rvec.end(), rv, rv + 3);
    }
  d_slot_img.upload(slot_img, st);
  HIP_OK(hipStreamSynchronize(st));
  const int NPf = (int)np_u;
  point_id3.alloc((size_t)std::max(NPf, 1)); pts.alloc((size_t)std::max(NPf, 1) * 3);
  hipLaunchKernelGGL(k_sel_points, dim3((NO + 255) / 256), dim3(256), 0, st, NO, o_id3.p, first_pos.p, isfirst.p, dev.xyz.p, o_pt.p, point_id3.p, pts.p);
  hipLaunchKernelGGL(k_sel_images, dim3((NO + 255) / 256), dim3(256), 0, st, NO, o_slot.p, d_slot_img.p, o_img.p);
  std::vector<int> h_point_id3((size_t)NPf);
  if (NPf) HIP_OK(copy_d2h_staged_sync(h_point_id3.data(), point_id3.p, (size_t)NPf * 4, st));
  HIP_OK(hipStreamSynchronize(st));
  release_staged(st);
  point_ids.assign(h_point_id3.begin(), h_point_id3.end());
  // ---- the session, built from the arrays where they are ----
  mavba_problem P;
  std::memset(&P, 0, sizeof(P));
  P.num_images = (int32_t)image_ids.size(); P.num_cameras = (int32_t)camera_ids.size(); P.num_points = NPf; P.num_obs = NO;
  P.poses = poses.data(); P.pose_const = pose_const.data(); P.image_camera = image_camera.data();
  P.intrinsics = intrinsics.data(); P.camera_model = camera_model.data(); P.intr_const = intr_const.data();
  P.num_rot_priors = (int32_t)prior_image.size(); P.rot_prior_image = prior_image.data(); P.rot_prior_rvec = prior_rvec.data();
  P.rot_prior_weight = o.constrain_rotation_weight;
  DeviceRaw raw{reinterpret_cast<const double*>(o_uv.p), o_img.p, o_pt.p, pts.p};
  std::unique_ptr<mavba_session> sess(new mavba_session());
  sess->opt = options;
  sess->device = dev.device;
  HIP_OK(stream_acquire(&sess->st));
  sess->build(&P, &raw);
  const double t_built = now_s();
  int done = 0;
  sess->iterate(options.max_num_iterations + 1, &done);
  sess->fill_result(res);
  res->setup_seconds = t_built - t_begin;  // (selection + set-up: everything before the first iteration)
  const int term = sess->termination;
  points.assign((size_t)NPf * 3, 0.0);
  if (term != MAVBA_TERM_NUMERICAL_FAILURE) {
    // cameras to the mirror through the host, points into the resident array AND the mirror
    const int rc = mavba_session_get_params(sess.get(), poses.data(), intrinsics.data(), points.data());
    if (rc != MAVBA_OK) throw Failure(rc, g_last_error);
    hipLaunchKernelGGL(k_scene_scatter_points, dim3((NPf + 255) / 256), dim3(256), 0, sess->st, NPf, sess->d_pt_orig.p, point_id3.p, sess->d_points.p, dev.xyz.p);
    for (size_t i = 0; i < image_ids.size(); ++i)
      for (int k = 0; k < 6; ++k) images[(size_t)image_ids[i]].pose[k] = poses[6 * i + k];
    for (size_t c = 0; c < camera_ids.size(); ++c) {
      Camera& cam = cameras[(size_t)camera_ids[c]];
      for (int k = 0; k < model_k(cam.model); ++k) cam.p[k] = intrinsics[MAVBA_MAX_INTR * c + k];
    }
    parallel_ranges(NPf, [&](long long p0, long long p1) {
      for (long long p = p0; p < p1; ++p)
        for (int k = 0; k < 3; ++k) xyz[3 * (size_t)point_ids[p] + k] = points[3 * p + k];
    });
  }
  perr.assign((size_t)std::max(NPf, 1), 0.0);
  if (options.update_point_errors) {
    if (term == MAVBA_TERM_NUMERICAL_FAILURE) sess->restore_initial_params();
    sess->point_errors(perr.data());
  }
  sess->sync();
  return MAVBA_OK;
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
  s->touch2((size_t)point2D_id);
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
    s->touch2(id);
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
  s->touch3((size_t)point3D_id);
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_link(mavba_scene* s, int64_t point2D_id, int64_t point3D_id) {
  SCENE_TRY
  if (!s) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (point2D_id < 0 || (size_t)point2D_id >= s->p2d_set.size() || !s->p2d_set[(size_t)point2D_id]) throw Failure(MAVBA_ERR_BAD_INDEX, "unknown 2-D point id");
  s->link[(size_t)point2D_id] = point3D_id < 0 ? -1 : point3D_id;
  s->touch2((size_t)point2D_id);
  return MAVBA_OK;
  SCENE_CATCH
}
int mavba_scene_delete_point3d(mavba_scene* s, int64_t point3D_id) {
  SCENE_TRY
  if (!s) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (s->has_point3D(point3D_id)) { s->p3d_alive[(size_t)point3D_id] = 0; s->touch3((size_t)point3D_id); }  // its 2-D points read as unmatched from now on
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
  static thread_local std::vector<double> perr;
  // Big calls without ground control points go through the device-resident scene (MAVBA_SCENE=host | device forces a route)
  if (s && so && n_gcp == 0 && so->min_track_len >= 2 && n_fixed * 6 + n_fixed_x >= 7 && !so->constrain_rotation && multi_gpu_ranks() <= 1) {
    const char* route = std::getenv("MAVBA_SCENE");
    long long cands = 0;
    const int64_t* ls[3] = {free_ids, fixed_ids, fixed_x_ids};
    const int64_t cn[3] = {n_free, n_fixed, n_fixed_x};
    bool known = true;
    for (int l = 0; l < 3 && known; ++l)
      for (int64_t e = 0; e < cn[l]; ++e) {
        const int64_t id = ls[l][e];
        if (id < 0 || (size_t)id >= s->images.size() || !s->images[(size_t)id].set) { known = false; break; }
        cands += (long long)s->images[(size_t)id].p2d.size();
      }
    const bool want = route ? std::string(route) == "device" : cands >= 200000;
    if (known && want && !(route && std::string(route) == "host") && mavba_device_count() > 0) {
      mavba_result local;
      mavba_result* res = result ? result : &local;
      try {
        const long long* lists[3] = {reinterpret_cast<const long long*>(free_ids), reinterpret_cast<const long long*>(fixed_ids),
                                     reinterpret_cast<const long long*>(fixed_x_ids)};
        const int rc = s->device_bundle_adjust(lists, cn, reinterpret_cast<const long long*>(rot_image_ids), rot_rvecs, n_rot, *so, *options, res, perr);
        if (rc == MAVBA_OK) {
          if (final_cost_px) *final_cost_px = std::sqrt(res->final_cost / (double)res->num_residuals);
          if (error_point_ids) *error_point_ids = reinterpret_cast<const int64_t*>(s->point_ids.data());
          if (error_values) *error_values = options->update_point_errors ? perr.data() : nullptr;
          if (num_errors) *num_errors = options->update_point_errors ? (int64_t)s->point_ids.size() : 0;
          return MAVBA_OK;
        }
        // (MAVBA_ERR_NEEDS_REBUILD: nothing selected - the host route below reproduces the reference's behaviour for that)
      }
      catch (const Failure& f) { g_last_error = f.what(); return f.code; }
      catch (const std::bad_alloc&) { g_last_error = "host out of memory"; return MAVBA_ERR_OUT_OF_MEMORY; }
      catch (const std::exception& e) { g_last_error = e.what(); return MAVBA_ERR_HIP; }
    }
  }
  mavba_problem P;
  int rc = mavba_scene_flatten(s, free_ids, n_free, fixed_ids, n_fixed, fixed_x_ids, n_fixed_x, gcp_ids, n_gcp, rot_image_ids, rot_rvecs,
                               n_rot, so, &P, nullptr, nullptr, nullptr);
  if (rc != MAVBA_OK) return rc;
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
  for (size_t p = 0; p < s->point_ids.size(); ++p) {
    for (int k = 0; k < 3; ++k) s->xyz[3 * (size_t)s->point_ids[p] + k] = s->points[3 * p + k];
    s->touch3((size_t)s->point_ids[p]);  // (the resident copy follows at the next device call)
  }
  if (final_cost_px) *final_cost_px = std::sqrt(res->final_cost / (double)res->num_residuals);  // bundle_adjustment.cc:610 (NaN for no residuals)
  if (error_point_ids) *error_point_ids = reinterpret_cast<const int64_t*>(s->point_ids.data());
  if (error_values) *error_values = options->update_point_errors ? perr.data() : nullptr;
  if (num_errors) *num_errors = options->update_point_errors ? (int64_t)s->point_ids.size() : 0;
  return MAVBA_OK;
}

}  // extern "C"
