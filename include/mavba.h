/*
 * mavba.h — C ABI of the MI355X-native bundle-adjustment backend for MAVMAP.
 *
 * This is the drop-in boundary. The reference has exactly one function boundary
 * for this path:
 *
 *   double bundle_adjustment(FeatureManager&, free_ids, fixed_ids, fixed_x_ids,
 *                            options, point3D_errors, rotation_constraints,
 *                            gcp_ids)        reference src/base3d/bundle_adjustment.h:221-230
 *   double pose_refinement(rvec, tvec, camera_params, points2D, points3D,
 *                          inlier_mask, options)  reference src/base3d/bundle_adjustment.h:212-218
 *
 * Both are C++ functions over Eigen/STL containers; everything below them is
 * Ceres. A replacement translation unit (shim/base3d/bundle_adjustment.cc in
 * this repo) flattens FeatureManager into the plain arrays declared here,
 * calls mavba_solve()/mavba_pose_refine(), and writes the results back in
 * place. No exceptions, no C++ types and no torch types cross this ABI.
 *
 * All floating point is FP64 (the reference is `double` throughout). Indices
 * are 0-based int32 (the shim maps FeatureManager's 1-based size_t ids).
 *
 * The implementation behind this header is HIP for gfx950 only. There is no
 * CPU fallback: every compute entry point returns MAVBA_ERR_NO_DEVICE when no
 * GPU is present.
 */
#ifndef MAVBA_H_
#define MAVBA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAVBA_VERSION 1

/* Camera model codes — identical to the reference's CameraModel::code values
 * (reference src/base3d/camera_models.h:106-109, 165-168, 272-275). */
#define MAVBA_MODEL_PINHOLE 1 /* K = 4: fx fy cx cy                */
#define MAVBA_MODEL_OPENCV 2  /* K = 8: fx fy cx cy k1 k2 p1 p2    */
#define MAVBA_MODEL_CATA 3    /* K = 9: fx fy cx cy k1 k2 p1 p2 xi */
#define MAVBA_MAX_INTR 9      /* stride of the intrinsics table    */

/* Per-image constancy bitmask. The reference splits a pose into the Ceres
 * parameter blocks rvec(3), tx(1), ty(1), tz(1) so that BA_POSE_FIXED_X can
 * freeze tx alone (reference src/base3d/bundle_adjustment.h:125-128,
 * bundle_adjustment.cc:361-385). */
#define MAVBA_CONST_RVEC 1u
#define MAVBA_CONST_TX 2u
#define MAVBA_CONST_TY 4u
#define MAVBA_CONST_TZ 8u
#define MAVBA_CONST_POSE 15u /* BA_POSE_FIXED   */
/* BA_POSE_FIXED_X == MAVBA_CONST_TX, BA_POSE_FREE == 0 */

/* Return codes (negative = error). */
#define MAVBA_OK 0
#define MAVBA_ERR_INVALID_ARGUMENT (-1)
#define MAVBA_ERR_NO_DEVICE (-2)   /* no gfx950 device / HIP runtime failure at init */
#define MAVBA_ERR_HIP (-3)         /* a HIP call failed mid-flight (message via mavba_last_error) */
#define MAVBA_ERR_OUT_OF_MEMORY (-4)
#define MAVBA_ERR_BAD_INDEX (-5)   /* obs/image/camera index out of range */
#define MAVBA_ERR_BAD_MODEL (-6)   /* camera model code not in {1,2,3}     */
#define MAVBA_ERR_NEEDS_REBUILD (-7) /* mavba_session_filter_points: the filtered problem has a different block structure
                                        (an image with constant blocks is left with one residual block and becomes free,
                                        bundle_adjustment.cc:361): the session is unchanged, build the problem afresh */

/* Termination types, mirroring the subset of ceres::SolverTerminationType the
 * trust-region minimizer can produce (Ceres 1.8 semantics). */
#define MAVBA_TERM_NO_CONVERGENCE 0      /* max_num_iterations reached      */
#define MAVBA_TERM_FUNCTION_TOLERANCE 1
#define MAVBA_TERM_GRADIENT_TOLERANCE 2
#define MAVBA_TERM_PARAMETER_TOLERANCE 3 /* step or trust-region radius too small */
#define MAVBA_TERM_NUMERICAL_FAILURE 4   /* too many consecutive invalid steps    */
#define MAVBA_TERM_RUNNING (-1)          /* session only: not terminated yet       */

/*
 * The flattened problem. Caller owns every array. `poses`, `intrinsics` and
 * `points` are read AND written in place (the reference hands Ceres raw
 * double* into FeatureManager's hash-map nodes, bundle_adjustment.cc:311-317).
 *
 * Observations must already be filtered (min_track_len) and are given in the
 * reference's residual-block order (FREE images, FIXED images, FIXED_X images;
 * inside an image the image_to_points2D order — bundle_adjustment.cc:511-533).
 * The order only fixes the meaning of `point_error` accumulation and the
 * summation order of the CPU oracle; the device path re-sorts internally.
 */
typedef struct mavba_problem {
  int32_t num_images;
  int32_t num_cameras;
  int32_t num_points;
  int64_t num_obs;

  double* poses;               /* [num_images][6]  rvec(3) then tvec(3); in/out   */
  const uint8_t* pose_const;   /* [num_images]     MAVBA_CONST_* bitmask          */
  const int32_t* image_camera; /* [num_images]     physical camera of each image  */

  double* intrinsics;          /* [num_cameras][MAVBA_MAX_INTR]; first K used; in/out */
  const int32_t* camera_model; /* [num_cameras]    MAVBA_MODEL_*                  */
  const uint8_t* intr_const;   /* [num_cameras]    1 = intrinsics held constant   */

  double* points;              /* [num_points][3]; in/out                         */
  const uint8_t* point_const;  /* [num_points] 1 = held constant (GCP); may be NULL */

  const double* obs_uv;        /* [num_obs][2]     measured pixel                 */
  const int32_t* obs_image;    /* [num_obs]                                        */
  const int32_t* obs_point;    /* [num_obs]                                        */

  /* Rotation-prior residuals, one scalar residual per listed image, NULL loss
   * (reference BARotationConstraintCostFunction, bundle_adjustment.cc:57-111,
   * added for FREE images only at :428-444). */
  int32_t num_rot_priors;
  const int32_t* rot_prior_image; /* [num_rot_priors]                             */
  const double* rot_prior_rvec;   /* [num_rot_priors][3]  rvec0                   */
  double rot_prior_weight;
} mavba_problem;

/*
 * Solver options. The first block mirrors BundleAdjustmentOptions
 * (reference src/base3d/bundle_adjustment.h:38-114); the second block are the
 * Ceres-Solver defaults the reference never overrides, kept
 * here so tests can pin them. Always start from mavba_options_init().
 */
typedef struct mavba_options {
  int32_t max_num_iterations;   /* 100  */
  double function_tolerance;    /* 1e-4 */
  double gradient_tolerance;    /* 1e-8 (relative to the initial max|g|)          */
  double loss_scale_factor;     /* 1.0  Cauchy scale a; rho(s) = a^2 log(1+s/a^2)  */
  int32_t update_point_errors;  /* 0    fill point_error (needs non-NULL array)   */
  int32_t print_progress;       /* 0    per-iteration table on stdout              */

  double parameter_tolerance;               /* 1e-8  */
  double initial_trust_region_radius;       /* 1e4   */
  double max_trust_region_radius;           /* 1e16  */
  double min_trust_region_radius;           /* 1e-32 */
  double min_relative_decrease;             /* 1e-3  */
  double min_lm_diagonal;                   /* 1e-6  */
  double max_lm_diagonal;                   /* 1e32  */
  int32_t max_num_consecutive_invalid_steps; /* 10 (bundle_adjustment.cc:559)      */
  int32_t jacobi_scaling;                   /* 1     */

  int32_t device;               /* HIP device ordinal; -1 = current device         */
  int32_t profile_kernels;      /* 1 = bracket kernels with HIP events (bench)     */
} mavba_options;

typedef struct mavba_result {
  double initial_cost;           /* 1/2 sum rho(|r|^2), incl. fixed cost           */
  double final_cost;
  double fixed_cost;             /* cost of residual blocks with no free parameter */
  int64_t num_residuals;         /* 2*num_obs + num_rot_priors                     */
  int64_t num_residuals_reduced;
  int64_t num_parameters_reduced;
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t termination;           /* MAVBA_TERM_*                                   */
  double final_gradient_max_norm;
  double final_trust_region_radius;
  double setup_seconds;          /* host indexing + upload                          */
  double solve_seconds;          /* LM loop                                         */
} mavba_result;

void mavba_options_init(mavba_options* opt);

/* Human-readable message of the last failing call on this thread. */
const char* mavba_last_error(void);

/* Number of usable HIP devices (0 when there is none; never fails). */
int mavba_device_count(void);

/*
 * One-shot global/local BA: upload, solve, write parameters back into
 * problem->poses/intrinsics/points. `point_error` (may be NULL) receives, for
 * every point, sum over its observations of |r_raw| / (number of its
 * observations in the problem) — bundle_adjustment.cc:575-598 — and is left
 * untouched for points without observations.
 * Replaces: the body of bundle_adjustment(), bundle_adjustment.cc:473-612.
 */
int mavba_solve(const mavba_problem* problem, const mavba_options* options,
                mavba_result* result, double* point_error);

/*
 * Global BA, point filter, global BA again on ONE resident session (reference src/mapper.cc:1206 + 1218-1224:
 * adjust_global_bundle, filter_point_cloud, adjust_global_bundle). Parameters are written back in place after
 * the second solve; `removed` [num_points] (may be NULL) marks the filtered points, whose FeatureManager entries
 * the caller deletes; `point_error` (may be NULL) receives the errors of the points that remain.
 * `first` / `second` (may be NULL) summarise the two solves.
 */
int mavba_solve_filter_solve(const mavba_problem* problem, const mavba_options* options, double filter_max_error,
                             const uint8_t* keep, mavba_result* first, mavba_result* second, double* point_error,
                             uint8_t* removed, int64_t* num_removed);

/*
 * Single-camera 6-DoF refinement with points and intrinsics held constant.
 * `uv` [n][2], `xyz` [n][3], `inlier_mask` [n] (NULL = all inliers).
 * Replaces: pose_refinement(), bundle_adjustment.cc:139-225.
 */
int mavba_pose_refine(double rvec[3], double tvec[3], const double* intrinsics,
                      int32_t camera_model, const double* uv, const double* xyz,
                      const uint8_t* inlier_mask, int64_t n,
                      const mavba_options* options, mavba_result* result);

/*
 * Many pose refinements in ONE launch: the inlier sets of several RANSAC hypotheses of one image, or the image
 * pairs of several images (reference call site: src/sfm/sequential_mapper.cc:709-732, one call per processed
 * pair). Each item is an independent pose_refinement() problem; one work-group runs the whole trust-region
 * loop of one item on the device. rvec / tvec of every item are updated in place; `results` [count] may be NULL.
 * mavba_pose_refine() is this call with count = 1.
 */
typedef struct mavba_pose_refine_item {
  double rvec[3];              /* in/out */
  double tvec[3];              /* in/out */
  const double* intrinsics;    /* first K values of the camera's parameters */
  int32_t camera_model;        /* MAVBA_MODEL_*                             */
  const double* uv;            /* [n][2]                                    */
  const double* xyz;           /* [n][3]                                    */
  const uint8_t* inlier_mask;  /* [n], NULL = all inliers                   */
  int64_t n;
} mavba_pose_refine_item;
int mavba_pose_refine_batch(int32_t count, mavba_pose_refine_item* items, const mavba_options* options,
                            mavba_result* results);

/* ------------------------------------------------------------------------
 * Scene API: an incremental flat mirror of the FeatureManager.
 *
 * The reference re-walks the FeatureManager's hash maps on every bundle_adjustment() call
 * (src/base3d/bundle_adjustment.cc:228-387; local BA runs after every image, src/mapper.cc:1120-1135).
 * A scene receives the same information as deltas when the FeatureManager changes - one call per
 * add_camera / add_image / set_pose / add_point2D / set_point3D / correspondence / delete_point3D
 * (reference src/fm/feature_management.h:30-60; the hook lines are listed in INTEGRATION.md) - keeps it
 * in dense arrays indexed by the caller's own ids (the 1-based ids of the FeatureManager are fine) and
 * builds the flat problem of a call from them without hashing, by the reference's rules
 * (bundle_adjustment.cc:228-549): observation count inside the selected image set, min_track_len,
 * residual order FREE / FIXED / FIXED_X, constancy only for images with more than one residual,
 * GCPs constant, rotation priors incl. the pre-rotation of the WHOLE scene.
 * ---------------------------------------------------------------------- */
typedef struct mavba_scene mavba_scene;

/* The BundleAdjustmentOptions members that act on problem construction (bundle_adjustment.h:38-114). */
typedef struct mavba_scene_options {
  int32_t min_track_len;            /* 2 */
  int32_t refine_camera_params;     /* 0 */
  int32_t constrain_rotation;       /* 0 */
  double constrain_rotation_weight; /* 0 */
} mavba_scene_options;

int mavba_scene_create(mavba_scene** out);
void mavba_scene_destroy(mavba_scene* s);
/* add or update; `params`: the first K values of the model (FeatureManager::add_camera without the code slot) */
int mavba_scene_set_camera(mavba_scene* s, int64_t camera_id, int32_t model, const double* params);
/* add_image / set_pose: camera_id < 0, rvec == NULL or tvec == NULL leave that part as it is */
int mavba_scene_set_image(mavba_scene* s, int64_t image_id, int64_t camera_id, const double* rvec, const double* tvec);
/* add_point2D: appended to the image's list (the image_to_points2D order is the residual order inside an image) */
int mavba_scene_add_point2d(mavba_scene* s, int64_t image_id, int64_t point2D_id, const double* xy);
/* the same for `count` points of one image in one call (mirroring an existing FeatureManager, or add_image with its
 * features): ids and pixels as arrays, `point3D_ids` (may be NULL) the links, < 0 = none */
int mavba_scene_add_points2d(mavba_scene* s, int64_t image_id, int64_t count, const int64_t* point2D_ids, const double* xy,
                             const int64_t* point3D_ids);
/* add_point3D + set_point3D */
int mavba_scene_set_point3d(mavba_scene* s, int64_t point3D_id, const double* xyz);
/* point2D_to_point3D[point2D_id] = point3D_id (add_correspondence, merges); point3D_id < 0 removes the entry */
int mavba_scene_link(mavba_scene* s, int64_t point2D_id, int64_t point3D_id);
/* delete_point3D: its 2-D points read as unmatched from now on */
int mavba_scene_delete_point3d(mavba_scene* s, int64_t point3D_id);
int mavba_scene_get_image(mavba_scene* s, int64_t image_id, double* rvec, double* tvec);
int mavba_scene_get_point3d(mavba_scene* s, int64_t point3D_id, double* xyz);
int mavba_scene_get_camera(mavba_scene* s, int64_t camera_id, int32_t* model, double* params);

/* The flat problem of one bundle_adjustment() call (views into the scene, valid until its next flatten / bundle
 * adjust): exactly what shim/base3d/bundle_adjustment.cc hands to mavba_solve for the same FeatureManager content.
 * `image_ids` / `camera_ids` / `point_ids` (may be NULL) receive the flat-index -> caller-id tables. The two
 * std::invalid_argument conditions of the reference (:459-471) are MAVBA_ERR_INVALID_ARGUMENT here. With
 * constrain_rotation the whole scene is rotated first (:399-425), as the reference rotates the whole FeatureManager. */
int mavba_scene_flatten(mavba_scene* s, const int64_t* free_ids, int64_t n_free, const int64_t* fixed_ids, int64_t n_fixed,
                        const int64_t* fixed_x_ids, int64_t n_fixed_x, const int64_t* gcp_ids, int64_t n_gcp,
                        const int64_t* rot_image_ids, const double* rot_rvecs, int64_t n_rot,
                        const mavba_scene_options* scene_options, mavba_problem* problem, const int64_t** image_ids,
                        const int64_t** camera_ids, const int64_t** point_ids);

/* bundle_adjustment() on the scene: flatten, mavba_solve, results written back into the scene. `final_cost_px` =
 * the reference's return value sqrt(final_cost / num_residuals) (:610). With options->update_point_errors the views
 * `error_point_ids` / `error_values` (`num_errors` entries, valid until the next call) list the point3D errors. */
int mavba_scene_bundle_adjust(mavba_scene* s, const int64_t* free_ids, int64_t n_free, const int64_t* fixed_ids, int64_t n_fixed,
                              const int64_t* fixed_x_ids, int64_t n_fixed_x, const int64_t* gcp_ids, int64_t n_gcp,
                              const int64_t* rot_image_ids, const double* rot_rvecs, int64_t n_rot,
                              const mavba_scene_options* scene_options, const mavba_options* options, mavba_result* result,
                              double* final_cost_px, const int64_t** error_point_ids, const double** error_values,
                              int64_t* num_errors);

/* ------------------------------------------------------------------------
 * Session API: the same solver with device-resident state, for callers that
 * iterate (local BA after every image), for the parity tests (intermediate
 * quantities) and for bench.py (inputs resident in HBM before timing starts).
 * ---------------------------------------------------------------------- */
typedef struct mavba_session mavba_session;

int mavba_session_create(const mavba_problem* problem,
                         const mavba_options* options, mavba_session** out);
void mavba_session_destroy(mavba_session* s);

/* Restore the parameters given at creation and reset the LM state. */
int mavba_session_reset(mavba_session* s);

/* Run LM iterations until termination or until `max_iters` more iterations
 * have been done; returns the number done in *iters_done. A terminated
 * session does nothing until reset. */
int mavba_session_iterate(mavba_session* s, int32_t max_iters,
                          int32_t* iters_done, int32_t* termination);

/* Summary so far (costs, counts, termination). */
int mavba_session_result(mavba_session* s, mavba_result* result);

/* Copy current parameters to the host arrays (any may be NULL). */
int mavba_session_get_params(mavba_session* s, double* poses,
                             double* intrinsics, double* points);

/* Per-point mean raw reprojection error at the current parameters. */
int mavba_session_point_errors(mavba_session* s, double* point_error);

/* Overwrite current parameters on the device (any array may be NULL = keep). The structure of the problem does
 * not change, so nothing of the set-up is repeated; the LM state is NOT touched (call mavba_session_restart). */
int mavba_session_set_params(mavba_session* s, const double* poses, const double* intrinsics, const double* points);

/* Begin a new solve from the current parameters: what a second bundle_adjustment() call on the same data does
 * (fresh trust region, Jacobi scaling re-estimated, counters zeroed) without repeating the set-up. */
int mavba_session_restart(mavba_session* s);

/*
 * Point filtering on the resident session. Replaces: filter_point_cloud(), reference src/mapper.cc:382-402, whose
 * callers run a global BA, filter, and run the global BA again (src/mapper.cc:1206, 1218-1224).
 * Every point whose point3D error (mean raw reprojection error at the current parameters,
 * bundle_adjustment.cc:575-598) exceeds `max_error` leaves the problem - all its residual blocks are removed -
 * unless keep[point] != 0 (`keep` may be NULL; the reference's keep_point3D_ids). Counts, used / constant
 * blocks and the fixed cost are re-derived and the session is restarted, ready for mavba_session_iterate.
 *   removed   [num_points] (may be NULL): 1 for every point filtered so far
 *   errors    [num_points] (may be NULL): the errors the decision was taken on (untouched for points without observations)
 */
int mavba_session_filter_points(mavba_session* s, double max_error, const uint8_t* keep, uint8_t* removed,
                                double* errors, int64_t* num_removed);

/*
 * Multi-GPU hook. When the points are sharded over several processes (one
 * per GPU), every rank holds all cameras and a disjoint subset of points and
 * observations; the reduced camera system and a handful of scalars must be
 * summed over ranks once per linear solve. The session calls `fn` with a
 * device pointer to `count` contiguous doubles that must be all-reduced in
 * place (op 0 = sum, 1 = max, 2 = sum for the first count-1 doubles and max
 * for the last one) before `fn` returns. The stream the session
 * works on is idle while `fn` runs. With no hook set the session is
 * single-rank.
 */
typedef int (*mavba_allreduce_fn)(void* ctx, void* device_ptr, int64_t count,
                                  int32_t op);
int mavba_session_set_allreduce(mavba_session* s, mavba_allreduce_fn fn,
                                void* ctx, int32_t rank, int32_t world_size);

/*
 * Native collective: the same exchange through RCCL inside the library. ncclAllReduce is enqueued on the
 * session's own HIP stream (no host synchronisation, no callback), so the LM loop keeps its one read-back per
 * iteration. One process per GPU: rank 0 obtains an id with mavba_rccl_unique_id() (128 bytes, ncclUniqueId),
 * the launcher hands it to every rank (bench.py broadcasts it with torch.distributed), every rank then calls
 * mavba_session_set_rccl() - collectively, like ncclCommInitRank - before the first iteration. librccl.so is
 * loaded on first use.
 */
int mavba_rccl_unique_id(void* out128);
int mavba_session_set_rccl(mavba_session* s, const void* unique_id128, int32_t rank, int32_t world_size);

/* ---- probes used by tests/ and bench.py -------------------------------- */

/* Evaluate residuals + Jacobians at the current parameters (the Jacobian
 * sweep) and download them in the caller's observation order:
 *   r  [num_obs][2]        loss-corrected residual
 *   Jc [num_obs][2][6]     d r / d (rvec, t)
 *   Jp [num_obs][2][3]     d r / d point
 *   Jk [num_obs][2][9]     d r / d intrinsics (columns >= K are 0)
 * Any output may be NULL. *cost receives 1/2 sum rho. */
int mavba_session_eval_jacobian(mavba_session* s, double* cost, double* r,
                                double* Jc, double* Jp, double* Jk);

/* Test / debugging aid, needs no device: the elimination tree (nested-dissection order of the reduced camera system,
 * what replaces CHOLMOD's fill-reducing ordering behind ceres SPARSE_SCHUR, bundle_adjustment.cc:555) the session
 * set-up chooses for an image graph given as `npairs` coupled image pairs (images that share a 3-D point).
 * node_of_image [NI]: tree node of every image; node_parent [cap]: parent of every node, -1 for the root; nodes are
 * numbered in elimination order (children before parents). Returns the number of nodes, 0 = no dissection. */
/* Host-only: the process-wide RCCL communicator group of the in-process ranks (MAVBA_GPUS; csrc/multi_gpu.hip), `calls`
 * acquisitions for `world` ranks, optionally an abort + one more. out[3]: communicators, kept across calls (0/1), rebuilt
 * after the abort (0/1). Meant for tests with a stand-in library (MAVBA_RCCL_LIB). */
int mavba_debug_inproc_comms(int32_t world, int32_t calls, int32_t abort_after, int64_t* out);

int mavba_debug_elimination_tree(int32_t num_images, int32_t num_cameras, int64_t npairs, const int32_t* pair_a,
                                 const int32_t* pair_b, int32_t max_depth, int32_t* node_of_image, int32_t* node_parent,
                                 int32_t cap);

/* Test entry of the set-up's device sort (device_setup.hip): order_out [n] = the stable ascending order of 0..n-1 by
 * keys[i] (the low `key_bytes` bytes are significant). */
int mavba_debug_radix_sort(int32_t n, const uint32_t* keys, int32_t key_bytes, int32_t* order_out, int32_t device);
/* Test entry of the set-up's batched small uploads (csrc/host_util.hip: one copy + one scatter kernel for the ~50 small tables of a
 * local-window session): n buffers of |sizes[i]| bytes - sizes[i] < 0: dirtied, then cleared; every third one uploaded twice -
 * inside one batch whose arena holds `arena_bytes` (small values exercise the arena-full path), read back and compared.
 * Returns the number of wrong bytes (0 = pass), -1 on error. */
int64_t mavba_debug_upload_batch(int32_t n, const int64_t* sizes, int64_t arena_bytes, int32_t device);
/* Test entry: the LM accept / reject / terminate decision (csrc/lm_decide.h) by the host build and by the device build on
 * the same n cases (16 scalars + 8 parameters each, 6 doubles out each); the speculative evaluation needs them identical. */
int mavba_debug_lm_decide(int32_t n, const double* cases, double* out_host, double* out_device, int32_t device);
/* Test entry (no device needed): tile structure + persistent schedule of a factorisation with `nb` tile columns, elimination
 * tree nodes [begin, end) / parent, non-zero lower tiles (row, col), for a device of `cus` compute units.
 * out[8] = schedule exists, modelled forward us, launch-per-panel estimate us, grid, chain work-groups, tiles, updates, nodes;
 * tasks_out: 6 ints per task in queue order (work-group, kind, i, j, first update, end update); upd_out: the update lists;
 * chain_info_out[nb]: bit 0 / 1 = the chain column waits for a PRE tile of its diagonal / sub-diagonal tile. */
int mavba_debug_chol_schedule(int32_t nb, int32_t num_nodes, const int32_t* node_begin, const int32_t* node_end, const int32_t* node_parent,
                              int64_t num_pairs, const int32_t* pair_row, const int32_t* pair_col, int32_t cus, double* out,
                              int32_t* tasks_out, int64_t tasks_cap, int64_t* num_tasks, int32_t* upd_out, int64_t upd_cap,
                              int64_t* num_upd, int32_t* chain_info_out);

/* Build the reduced camera system for the current Jacobian and trust-region
 * radius and download it: S [n][n] row-major (both triangles), v [n], with
 * n = mavba_session_reduced_dim(). Column j of image i's pose is 6*i+j,
 * intrinsic k of camera c is 6*num_images + 9*c + k; constant / unused
 * columns have S_jj = 1, v_j = 0. */
int mavba_session_reduced_dim(mavba_session* s);
int mavba_session_reduced_system(mavba_session* s, double radius, double* S,
                                 double* v);

/* Solve the current LM linear system at `radius` and download the
 * (Jacobi-unscaled) step: d_poses [num_images][6], d_intr [num_cameras][9],
 * d_points [num_points][3]; *model_cost_change as used by the step test. */
int mavba_session_linear_step(mavba_session* s, double radius, double* d_poses,
                              double* d_intr, double* d_points,
                              double* model_cost_change);

/* Time `reps` back-to-back launches of the Jacobian-sweep kernel alone with
 * HIP events on the session's stream; *ms_avg = average per launch. */
int mavba_session_time_jacobian(mavba_session* s, int32_t reps, float* ms_avg);
/* options.profile_kernels for the iterations that follow (the event brackets cost ~10 us per kernel: time without, profile after). */
int mavba_session_set_profiling(mavba_session* s, int32_t on);
/* Probe: `reps` passes of the linear solve's front end (Jacobians, point blocks, cluster Schur complements: k_schur_rows)
 * at trust-region radius `radius`; average milliseconds per pass. */
int mavba_session_time_front(mavba_session* s, double radius, int32_t reps, float* ms_avg);

/* Structure of the session's linear algebra (for roofline accounting in bench.py). */
typedef struct mavba_session_info {
  int64_t num_obs_kept;        /* observations in the reduced program                         */
  int32_t reduced_dim;         /* n = 6*num_images + 9*num_cameras (incl. constant columns)   */
  int32_t padded_dim;          /* n rounded up to the 64-column tile                           */
  int64_t schur_terms[3];      /* entry-pair terms of the pose-pose / intr-pose / intr-intr blocks */
  int64_t schur_blocks;        /* blocks of the reduced camera system that are assembled      */
  int64_t intr_entries;        /* (point, camera) intrinsics entries                          */
  int64_t envelope_tiles;      /* 64x64 tiles inside the factorisation's envelope (lower)      */
  int64_t dense_tiles;         /* nb*(nb+1)/2                                                   */
  double factor_flops;         /* FP64 flops of one factorisation + solves on the envelope     */
  double dense_factor_flops;   /* n^3/3 + 2 n^2, the dense-equivalent count of SURVEY.md 8(d)  */
  int32_t matrix_dim;          /* columns of the factorised matrix (elimination order, parts padded to tiles) */
  int32_t nd_parts;            /* uncoupled leading parts factorised concurrently (0 = single chain) */
  int32_t chain_steps;         /* dependent 64-column panel steps of the factorisation schedule */
  int32_t num_clusters;        /* point clusters of the Schur complement (k_schur_clusters)     */
  int64_t clustered_points;    /* points whose Schur terms are formed inside a cluster          */
  int64_t cluster_partials;    /* (cluster, block) partials the clusters emit per linear solve  */
  double cluster_flops;        /* FP64 MFMA flops k_schur_clusters executes per linear solve (E E^T incl. structural zeros) */
  double chol_model_forward_us; /* the host-side timing model's forward factorisation on the persistent schedule (0: other schedule) */
  int64_t reduced_store_bytes; /* device bytes of the reduced system S | v (as many again for its factor): the envelope's tiles + one
                                  right-hand-side tile per tile column, 32 KiB each - what SPARSE_SCHUR's sparse storage is to Ceres
                                  (reference src/base3d/bundle_adjustment.cc:555); a dense (n + 64) n array is matrix_dim^2 * 8 */
} mavba_session_info;
int mavba_session_get_info(mavba_session* s, mavba_session_info* out);

/* Per-kernel event timings accumulated while options.profile_kernels != 0.
 * Returns the number of kernels; fills up to `cap` entries. */
typedef struct mavba_kernel_stat {
  char name[48];
  int64_t launches;
  double total_ms;
} mavba_kernel_stat;
int mavba_session_kernel_stats(mavba_session* s, mavba_kernel_stat* out,
                               int32_t cap);

/* Dense SPD solve used for the reduced camera system, exposed for tests:
 * solves A x = b, A [n][n] row-major symmetric positive definite (host
 * arrays). Returns MAVBA_ERR_INVALID_ARGUMENT if A is not numerically SPD. */
int mavba_dense_spd_solve(int32_t n, const double* A, const double* b,
                          double* x, int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* MAVBA_H_ */
