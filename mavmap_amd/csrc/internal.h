// internal.h — shared declarations between the HIP kernel files and the host LM driver.
#ifndef MAVBA_INTERNAL_H_
#define MAVBA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>
#include <functional>

namespace mavba {

// Record strides (doubles) of the Schur "entry" arrays.
//   pose entry  (one per observation a of a free point):   U_a (6x3) | e_a (6)
//   intr entry  (one per (free point, free camera) pair):  Uk  (9x3) | ek  (9)
constexpr int kPoseRec = 24;
constexpr int kIntrRec = 36;

// Per-image / per-camera camera-side sums produced by the camera sweep
// (unscaled, loss-corrected):
//   image record  [81] = PP(21 sym 6x6) | Pg(6) | PI(6x9 row-major)
//   camera record [54] = II(45 sym 9x9) | Ig(9)
constexpr int kImgRec = 81;
constexpr int kCamRec = 54;
constexpr int kSweepAcc = 135;  // PP21 + Pg6 + PI54 + II45 + Ig9 (per chunk partial)

// One unit of work of the camera sweep: a run of image-major observations.
struct SweepChunk { int image; int begin; int end; };

// Schur block kinds.
enum { BLK_PP = 0, BLK_IP = 1, BLK_II = 2 };

// A block of the reduced camera system S (lower triangle): rows belong to
// `row_ent`, columns to `col_ent` (image index for poses, camera index for
// intrinsics); its terms live in chunks [chunk_begin, chunk_end) of its kind.
struct SchurBlock {
  int kind;
  int row_ent, col_ent;
  int chunk_begin, chunk_end;
};
struct SchurChunk { int begin; int end; int slot; };
// Pre-reduction of a long run of partials of one block: partial[dst] = sum partial[src_begin..src_end) (fixed order)
struct PartialReduce { int kind, src_begin, src_end, dst; };
// The pre-reduction's tasks as EXTRA work-groups of an evaluation kernel that runs right behind the cluster kernel and does not
// depend on it (k_camera_reduce_img / k_eval_head, round 6): one launch less per linear solve. n == 0: nothing rides.
struct PartialRide { int n = 0; const PartialReduce* tasks = nullptr; double* pp = nullptr; double* ip = nullptr; double* ii = nullptr; };
void launch_partial_reduce(hipStream_t st, int num_tasks, const PartialReduce* tasks, double* part_pp, double* part_ip,
                           double* part_ii);  // term range; slot = index of the partial it writes

// Point clusters of the Schur complement (k_schur_clusters): a run of consecutive points [p0, p1) whose
// images fit a local list of kClImages and whose shared cameras fit kClCams. The cluster's contribution to
// EVERY block among its images / cameras is one symmetric product E E^T of its stacked entry matrix
//   rows  6*la + r        : U_a (6 x 3) of the observation in local image la      (la < kClImages)
//   rows  96 + 9*lc + r   : Uk  (9 x 3) of the point's entry for local camera lc  (lc < kClCams)
//   row   123             : h^T of the point                                       (gives e_a / ek)
//   cols  3*point + t
// computed on the FP64 matrix cores from LDS; each block that is present in the cluster then leaves as
// ONE partial (slot table) instead of one gathered term per (point, pair).
// Two shapes: 16 images x 3 cameras (128 rows, 36 lower 16x16 tiles) and 12 images x 2 cameras (96 rows, 21 tiles:
// 42 % fewer matrix instructions when the tracks are short and at most two cameras are refined).
struct ClusterShape {
  int images, cams;
  int cam_row0() const { return 6 * images; }
  int hrow() const { return 6 * images + 9 * cams; }
  int rows() const { return (hrow() + 1 + 15) / 16 * 16; }
  int tab_pp() const { return 0; }
  int tab_ip() const { return images * (images + 1) / 2; }
  int tab_ii() const { return tab_ip() + cams * images; }
  int tab() const { return tab_ii() + cams * (cams + 1) / 2; }
};
constexpr int kClImagesMax = 16, kClCamsMax = 3;
constexpr int kClBatch = 32;                     // points per LDS batch -> K = 96 columns (16 -> two work-groups per CU, measured slower)
constexpr int kTailObs = 16;                     // a point with more observations than a cluster has images sorts into the tail of the point order
constexpr int kClMaxBatches = 64;                // a cluster spans at most kClMaxBatches * kClBatch consecutive points
struct SchurCluster { int p0, p1; };

// Point order (order_on_host / k_point_keys): one image's contribution to the 32-bit hash of a point's image set (summed
// over the point's observations, so the order of the observations does not matter).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned image_set_mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// ---- k_schur_rows (schur_rows.hip): the fused front end + cluster Schur complement of round 4 ----
// A point owns a 16-lane DPP row, a batch is 16 points (K = 48 columns of the entry matrix). The entry matrix of a cluster
// has rows 6 * slot + e for its `ni` image slots, then the parameters of its `nc` camera slots (4 / 8 / 9 rows each), then the
// h row; the cluster's ROW CLASS (16-row blocks NT of the matrix) selects the batch loop: 80 rows hold 11 images + a PINHOLE and
// an OPENCV camera + h (15 tiles), 96 rows 13 images (21 tiles), 128 rows the general 16 x 3 list (36 tiles). Slot tables and
// image / camera lists keep the 16 x 3 layout of ClusterShape{16, 3}: a cluster of k_schur_rows is also a valid cluster of
// k_schur_clusters.
constexpr int kRowsBatch = 16;
constexpr int kRowsMaxPoints = 256;
constexpr int kRowsClasses = 3;
constexpr int kRowsClassNT[kRowsClasses] = {5, 6, 8};
// emit_off / emit_n: the cluster's emit map (rows_emit_map below) - where every element of its block partials comes from in the
// packed lower triangle of its product; one map per cluster shape, passes as in the kernel (a 128-row triangle leaves in two).
// flags: bit 0 = kRowsUnplaced; bits 8-15 / 16-23 = first row of camera slot 1 / 2 inside the camera rows, bits 24-31 = camera
// rows in all. Round 6: a camera slot takes as many rows as its model has parameters (4 / 8 / 9), not 9 - the common cluster of
// 11 images + a PINHOLE and an OPENCV camera has 66 + 12 + 1 = 79 rows (15 tiles) instead of 85 (21 tiles).
struct SchurRowsCluster { int p0, p1, ni, nc, emit_off, emit_n[2], flags; };
constexpr int kRowsUnplaced = 1;  // flags: two camera slots, but some point has more than 8 observations of one of them (see lanemap)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rows_cam_off(int flags, int lc) { return lc <= 0 ? 0 : lc >= 3 ? (flags >> 24) & 255 : (flags >> (8 * lc)) & 255; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rows_cam_rows(int flags) { return (flags >> 24) & 255; }
// k[lc]: parameters of the camera model in slot lc (nc slots)
inline int rows_pack_flags(bool unplaced, const int* k, int nc) {
  int off[4] = {0, 0, 0, 0};
  for (int c = 0; c < 3; ++c) off[c + 1] = off[c] + (c < nc ? k[c] : 0);
  return (unplaced ? kRowsUnplaced : 0) | off[1] << 8 | off[2] << 16 | off[3] << 24;
}
// One entry of an emit map: source index in the pass's packed triangle (13 bits; kRowsEmitZero: no source row - a parameter the
// slot's model does not have - the partial's element is written as 0), offset inside the destination partial (7 bits), index of
// the partial's slot in the cluster's slot table (8 bits).
constexpr unsigned kRowsEmitZero = 8191u;
constexpr unsigned rows_emit_entry(int src, int o, int tab_index) { return (unsigned)src | (unsigned)o << 13 | (unsigned)tab_index << 20; }
// the lists of a k_schur_rows cluster: 16 image slots, the camera of each, 3 camera slots, 1 pad
constexpr int kRowsListsCam = kClImagesMax, kRowsListsCams = 2 * kClImagesMax, kRowsLists = 2 * kClImagesMax + kClCamsMax + 1;
constexpr int kRowsPassSplit = 96;  // rows [0, 96) and [96, 128) of a 128-row product are staged one after the other
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rows_class_of_rows(int rows) { return rows <= 16 * kRowsClassNT[0] ? 0 : rows <= 16 * kRowsClassNT[1] ? 1 : 2; }
// rows of a cluster's entry matrix: 6 per image slot, the camera slots' parameters, the h row
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rows_count(int ni, int cam_rows) { return 6 * ni + cam_rows + 1; }

// Scalars exchanged with the host every LM iteration (device array of doubles).
// Two groups, reduced over ranks SEPARATELY (an evaluation enqueued behind an accepted step and the next candidate
// are read back together: a collective over one group must never touch - and re-sum - the other group's slots):
//   evaluation [SC_COST, SC_GRAD_MAX]: two sums, then the max (all-reduce op 2 over SC_EVAL_COUNT doubles)
//   candidate  [SC_NEW_COST, SC_CAND_END): sums
enum {
  SC_COST = 0,        // 1/2 sum rho at the evaluation point (no fixed cost)
  SC_XNORM2,          // |x|^2 over free parameters
  SC_GRAD_MAX,        // max |g_j| over free parameters (unscaled gradient); directly after the evaluation's sums
  SC_NEW_COST,        // candidate cost
  SC_STEP_NORM2,      // |delta|^2
  SC_MODEL_CHANGE,    // model cost change
  SC_CAND_XNORM2,     // |x + delta|^2
  SC_FAIL,            // > 0: the reduced solve failed (non-positive pivot; >= 1e30: the persistent launch gave up)
  SC_FAIL_FRONT,      // > 0: a point's damped 3x3 block is not SPD (written by the front end, which may run once for several solves)
  SC_CAND_END,
  SC_EVAL_COUNT = SC_GRAD_MAX + 1,
  SC_CAND_BEGIN = SC_NEW_COST,
  SC_CAND_COUNT = SC_CAND_END - SC_CAND_BEGIN,
  SC_COUNT = 16
};

// ---- the LM decision shared by the host loop and the speculative evaluation's kernels (lm_decide.h) ----
enum { LM_REJECTED = 0, LM_ACCEPTED = 1, LM_INVALID = 2, LM_TERM_PTOL = 3, LM_TERM_FTOL = 4, LM_TERM_GTOL = 5 };
// Loop state and options at the moment the candidate was enqueued (what k_lm_snapshot and the host loop feed lm_decide
// with). dec: where k_lm_snapshot left {code, radius} on the device - a kernel of the speculative evaluation runs only if
// dec[0] == LM_ACCEPTED (null: not speculative, always runs).
struct LmSpec {
  const double* dec;
  const double* scal;
  double radius, decrease_factor;                       // trust region before the decision
  double ptol, ftol, min_rel_dec, max_radius, abs_gtol;  // options (abs_gtol = gradient_tolerance * max|g_0|)
  int pending_eval;                                      // the scalars of the evaluation at the current point arrive with this candidate
};
struct LmDecision {
  int code;
  double radius, decrease_factor;        // trust region after the decision (unchanged by a termination)
  double rel, step_norm, cost_change;    // for the progress line
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline LmSpec lm_spec_off() { LmSpec s{}; s.dec = nullptr; s.scal = nullptr; return s; }

// ---- launch wrappers (kernels.hip) -----------------------------------------
struct SweepArgs {
  int N, Nstride, NI, NC, KMAX;
  const double2* uv; const int* obs_img; const int* obs_pt;
  const double* camrec; const double* intr; const int* img_cam; const int* cam_model;
  const double* points;
  const unsigned char* pt_active;                    // null, or 0 for points filtered out of the resident problem (their rows are zero)
  double loss_b, loss_inv_b;
  double* R; double* Jp; double* Jc; double* Jk;   // SoA planes, stride Nstride
  double* cost_partial;                              // [grid]
};
int jacobian_sweep_grid(int N);
void launch_cam_prepare(hipStream_t st, int NI, const double* poses, double* camrec);
void launch_jacobian_sweep(hipStream_t st, const SweepArgs& a);
void launch_cost_only(hipStream_t st, const SweepArgs& a);  // uses uv/obs/points/camrec/intr, writes cost_partial
void launch_raw_residual_norm(hipStream_t st, const SweepArgs& a, double* out_norm);  // |r_raw| per obs

void launch_point_reduce(hipStream_t st, int NP, int NPs, int Nstride, int KMAX, const int* pt_start,
                         const int* q_start, const int* q_cam, const int* obs_img, const int* img_cam,
                         const double* R, const double* Jp, const double* Jk, double* Cu, double* gu,
                         double* Wk /*[Q][27]*/);

// ---- J-free, point-major Schur front end (k_point_front) ------------------------------------------------------
// One work-group per TILE of consecutive points (<= kFrontObs observations, <= kFrontPts points, <= kFrontQ (point,
// camera) intrinsics entries; a point with more observations than a tile holds is a tile of its own and is walked in
// windows). Residual and Jacobian of every observation are computed in registers, never stored: the per-point sums
// Cu = sum Jp^T Jp, gu = sum Jp^T r and Wk = sum Jk^T Jp go through LDS, the point's damped 3x3 block is factorised and
// the pose / intrinsics Schur entry records are written directly. The trust-region radius is a kernel argument: a
// rejected step re-runs the kernel, no Jacobian is kept between linear solves.
constexpr int kFrontObs = 256, kFrontPts = 64, kFrontQ = 96, kFrontMaxGrid = 65536;  // (one tile per work-group up to 65536 tiles, then contiguous runs)
struct FrontTile { int p0, p1, o0, o1, q0, q1; };  // points [p0, p1), their observations [o0, o1) and intrinsics entries [q0, q1)
struct FrontArgs {
  SweepArgs sw;                         // observations, cameras, points, loss, pt_active; cost_partial [grid]
  int num_tiles, NPs;
  const FrontTile* tiles;
  const int* pt_start; const int* q_start; const int* q_cam; const int* q_pt;
  const unsigned char* pt_free;
  const double* scale_cam; const double* scale_pt;
  double radius, dmin, dmax;
  double* Cu; double* gu; double* Gi; double* h;   // per-point planes, stride NPs
  double* Epose; double* Eintr;                     // entry records (kPoseRec / kIntrRec doubles)
  double* fail;
  long long* trace;  // MAVBA_FRONT_TRACE (debugging): s_memtime stamps per work-group, else null
  LmSpec spec;       // speculative evaluation: run only if the pending candidate is accepted, with the radius it leaves (`radius` is ignored then)
};
int point_front_grid(int num_tiles);
// kmax_intr: 0 when no intrinsics block is free (no Wk products), else the widest camera model (4, 8, 9).
// entries false: only Cu, gu and the cost (the first evaluation of a solve, before the Jacobi scales exist).
void launch_point_front(hipStream_t st, const FrontArgs& a, int kmax_intr, bool entries);

struct CamSweepArgs {
  int NI, NC;
  const SweepChunk* chunks; int num_chunks;
  const double2* im_uv; const int* im_pt;
  const double* camrec; const double* intr; const int* img_cam; const int* cam_model;
  const double* points; const unsigned char* pt_active; double loss_b, loss_inv_b;
  double* partial;  // [num_chunks][kSweepAcc]
  LmSpec spec;      // speculative evaluation (see FrontArgs)
};
void launch_camera_sweep(hipStream_t st, const CamSweepArgs& a, int kmax, bool any_intr_free);
void launch_camera_reduce(hipStream_t st, int NI, int NC, const int* img_chunk_start,
                          const double* partial, const int* prior_start, const double* prior_res,
                          const double* prior_jac, const int* cam_img_start, const int* cam_imgs,
                          double* img_rec, double* cam_rec, double* img_intr_tmp, bool with_cams = true, const LmSpec& spec = lm_spec_off(), const PartialRide& ride = PartialRide());
void launch_rot_prior(hipStream_t st, int n, const int* prior_img, const double* prior_R0, double w,
                      const double* poses, double* res, double* jac, double* cost_partial, const LmSpec& spec = lm_spec_off());

void launch_scales(hipStream_t st, int NI, int NC, int NP, int NPs, int jacobi,
                   const unsigned char* pose_free, const unsigned char* intr_free,
                   const unsigned char* pt_free, const double* img_rec, const double* cam_rec,
                   const double* Cu, double* scale_cam, double* scale_pt);

constexpr int kStateNormsCamBlocks = 64;  // partial[] holds 2 x (512 + this) values
void launch_state_norms(hipStream_t st, int NI, int NC, int NP, int NPs, bool cam_part,
                        const unsigned char* pose_free, const unsigned char* intr_free,
                        const unsigned char* pt_free, const double* poses, const double* intr,
                        const double* points, const double* img_rec, const double* cam_rec,
                        const double* gu, double* partial /*[grid][2]*/, int* grid_out, const LmSpec& spec = lm_spec_off());


void launch_entries_pose(hipStream_t st, int N, int Nstride, int NPs, const int* obs_img,
                         const int* obs_pt, const unsigned char* pt_free, const double* Jc,
                         const double* Jp, const double* scale_cam, const double* scale_pt,
                         double* Gi, double* h, double* Epose);
void launch_factor_entries_pose(hipStream_t st, int N, int Nstride, int NPs, const int* obs_img, const int* obs_pt,
                                const unsigned char* pt_free, const double* Jc, const double* Jp, const double* scale_cam,
                                const double* scale_pt, double* Gi, double* h, double* Epose, const int* pt_start,
                                const double* Cu, const double* gu, double radius, double dmin, double dmax, double* fail);
void launch_entries_intr(hipStream_t st, int Q, int NI, int NPs, const int* q_pt, const int* q_cam,
                         const double* Wk, const double* scale_cam, const double* scale_pt,
                         const double* Gi, const double* h, double* Eintr);

void launch_schur_chunks(hipStream_t st, int kind, int num_chunks, const SchurChunk* chunks,
                         const int2* terms, const double* Epose, const double* Eintr,
                         double* partial);
int schur_partial_stride(int kind);
// front end + cluster kernel in one launch (problems whose every observed point is clustered; cluster shape 16 x 3):
// a.sw.cost_partial gets one partial per cluster (k_schur_rows, schur_rows.hip; its round-3 predecessor k_schur_fused lives in
// scripts/_dbg/pruned_r06.patch)
void launch_schur_rows(hipStream_t st, const FrontArgs& a, int kmax_intr, bool generic, int num_clusters,
                       const SchurRowsCluster* clusters, const int* tab, const int* cl_lists, const unsigned short* obs_meta,
                       const unsigned long long* lanemap, const unsigned* emit_map, double* part_pp, double* part_ip, double* part_ii,
                       const struct CamSweepArgs* with_sweep = nullptr /* local windows, constant intrinsics: the camera sweep's chunks as extra work-groups of this launch */);
// The emit map of a cluster shape (host): for pass 0 and pass 1, in the order the lanes walk them, one rows_emit_entry per
// element of the block partials the shape can touch - pose x pose (42 per image pair: the 6 x 6 block, lower triangle only on
// the diagonal, + the h row's 6 on the diagonal), intrinsics x pose (54), intrinsics x intrinsics (90: 9 x 9 + the h row's 9).
void rows_emit_map(int ni, int nc, int flags, std::vector<unsigned>& pass0, std::vector<unsigned>& pass1);
// lanemap: per point, nibble i = which of the point's observations lane i of its 16-lane row takes (kRowsLanesIdentity: lane i takes
// observation i). In a cluster with two camera slots the observations of slot 0 sit in lanes 0-7, those of slot 1 in lanes 8-15.
constexpr unsigned long long kRowsLanesIdentity = 0xFEDCBA9876543210ull;
// obs_meta / q_meta: per observation / intrinsics entry, local index << 8 | (point - cluster.p0) % kClBatch,
// 0xFFFF for records that are not part of a cluster.
void launch_schur_clusters(hipStream_t st, ClusterShape shape, int num_clusters, const SchurCluster* clusters, const int* tab,
                           const int* pt_start, const int* q_start, const unsigned short* obs_meta,
                           const unsigned short* q_meta, const unsigned char* pt_clustered, const double* Epose,
                           const double* Eintr, const double* h, int NPs, double* part_pp, double* part_ip,
                           double* part_ii);
// off_img[i] / off_cam[c]: first matrix column of image i's pose block / camera c's intrinsics block
// (the matrix is assembled in the factorisation's elimination order, the vectors keep the variables' order).
void launch_schur_finalize(hipStream_t st, int num_blocks, const SchurBlock* blocks,
                           const double* part_pp, const double* part_ip, const double* part_ii,
                           int NI, int NC, const int* slot, int nb, bool add_base, double radius, double dmin,
                           double dmax, const int* img_cam, const double* img_rec,
                           const double* cam_rec, const double* scale_cam, const int* off_img, const int* off_cam,
                           double* S);
// Tile store of the reduced system (round 6): the matrix and its factor are kept as the 64 x 64 tiles of the factorisation's
// ENVELOPE only - tile (i, k), k <= i, lives at store + slot[i * nb + k] * 4096 (row-major, pitch 64), slot < 0 = outside the
// envelope (structurally zero, never touched); tile row nb holds the right-hand side in the first row of its tiles.
// CholStructure::tile_slot / d_tile_slot is the table (dense_chol.hip), shared by the assembly, the exchange and the solve.
// pack: tiles[t] (a slot) -> buf[t * 4096 ...] and the right-hand side -> buf[num_tiles * 4096 + c]; unpack: the reverse.
void launch_tiles_copy(hipStream_t st, int num_tiles, const int* tiles, const int* slot, int nb, double* M, double* buf, bool to_buf);
// col_var[t]: variable (index into scale_cam) held by matrix column t, -1 for padding columns.
void launch_fix_diag(hipStream_t st, int n_mat, const int* slot, bool add_one, const int* col_var,
                     const double* scale_cam, double* S);

int backsub_points_grid(int NP);
// the same from recomputed Jacobians (no entry records read); `a`: the sweep arguments of the CURRENT state,
// delta_cam from launch_update_cameras (which runs first). cost_partial != null: the candidate's cost as well (one partial per
// work-group, backsub_points_grid of them; cand_camrec / cand_intr = the candidate cameras) - no separate k_cost_only launch
void launch_backsub_points_jvp(hipStream_t st, int NP, int NPs, int NI, double radius, double dmin, double dmax,
                               const SweepArgs& a, const int* pt_start, const double* delta_cam,
                               const unsigned char* pt_free, const double* Gi, const double* h, const double* Cu,
                               const double* gu, const double* scale_pt, double* cand_points, double* delta_points,
                               double* partial /*[grid][3]*/, const double* cand_camrec = nullptr, const double* cand_intr = nullptr,
                               double* cost_partial = nullptr);
int update_cameras_groups(int NI);  // triples launch_update_cameras writes to partial3
void launch_update_cameras(hipStream_t st, int NI, int NC, bool cam_part, double radius, double dmin,
                           double dmax, const double* y, const double* scale_cam,
                           const double* img_rec, const double* cam_rec, const double* poses,
                           const double* intr, double* cand_poses, double* cand_intr,
                           double* delta_cam, double* partial3 /*[groups][3]*/, double* cand_camrec /* null: no camera records */);

// Deterministic reductions of per-block partials: out[c] = op_c(partial[:, c]).
// op bit c of `max_mask` set -> max, else sum. Adds into out if accumulate.
void launch_reduce_cols(hipStream_t st, const double* partial, int rows, int cols, int stride,
                        unsigned max_mask, double* out, bool accumulate);

struct ReduceTask { const double* src; int rows, stride, is_max; const double* src2; int rows2; double* out; };
struct ReduceTasks { ReduceTask t[6]; };
void launch_reduce_tasks(hipStream_t st, const ReduceTasks& tasks, int n, const LmSpec& spec = lm_spec_off());
// The decision on the candidate whose scalars sit in spec.scal: {code, radius} to dec (device) for the speculative
// evaluation's kernels; the scalars, the decision and - last - `seq` to host_pub (host-mapped, coherent: kLmPubDoubles doubles).
constexpr int kLmPubDoubles = SC_COUNT + 8;  // scalars | code radius decrease_factor rel step_norm cost_change - | seq
void launch_lm_snapshot(hipStream_t st, const LmSpec& spec, double* dec, double* host_pub, double seq, double* fail_slots /* SC_FAIL, SC_FAIL_FRONT: cleared behind the read */);
void launch_lm_decide_cases(hipStream_t st, int n, const double* in, double* out);  // test entry
// k_reduce_tasks + k_lm_snapshot in one work-group (the n reductions one after the other, then the decision)
void launch_lm_tail(hipStream_t st, const ReduceTasks& tasks, int n, const LmSpec& spec, double* dec, double* host_pub, double seq, double* fail_slots);
// The tail of an evaluation in one work-group - k_camera_reduce_img, k_camera_reduce_cam, k_state_norms, k_reduce_tasks - for
// problems whose images, cameras and points one work-group walks in a few microseconds (a local window). Same sums, same order.
void state_norms_grid(int NI, int NC, int NP, int* gp, int* gc);  // point groups + camera groups of launch_state_norms
struct EvalSmallArgs {
  int NI, NC, NP, NPs, with_cams, cam_part, gp, gc, num_tasks;
  int first_group = 0;  // k_eval_tail: the norm groups below this one (the points') were done by k_eval_head
  const int* img_chunk_start; const double* cam_partial; const int* prior_start; const double* prior_res; const double* prior_jac;
  const int* cam_img_start; const int* cam_imgs; double* img_rec; double* cam_rec; double* img_intr_tmp;
  const unsigned char* pose_free; const unsigned char* intr_free; const unsigned char* pt_free;
  const double* poses; const double* intr; const double* points; const double* gu; double* norm_partial;
  ReduceTasks T; LmSpec spec;
  PartialRide ride;  // k_eval_head only
};
void launch_eval_small(hipStream_t st, const EvalSmallArgs& a);
// the same for problems of any size, as TWO launches (round 5): per-image sums and the points' norm groups side by side
// in one grid, then one work-group for the per-camera sums, the cameras' norm groups and the three reductions
void launch_eval_head_tail(hipStream_t st, const EvalSmallArgs& a);
bool eval_head_tail_fits(int gc);

void launch_points_to_caller(hipStream_t st, int NP, int width, const int* orig, const double* in, double* out);
void launch_point_errors(hipStream_t st, int NP, const int* pt_start, const double* rnorm,
                         const int* pt_count, double* perr);

// ---- device memory (host_util.hip) ---------------------------------------------
// Process-wide caching allocator: local BA creates and destroys a session per call (hundreds per run) and
// ~60 hipMalloc/hipFree pairs cost more than the solve of a small window. Blocks are cached per (device, size
// class) and handed out again; contents are NOT cleared. MAVBA_POOL_MB caps the cache (default 16384, 0 = off).
hipError_t device_alloc(void** p, size_t bytes);
void device_free(void* p);
// host_util.hip: pageable host memory <-> device through pooled page-locked blocks (never page-locked in place)
hipError_t copy_h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t st);       // asynchronous; staging released by release_staged
hipError_t copy_d2h_staged_sync(void* dst, const void* src, size_t bytes, hipStream_t st);  // returns with the data in dst
void release_staged(hipStream_t st);  // call behind a synchronisation of st
// Batched small uploads (host_util.hip): between upload_batch_begin(st) and upload_batch_end on ONE host thread, copies
// below 128 KiB through copy_h2d_staged and clears through zero_async on st are recorded and leave as one copy + one
// kernel at the end (emit = false drops them: error paths). Nothing that READS the destinations may be enqueued in between.
bool upload_batch_begin(hipStream_t st);      // false: the thread already has one open (the outer one keeps collecting)
hipError_t upload_batch_end(bool emit);
hipError_t zero_async(void* p, size_t bytes, hipStream_t st);
void upload_batch_debug_arena(size_t bytes);  // tests: arena size of the batches that follow (0 = default)
// host worker threads of the set-up passes (host_util.hip): body(0 .. T-1), T <= host_threads(); calls are serialised, never nest them
int host_threads();
void host_run(int T, const std::function<void(int)>& body);

// ---- dense SPD solve (dense_chol.hip) --------------------------------------
// M: (n_pad + 64) x n_pad row-major, n_pad % 64 == 0. Rows [0, n_pad) hold the
// SPD matrix (lower triangle is read), row n_pad holds the right-hand side.
// On return y[0..n_pad) = A^-1 b; M is overwritten by the factor. *fail
// (device double) is incremented if a pivot is not positive.
// diag_ws: workspace of n_pad * 64 doubles (the inverses of the factor's diagonal tiles).
// L: scratch matrix of the same shape as M (receives the factor).
// CholStructure: tile structure of the matrix (64x64 tiles) and the launch schedule derived from it.
//
// Envelope: inside a segment of tile columns, tile row i is structurally zero left of seg_first[i][seg];
// the factorisation only visits tiles inside the envelope (a skyline per segment).
//
// Segments = nodes of an elimination tree (nested dissection, children before parents): nodes of the same
// level are mutually uncoupled, so their panels are factorised CONCURRENTLY (one "front" per node in the
// same launch); a front's Schur contribution to its ancestors goes to its own shadow block (the first node of
// a level writes the matrix itself), the level's shadows are merged, then the next level runs. The number
// of dependent panel steps drops from nb to the sum over levels of the longest node. Without a tree there is
// one segment and the schedule is the plain right-looking chain over the envelope.
struct CholFront {
  int k;             // panel (tile column) this front eliminates
  int na;            // active row blocks below tile k (rows + act_off)
  int act_off;
  int factor_next;   // the owner of tile (k+1, k+1) factorises it for the front's next step
  int sh_begin;      // updates of tiles in columns >= sh_begin go to the front's shadow block (if it has one)
  long long sh_off;  // offset (doubles) of that shadow block, -1: everything goes to the matrix
};
// One launch group of the schedule. kind 0: panel step of `nf` concurrent fronts with na > 0 followed by nf0 with
// na == 0 (fronts + front_off). kind 1: end of a tree level - merge the level's shadow blocks (merges +
// front_off .. + nf) into the matrix over tiles >= merge_begin, then factorise the diagonal tiles listed at
// init + init_off .. + nf0 (the first tiles of the next level's nodes).
struct CholStep { int kind, front_off, nf, nf0, max_na, merge_begin, init_off, tasks; };  // tasks = tile updates of the step, all fronts
struct CholMerge { int sh_begin; long long sh_off; };
// Node of the elimination tree: tile columns [begin, end); parent = index of the separator it hangs under
// (-1: root). Nodes are listed in column order, children before parents.
struct CholNode { int begin, end, parent; };
// Persistent schedule (k_chol_persist, dense_chol.hip): a task is one tile owned by one work-group (TILE: apply the
// updates upd[ub, ue), solve against L_jj^-1, publish; PRE_*: a chain tile with every update but the chain's own) or
// the diagonal of one tree node (CHAIN: tile columns [i, j)).
enum { CHOL_TASK_TILE = 0, CHOL_TASK_PRE_DIAG = 1, CHOL_TASK_PRE_SUB = 2, CHOL_TASK_CHAIN = 3 };
struct CholTask { int kind, i, j, ub, ue; };
struct CholStructure {
  int nb = 0, nseg = 1;
  std::vector<CholNode> nodes;
  std::vector<int> seg_of_tile;   // [nb]
  std::vector<int> seg_first;     // [nb * nseg]
  std::vector<CholFront> fronts;
  std::vector<CholMerge> merges;
  std::vector<CholStep> steps;
  std::vector<int> init_tiles;    // diagonal tiles factorised before the first step / after the merges
  int num_leaf_init = 0;          // the first num_leaf_init entries of init_tiles start the schedule
  int chain_steps = 0;            // dependent panel steps of the schedule
  int num_fronts_max = 0;         // widest level
  long long envelope_tiles = 0;   // lower-triangle tiles visited (incl. diagonal)
  double factor_flops = 0.0;      // MFMA flops executed by the factorisation
  size_t shadow_doubles = 0;      // total size of the shadow blocks
  int* d_ints = nullptr;          // one allocation: rows | seg_of_tile | seg_first | init_tiles | flags
  int *d_rows = nullptr, *d_seg_of_tile = nullptr, *d_seg_first = nullptr, *d_init = nullptr;
  unsigned* d_flags = nullptr;    // [nb] 'solution segment published' flags of the backward substitution
  CholFront* d_fronts = nullptr;
  CholMerge* d_merges = nullptr;
  mutable double* d_shadow = nullptr;  // launch-per-panel schedule only: allocated by the first solve that takes it (C5: 5 GB, 10 000 images: 40 GB - a persistent session never needs it)
  // Tile store (round 6): slot of tile (i, k) = tile_slot[i * nb + k], i in [0, nb] (row nb: the right-hand side), -1 outside the
  // envelope; per column k: the diagonal tile, the coupled rows ascending, the right-hand-side tile. Also the index of the
  // tile's 'published' flag in the persistent launch.
  std::vector<int> tile_slot;
  int* d_tile_slot = nullptr;
  long long num_tiles = 0;
  size_t store_doubles() const { return (size_t)std::max<long long>(num_tiles, 1) * 4096; }
  // persistent schedule
  int active_tiles = 0;  // leading tile columns that hold a free parameter (0: all); the rest is identity with a zero right-hand side
  bool persist_ok = false;        // a schedule exists (structure consistent, fits the resident grid)
  double predicted_forward_us = 0.0;  // the timing model's forward factorisation (build_persistent)
  int persist_grid = 0;           // work-groups (<= CUs of the device, all resident)
  int persist_chain_wgs = 0;      // the first persist_chain_wgs work-groups walk the nodes' diagonals
  long long persist_tiles = 0;    // tiles with a 'published' flag
  long long persist_updates = 0;  // tile updates of one factorisation
  CholTask* d_tasks = nullptr;
  int* d_pints = nullptr;         // wg_begin | upd | tile_id | chain_info
  int *d_wg_begin = nullptr, *d_upd = nullptr, *d_tile_id = nullptr, *d_chain_info = nullptr;
  unsigned* d_pflags = nullptr;   // lflag[persist_tiles] | dflag[nb] | pflag[2 nb] | abort[1]
  double* d_pre = nullptr;        // [2 nb] 64x64 tiles: the chain's tiles after the helpers' updates
  unsigned long long* d_trace = nullptr;  // MAVBA_CHOL_TRACE=<file>: stamps of the last solve, dumped on release
  std::vector<CholTask> h_tasks;          // (kept for the trace dump and the host-only test entry)
  std::vector<int> h_wg_begin, h_upd, h_chain_info;
  // Host-only build (mavba_debug_chol_schedule, tests without a GPU): the structure and the persistent schedule are computed,
  // nothing is allocated on or copied to a device; `host_only_cus` stands for the CU count.
  bool host_only = false;
  int host_only_cus = 256;
  double lpp_estimate_us = 0.0;           // the timing model's estimate of the launch-per-panel schedule
  mutable unsigned epoch = 0;     // flags are compared with the solve's epoch: nothing is cleared between solves
  CholStructure() {}
  CholStructure(const CholStructure&) = delete;
  CholStructure& operator=(const CholStructure&) = delete;
  ~CholStructure();
  void release();
  // tile_pairs: (row tile, col tile), row >= col, of every structurally non-zero tile (diagonal tiles are
  // implied). tree: elimination tree over contiguous tile ranges covering [0, nb) (empty = one node = plain
  // chain over the envelope). Leaves keep a skyline envelope; separators are treated as dense.
  hipError_t build(int nb, const std::vector<std::pair<int, int>>& tile_pairs, const std::vector<CholNode>& tree,
                   hipStream_t st);
  hipError_t build_dense(int nb);
  hipError_t build_persistent(const std::vector<std::vector<int>>& col_rows, const std::vector<int>& col_step,
                              const std::vector<int>& height, hipStream_t st);
};
// y_scatter (may be null): y_nat[y_scatter[t]] = y[t] for every t with y_scatter[t] >= 0 (the solution in
// the caller's variable order when the matrix was assembled in a permuted order).
// upd (may be null): the camera update that follows the solve in the LM loop (launch_update_cameras' arguments; y = y_nat).
// A system of one or two tile columns is solved by ONE work-group, which then applies the update as well: the return value
// says whether it did (the caller skips its own launch).
struct CamUpdateArgs {
  int NI, NC, cam_part; double radius, dmin, dmax;
  const double* scale_cam; const double* img_rec; const double* cam_rec; const double* poses; const double* intr;
  double* cand_poses; double* cand_intr; double* delta_cam; double* partial3; double* cand_camrec;
};
bool dense_spd_solve_is_small(int n_pad, const CholStructure& cs);  // the one-work-group path will be taken
// M, L: tile stores of cs (cs.store_doubles() doubles each).
bool dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* diag_ws, double* L, const CholStructure& cs,
                            const int* y_scatter = nullptr, double* y_nat = nullptr, bool allow_persistent = true,
                            hipEvent_t after_factor = nullptr /* recorded between the factorisation and the backward substitution */,
                            const CamUpdateArgs* upd = nullptr);

}  // namespace mavba
#endif
