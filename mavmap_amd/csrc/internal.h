// internal.h — shared declarations between the HIP kernel files and the host LM driver.
#ifndef MAVBA_INTERNAL_H_
#define MAVBA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

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
struct SchurChunk { int begin; int end; };  // term range

// Scalars exchanged with the host every LM iteration (device array of doubles).
// [0, SC_NUM_SUMS) are summed over ranks, SC_GRAD_MAX is max-reduced.
enum {
  SC_COST = 0,        // 1/2 sum rho at the evaluation point (no fixed cost)
  SC_XNORM2,          // |x|^2 over free parameters
  SC_NEW_COST,        // candidate cost
  SC_STEP_NORM2,      // |delta|^2
  SC_MODEL_CHANGE,    // model cost change
  SC_CAND_XNORM2,     // |x + delta|^2
  SC_FAIL,            // > 0: linear solve failed (non-SPD point block or pivot)
  SC_NUM_SUMS,
  SC_GRAD_MAX = SC_NUM_SUMS,  // max |g_j| over free parameters (unscaled gradient); directly after the sums
  SC_COUNT = 16
};

// ---- launch wrappers (kernels.hip) -----------------------------------------
struct SweepArgs {
  int N, Nstride, NI, NC, KMAX;
  const double2* uv; const int* obs_img; const int* obs_pt;
  const double* camrec; const double* intr; const int* img_cam; const int* cam_model;
  const double* points;
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

struct CamSweepArgs {
  int NI, NC;
  const SweepChunk* chunks; int num_chunks;
  const double2* im_uv; const int* im_pt;
  const double* camrec; const double* intr; const int* img_cam; const int* cam_model;
  const double* points; double loss_b, loss_inv_b;
  double* partial;  // [num_chunks][kSweepAcc]
};
void launch_camera_sweep(hipStream_t st, const CamSweepArgs& a, int kmax, bool any_intr_free);
void launch_camera_reduce(hipStream_t st, int NI, int NC, const int* img_chunk_start,
                          const double* partial, const int* prior_start, const double* prior_res,
                          const double* prior_jac, const int* cam_img_start, const int* cam_imgs,
                          double* img_rec, double* cam_rec, double* img_intr_tmp);
void launch_rot_prior(hipStream_t st, int n, const int* prior_img, const double* prior_R0, double w,
                      const double* poses, double* res, double* jac, double* cost_partial);

void launch_scales(hipStream_t st, int NI, int NC, int NP, int NPs, int jacobi,
                   const unsigned char* pose_free, const unsigned char* intr_free,
                   const unsigned char* pt_free, const double* img_rec, const double* cam_rec,
                   const double* Cu, double* scale_cam, double* scale_pt);

void launch_state_norms(hipStream_t st, int NI, int NC, int NP, int NPs, bool cam_part,
                        const unsigned char* pose_free, const unsigned char* intr_free,
                        const unsigned char* pt_free, const double* poses, const double* intr,
                        const double* points, const double* img_rec, const double* cam_rec,
                        const double* gu, double* partial /*[grid][2]*/, int* grid_out);

void launch_point_factor(hipStream_t st, int NP, int NPs, double radius, double dmin, double dmax,
                         const unsigned char* pt_free, const double* Cu, const double* gu,
                         const double* scale_pt, double* Gi, double* h, double* fail);

void launch_entries_pose(hipStream_t st, int N, int Nstride, int NPs, const int* obs_img,
                         const int* obs_pt, const unsigned char* pt_free, const double* Jc,
                         const double* Jp, const double* scale_cam, const double* scale_pt,
                         const double* Gi, const double* h, double* Epose);
void launch_entries_intr(hipStream_t st, int Q, int NI, int NPs, const int* q_pt, const int* q_cam,
                         const double* Wk, const double* scale_cam, const double* scale_pt,
                         const double* Gi, const double* h, double* Eintr);

void launch_schur_chunks(hipStream_t st, int kind, int num_chunks, const SchurChunk* chunks,
                         const int2* terms, const double* Epose, const double* Eintr,
                         double* partial);
int schur_partial_stride(int kind);
void launch_schur_finalize(hipStream_t st, int num_blocks, const SchurBlock* blocks,
                           const double* part_pp, const double* part_ip, const double* part_ii,
                           int NI, int NC, int ld, bool add_base, double radius, double dmin,
                           double dmax, const int* img_cam, const double* img_rec,
                           const double* cam_rec, const double* scale_cam, double* S, double* v);
void launch_fix_diag(hipStream_t st, int n_full, int n_pad, int ld, bool add_one,
                     const double* scale_cam, double* S);

void launch_backsub_points(hipStream_t st, int NP, int NPs, int NI, double radius, double dmin,
                           double dmax, const int* pt_start, const int* obs_img, const int* q_start,
                           const int* q_cam, const unsigned char* pt_free, const double* Epose,
                           const double* Eintr, const double* y, const double* Gi, const double* h,
                           const double* Cu, const double* gu, const double* scale_pt,
                           const double* points, double* cand_points, double* delta_points,
                           double* partial /*[grid][3]*/, int* grid_out);
void launch_update_cameras(hipStream_t st, int NI, int NC, bool cam_part, double radius, double dmin,
                           double dmax, const double* y, const double* scale_cam,
                           const double* img_rec, const double* cam_rec, const double* poses,
                           const double* intr, double* cand_poses, double* cand_intr,
                           double* delta_cam, double* partial3 /*[3]*/);

// Deterministic reductions of per-block partials: out[c] = op_c(partial[:, c]).
// op bit c of `max_mask` set -> max, else sum. Adds into out if accumulate.
void launch_reduce_cols(hipStream_t st, const double* partial, int rows, int cols, int stride,
                        unsigned max_mask, double* out, bool accumulate);

void launch_point_errors(hipStream_t st, int NP, const int* pt_start, const double* rnorm,
                         const int* pt_count, double* perr);

// ---- dense SPD solve (dense_chol.hip) --------------------------------------
// M: (n_pad + 64) x n_pad row-major, n_pad % 64 == 0. Rows [0, n_pad) hold the
// SPD matrix (lower triangle is read), row n_pad holds the right-hand side.
// On return y[0..n_pad) = A^-1 b; M is overwritten by the factor. *fail
// (device double) is incremented if a pivot is not positive.
// diag_ws: workspace of n_pad * 64 doubles (the inverses of the factor's diagonal tiles).
// L: scratch matrix of the same shape as M (receives the factor).
// CholStructure: tile envelope (skyline) of the matrix, 64x64 tiles: first[i] = first structurally
// non-zero tile of tile row i. The factorisation only visits tiles inside the envelope.
struct CholStructure {
  int nb = 0;
  std::vector<int> first;  // [nb]
  std::vector<int> off;    // [nb + 1] into d_rows: active row blocks of every panel
  int* d_rows = nullptr;
  int* d_first = nullptr;        // [nb] copy of first[] (inside the d_rows allocation)
  unsigned* d_flags = nullptr;   // [nb] per-tile 'solution segment published' flags of the backward substitution
  CholStructure() {}
  CholStructure(const CholStructure&) = delete;
  CholStructure& operator=(const CholStructure&) = delete;
  ~CholStructure();
  void build(int nb, const std::vector<int>& first_tile, hipStream_t st);
  void build_dense(int nb);
};
void dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* diag_ws, double* L, const CholStructure& cs);

}  // namespace mavba
#endif
