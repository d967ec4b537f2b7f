// lm_bodies.h - the bodies of the LM loop's small kernels as device functions (round 4).
//
// A 10-image local window (the call MAVMAP issues after every image, reference src/mapper.cc:1120-1135) spends its
// iteration in launches of 4-8 us whose work is a fraction of that: per-image sums, norms, scalar reductions, the 128 x 128
// factorisation, the camera update, the decision. Each body below is what one work-group (or one wave) of the kernel of
// the same name does, with the work-group index as a parameter, so that
//   - the multi-work-group kernels (kernels.hip) call it with blockIdx.x - large problems are unchanged -, and
//   - the merged single-work-group kernels (k_eval_small, k_chol_small<true>, k_lm_tail) walk the same bodies in a loop:
//     the same additions in the same order, bit-identical results, one launch instead of three to four.
// Inside a merged kernel a body reads what an earlier body of the SAME work-group wrote to global memory (behind a
// __syncthreads()): no pointer here is __restrict__, so the compiler can not turn such a read into a scalar / invariant load.
#ifndef MAVBA_LM_BODIES_H_
#define MAVBA_LM_BODIES_H_
#include "internal.h"
#include "ba_math.h"
#include "dev_reduce.h"
#include "lm_decide.h"

namespace mavba {

// Per image: sum its chunks (fixed order) + its rotation priors -> img_rec[81] and the image's intrinsics part ->
// img_intr_tmp[54]. One element e (< kSweepAcc) of image i; k_camera_reduce_img gives an image to a 64-lane group.
__device__ __forceinline__ void camera_reduce_img_elem(int i, int e, const int* img_chunk_start, const double* partial,
                                                        const int* prior_start, const double* prior_res, const double* prior_jac,
                                                        double* img_rec, double* img_intr_tmp) {
  const int c0 = img_chunk_start[i], c1 = img_chunk_start[i + 1];
  double s = 0.0;
  int c = c0;
  for (; c + 4 <= c1; c += 4) {  // (four loads in flight; the additions keep their order)
    const double x0 = partial[(size_t)c * kSweepAcc + e], x1 = partial[(size_t)(c + 1) * kSweepAcc + e];
    const double x2 = partial[(size_t)(c + 2) * kSweepAcc + e], x3 = partial[(size_t)(c + 3) * kSweepAcc + e];
    s += x0; s += x1; s += x2; s += x3;
  }
  for (; c < c1; ++c) s += partial[(size_t)c * kSweepAcc + e];
  if (e < 27 && prior_start) {
    for (int q = prior_start[i]; q < prior_start[i + 1]; ++q) {
      const double* j = prior_jac + 3 * q;
      if (e < 21) {
        // PP upper-triangle slots touching the rvec block: (x,y) with y < 3
        int x = 0, rem = e;
        while (rem >= 6 - x) { rem -= 6 - x; ++x; }
        const int y = x + rem;
        if (y < 3) s += j[x] * j[y];
      } else if (e - 21 < 3) {
        s += j[e - 21] * prior_res[q];
      }
    }
  }
  if (e < kImgRec) img_rec[(size_t)i * kImgRec + e] = s;
  else img_intr_tmp[(size_t)i * kCamRec + (e - kImgRec)] = s;
}
__device__ __forceinline__ void camera_reduce_img_body(int i, int lane, const int* img_chunk_start, const double* partial,
                                                       const int* prior_start, const double* prior_res, const double* prior_jac,
                                                       double* img_rec, double* img_intr_tmp) {
  for (int e = lane; e < kSweepAcc; e += 64)
    camera_reduce_img_elem(i, e, img_chunk_start, partial, prior_start, prior_res, prior_jac, img_rec, img_intr_tmp);
}

// Element e of the `part`-th of the 16 interleaved partial sums k_camera_reduce_cam forms per camera (images
// start + part, + 16, + 32, ... in that order); the camera's value is their sum in the order 0..15.
__device__ __forceinline__ double camera_reduce_cam_part(int c, int part, int e, const int* cam_img_start, const int* cam_imgs,
                                                         const double* img_intr_tmp) {
  double s = 0.0;
  const int t1 = cam_img_start[c + 1];
  for (int t = cam_img_start[c] + part; t < t1; t += 16) s += img_intr_tmp[(size_t)cam_imgs[t] * kCamRec + e];
  return s;
}

// Work-group vb of k_state_norms' grid of gp + gc groups: max |g| and |x|^2 over its slice of the free parameters.
// state_norms_local: what lane lt (0..255) of that group accumulates.
__device__ __forceinline__ void state_norms_local(int vb, int lt, int gp, int gc, int NI, int NC, int NP, int NPs, int cam_part,
                                                  const unsigned char* pose_free, const unsigned char* intr_free,
                                                  const unsigned char* pt_free, const double* poses, const double* intr,
                                                  const double* points, const double* img_rec, const double* cam_rec,
                                                  const double* gu, double& gmax, double& x2) {
  gmax = 0.0; x2 = 0.0;
  if (vb < gp) {
    for (int p = vb * 256 + lt; p < NP; p += gp * 256) {
      if (!pt_free[p]) continue;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gmax = fmax(gmax, fabs(gu[k * NPs + p]));
        const double x = points[3 * (size_t)p + k];
        x2 += x * x;
      }
    }
  } else {
    const int ncam = 6 * NI + 9 * NC;
    for (int t = (vb - gp) * 256 + lt; t < ncam; t += gc * 256) {
      double g, x; bool fr;
      if (t < 6 * NI) {
        fr = pose_free[t] != 0; g = img_rec[(size_t)(t / 6) * kImgRec + 21 + t % 6]; x = poses[t];
      } else {
        const int c = (t - 6 * NI) / 9, k = (t - 6 * NI) % 9;
        fr = intr_free[9 * c + k] != 0; g = cam_rec[(size_t)c * kCamRec + 45 + k]; x = intr[9 * c + k];
      }
      if (!fr) continue;
      gmax = fmax(gmax, fabs(g));        // gradient is global after the camera-sum all-reduce
      if (cam_part) x2 += x * x;         // counted on one rank only
    }
  }
}
__device__ __forceinline__ void state_norms_body(int vb, int gp, int gc, int NI, int NC, int NP, int NPs, int cam_part,
                                                 const unsigned char* pose_free, const unsigned char* intr_free,
                                                 const unsigned char* pt_free, const double* poses, const double* intr,
                                                 const double* points, const double* img_rec, const double* cam_rec,
                                                 const double* gu, double* partial, double* s_red) {
  double gmax, x2;
  state_norms_local(vb, threadIdx.x, gp, gc, NI, NC, NP, NPs, cam_part, pose_free, intr_free, pt_free, poses, intr, points, img_rec, cam_rec,
                    gu, gmax, x2);
  const double m = block_max_256(gmax, s_red);
  const double s = block_sum_256(x2, s_red);
  if (threadIdx.x == 0) { partial[2 * vb] = m; partial[2 * vb + 1] = s; }
}

// The single-work-group kernels run FOUR 256-lane groups side by side (1024 lanes): group g = threadIdx.x >> 8 does what a
// 256-lane work-group does, its four waves leave their wave_sum / wave_max in w[0..3], and the combination below is
// block_sum_256's / block_max_256's - the same operations in the same order without their two barriers per reduction.
__device__ __forceinline__ double group4_sum(const double* w) { return (w[0] + w[1]) + (w[2] + w[3]); }
__device__ __forceinline__ double group4_max(const double* w) { return fmax(fmax(w[0], w[1]), fmax(w[2], w[3])); }

// One single-work-group reduction: out = op(src[r * stride], r < rows) (+ sum of src2[r], r < rows2).
__device__ __forceinline__ void reduce_task_body(const ReduceTask& t, double* s_red) {
  double v = 0.0;
  for (int r = threadIdx.x; r < t.rows; r += 256) {
    const double x = t.src[(size_t)r * t.stride];
    v = t.is_max ? fmax(v, x) : v + x;
  }
  double total = t.is_max ? block_max_256(v, s_red) : block_sum_256(v, s_red);
  if (t.rows2 > 0) {
    __syncthreads();
    double w = 0.0;
    for (int r = threadIdx.x; r < t.rows2; r += 256) w += t.src2[r];
    total += block_sum_256(w, s_red);
  }
  if (threadIdx.x == 0) *t.out = total;
}

// Up to four of those reductions side by side in a 1024-lane work-group (group g takes task g); s_t: [4][2][4] doubles of LDS.
// Ends with the results in memory and a barrier behind them.
__device__ __forceinline__ void reduce_tasks_grouped(const ReduceTasks& T, int n, double (*s_t)[2][4]) {
  const int tid = threadIdx.x, g = tid >> 8, lt = tid & 255, wv = (tid >> 6) & 3, lane = tid & 63;
  if (g < n) {
    const ReduceTask t = T.t[g];
    double v = 0.0;
    for (int r = lt; r < t.rows; r += 256) {
      const double x = t.src[(size_t)r * t.stride];
      v = t.is_max ? fmax(v, x) : v + x;
    }
    const double red = t.is_max ? wave_max(v) : wave_sum(v);
    double w = 0.0;
    for (int r = lt; r < t.rows2; r += 256) w += t.src2[r];
    const double red2 = wave_sum(w);
    if (lane == 0) { s_t[g][0][wv] = red; s_t[g][1][wv] = red2; }
  }
  __syncthreads();
  if (tid < n) {
    const ReduceTask t = T.t[tid];
    double total = t.is_max ? group4_max(s_t[tid][0]) : group4_sum(s_t[tid][0]);
    if (t.rows2 > 0) total += group4_sum(s_t[tid][1]);
    *t.out = total;
  }
  __syncthreads();
}

// Camera columns: step = -y, delta = s * step, candidate parameters. Work-group vb takes the images [42 vb, 42 vb + 42) -
// whole pose blocks, so that it can write their camera records too - and group 0 the intrinsics blocks as well; every group
// leaves one (|delta|^2, model change, |x + delta|^2) triple.
constexpr int kUpdImagesPerGroup = 42;  // 252 pose parameters: one trip of a 256-thread work-group
__device__ __forceinline__ void update_cameras_body(int vb, int NI, int NC, int cam_part, double radius, double dmin, double dmax,
                                                    const double* y, const double* scale_cam, const double* img_rec,
                                                    const double* cam_rec, const double* poses, const double* intr,
                                                    double* cand_poses, double* cand_intr, double* delta_cam, double* partial3,
                                                    double* cand_camrec, double* s_red) {
  const int i0 = vb * kUpdImagesPerGroup, i1 = min(i0 + kUpdImagesPerGroup, NI);
  double a_step = 0.0, a_model = 0.0, a_x2 = 0.0;
  auto one = [&](int t) {
    const double s = scale_cam[t];
    double n2, g, x;
    if (t < 6 * NI) {
      const int i = t / 6, e = t % 6;
      n2 = img_rec[(size_t)i * kImgRec + sym_idx(e, e, 6)]; g = img_rec[(size_t)i * kImgRec + 21 + e]; x = poses[t];
    } else {
      const int c = (t - 6 * NI) / 9, k = (t - 6 * NI) % 9;
      n2 = cam_rec[(size_t)c * kCamRec + sym_idx(k, k, 9)]; g = cam_rec[(size_t)c * kCamRec + 45 + k]; x = intr[9 * c + k];
    }
    double d = 0.0;
    if (s != 0.0) {
      const double yy = y[t];
      const double D2 = clampd(s * s * n2, dmin, dmax) / radius;
      d = -yy * s;
      if (cam_part) { a_model += 0.5 * yy * (s * g + D2 * yy); a_step += d * d; }
    }
    const double xn = x + d;
    if (s != 0.0 && cam_part) a_x2 += xn * xn;
    delta_cam[t] = d;
    if (t < 6 * NI) cand_poses[t] = xn; else cand_intr[t - 6 * NI] = xn;
  };
  for (int t = 6 * i0 + threadIdx.x; t < 6 * i1; t += 256) one(t);
  if (vb == 0)
    for (int t = 6 * NI + threadIdx.x; t < 6 * NI + 9 * NC; t += 256) one(t);
  const double s0 = block_sum_256(a_step, s_red);
  const double s1 = block_sum_256(a_model, s_red);
  const double s2 = block_sum_256(a_x2, s_red);
  if (threadIdx.x == 0) { partial3[3 * vb] = s0; partial3[3 * vb + 1] = s1; partial3[3 * vb + 2] = s2; }
  // the candidate's camera records of this group's images (k_cam_prepare's work; the stores above are the group's own)
  if (cand_camrec) {
    __syncthreads();
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
      double rec[9];
      cam_prepare(cand_poses + 6 * i, rec);
#pragma unroll
      for (int k = 0; k < 9; ++k) cand_camrec[9 * i + k] = rec[k];
    }
  }
}

// One wave: the LM decision on the device (for the speculative evaluation behind it) and the scalars to the host. Every lane
// evaluates the decision (uniform) and stores ONE word of the publication - 24 stores across the host link in parallel
// (one lane writing them in turn took 16 us) -, the sequence number follows behind a system-scope fence.
__device__ __forceinline__ void lm_snapshot_body(int lane, const LmSpec& spec, double* dec, double* host_pub, double seq, double* fail_slots) {
  const LmDecision d = lm_decide(spec.scal, spec);
  if (lane == 0) { dec[0] = (double)d.code; dec[1] = d.radius; }
  volatile double* out = host_pub;
  double v = 0.0;
  if (lane < SC_COUNT) v = spec.scal[lane];
  else if (lane == SC_COUNT) v = (double)d.code;
  else if (lane == SC_COUNT + 1) v = d.radius;
  else if (lane == SC_COUNT + 2) v = d.decrease_factor;
  else if (lane == SC_COUNT + 3) v = d.rel;
  else if (lane == SC_COUNT + 4) v = d.step_norm;
  else if (lane == SC_COUNT + 5) v = d.cost_change;
  if (lane < SC_COUNT + 6) out[lane] = v;
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) out[SC_COUNT + 7] = seq;
  // every lane has read the scalars: the two failure slots start the next linear solve clean (the speculative front end
  // behind this kernel and the factorisation after it add to them) - the two fill operations per iteration are gone
  if (lane < 2 && fail_slots) fail_slots[lane] = 0.0;
}

}  // namespace mavba
#endif
