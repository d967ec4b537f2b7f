// sweep_body.h - the body of the camera sweep's vector kernel (k_camera_sweep<K>, kernels.hip), shared with k_schur_rows
// (schur_rows.hip), which runs it in extra work-groups of its own launch for local windows (round 6: the two kernels are
// independent passes over the same state; one launch instead of two on a 65 us iteration). 256 lanes per chunk.
#ifndef MAVBA_SWEEP_BODY_H_
#define MAVBA_SWEEP_BODY_H_
#include "internal.h"
#include "ba_math.h"
#include "dev_reduce.h"

namespace mavba {

template <int K>
__device__ __forceinline__ void camera_sweep_body(const CamSweepArgs& a, int chunk, double* s_red /* LDS, 4 * kSweepAcc doubles */) {
  const SweepChunk ch = a.chunks[chunk];
  const int cam = a.img_cam[ch.image];
  const int model = a.cam_model[cam];
  double rec[9], kin[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) rec[k] = a.camrec[9 * ch.image + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) kin[k] = a.intr[9 * cam + k];
  constexpr int KK = K > 0 ? K : 1;
  double aPP[21], aPg[6], aPI[6 * KK], aII[KK * (KK + 1) / 2], aIg[KK];
#pragma unroll
  for (int i = 0; i < 21; ++i) aPP[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) aPg[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6 * KK; ++i) aPI[i] = 0.0;
#pragma unroll
  for (int i = 0; i < KK * (KK + 1) / 2; ++i) aII[i] = 0.0;
#pragma unroll
  for (int i = 0; i < KK; ++i) aIg[i] = 0.0;

  // The kernel holds ~120 accumulators per lane (one wave per SIMD), so nothing hides the dependent loads
  // observation -> point index -> point: they are issued two / one trips ahead by hand.
  int o = ch.begin + threadIdx.x;
  double2 m = make_double2(0.0, 0.0), m_n = m;
  int pt_n = 0;
  double X[3] = {0.0, 0.0, 1.0};
  double act = 1.0, act_n = 1.0;
  if (o < ch.end) {
    m = a.im_uv[o];
    const int pt = a.im_pt[o];
    X[0] = a.points[3 * (long long)pt]; X[1] = a.points[3 * (long long)pt + 1]; X[2] = a.points[3 * (long long)pt + 2];
    if (a.pt_active) act = a.pt_active[pt] ? 1.0 : 0.0;
  }
  if (o + 256 < ch.end) { m_n = a.im_uv[o + 256]; pt_n = a.im_pt[o + 256]; }
  for (; o < ch.end; o += 256) {
    double Xn[3] = {0.0, 0.0, 1.0};
    double2 m_nn = make_double2(0.0, 0.0);
    int pt_nn = 0;
    if (o + 256 < ch.end) {
      Xn[0] = a.points[3 * (long long)pt_n]; Xn[1] = a.points[3 * (long long)pt_n + 1]; Xn[2] = a.points[3 * (long long)pt_n + 2];
      if (a.pt_active) act_n = a.pt_active[pt_n] ? 1.0 : 0.0;
    }
    if (o + 512 < ch.end) { m_nn = a.im_uv[o + 512]; pt_nn = a.im_pt[o + 512]; }
    double r[2], Jc[12], Jp[6], Jk[18];
    obs_jacobian(model, rec, kin, X, m.x, m.y, r, Jc, Jp, Jk);
    double w, half_rho;
    cauchy_weight(r[0] * r[0] + r[1] * r[1], a.loss_b, a.loss_inv_b, w, half_rho);
    const double w2 = act * w * w;  // every product below carries two weighted factors (0 for a filtered point)
#pragma unroll
    for (int x = 0; x < 6; ++x) {
#pragma unroll
      for (int y = x; y < 6; ++y)
        aPP[sym_idx(x, y, 6)] += w2 * (Jc[x] * Jc[y] + Jc[6 + x] * Jc[6 + y]);
      aPg[x] += w2 * (Jc[x] * r[0] + Jc[6 + x] * r[1]);
    }
    if constexpr (K > 0) {
#pragma unroll
      for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int k = 0; k < K; ++k) aPI[x * K + k] += w2 * (Jc[x] * Jk[k] + Jc[6 + x] * Jk[9 + k]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int l = k; l < K; ++l) aII[sym_idx(k, l, K)] += w2 * (Jk[k] * Jk[l] + Jk[9 + k] * Jk[9 + l]);
        aIg[k] += w2 * (Jk[k] * r[0] + Jk[9 + k] * r[1]);
      }
    }
    X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
    act = act_n;
    m = m_n; m_n = m_nn; pt_n = pt_nn;
  }
  // block reduction into the fixed 135-slot layout
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * kSweepAcc; i += 256) s_red[i] = 0.0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 21; ++i) { const double v = wave_sum(aPP[i]); if (lane == 0) s_red[wv * kSweepAcc + i] = v; }
#pragma unroll
  for (int i = 0; i < 6; ++i) { const double v = wave_sum(aPg[i]); if (lane == 0) s_red[wv * kSweepAcc + 21 + i] = v; }
  if constexpr (K > 0) {
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int k = 0; k < K; ++k) { const double v = wave_sum(aPI[x * K + k]); if (lane == 0) s_red[wv * kSweepAcc + 27 + x * 9 + k] = v; }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int l = k; l < K; ++l) { const double v = wave_sum(aII[sym_idx(k, l, K)]); if (lane == 0) s_red[wv * kSweepAcc + 81 + sym_idx(k, l, 9)] = v; }
#pragma unroll
    for (int k = 0; k < K; ++k) { const double v = wave_sum(aIg[k]); if (lane == 0) s_red[wv * kSweepAcc + 126 + k] = v; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSweepAcc; i += 256)
    a.partial[(size_t)chunk * kSweepAcc + i] =
        (s_red[i] + s_red[kSweepAcc + i]) + (s_red[2 * kSweepAcc + i] + s_red[3 * kSweepAcc + i]);
}

}  // namespace mavba
#endif  // MAVBA_SWEEP_BODY_H_
