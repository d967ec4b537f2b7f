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

// ---- free intrinsics: the Gram form (k_camera_sweep_gram<K>) ----
typedef double cs_d4 __attribute__((ext_vector_type(4)));
constexpr int kCsPitch = 34;
constexpr int kCsLdsDoubles = 4 * 64 * kCsPitch;  // a work-group's rows: four waves x 64 lanes
// One wave's pass over its batches of the chunk (observations ch.begin + threadIdx.x + 256 j). MODEL is a compile-time
// constant and the look-ahead loads use clamped indices, so the loop body is ONE basic block: the scheduler lays the 32
// matrix instructions of batch j (operands from LDS) between the Jacobian arithmetic of batch j + 1 (rows kept in
// registers until the matrix instructions have read the previous ones).
template <int K, int MODEL>
__device__ __forceinline__ void camera_gram_wave(const CamSweepArgs& a, const SweepChunk ch, const double (&rec)[9],
                                                 const double (&kin)[9], double* __restrict__ wr,
                                                 const double* __restrict__ rd, cs_d4 (&acc)[4]) {
  const int last = ch.end - 1;
  auto rows = [&](int o, double2 m, const double (&X)[3], double act, double (&v)[32]) {
    double r[2], Jc[12], Jp[6], Jk[18];
    obs_jacobian(MODEL, rec, kin, X, m.x, m.y, r, Jc, Jp, Jk);
    double w, half_rho;
    cauchy_weight(r[0] * r[0] + r[1] * r[1], a.loss_b, a.loss_inv_b, w, half_rho);
    const double ws = (o <= last) ? act * w : 0.0;  // each row carries one weight factor (0: tail lane / filtered point)
#pragma unroll
    for (int x = 0; x < 6; ++x) { v[x] = ws * Jc[x]; v[16 + x] = ws * Jc[6 + x]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) { v[6 + k] = k < K ? ws * Jk[k] : 0.0; v[22 + k] = k < K ? ws * Jk[9 + k] : 0.0; }
    v[15] = ws * r[0]; v[31] = ws * r[1];
  };
  auto to_lds = [&](const double (&v)[32]) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the matrix instructions of the previous batch have taken their operands
#pragma unroll
    for (int e = 0; e < 32; e += 2) *reinterpret_cast<double2*>(wr + e) = make_double2(v[e], v[e + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto gram = [&]() {
#pragma unroll
    for (int q = 0; q < 32; q += 4) {
      const double x0 = rd[(2 * q) * kCsPitch], x1 = rd[(2 * q + 2) * kCsPitch], x2 = rd[(2 * q + 4) * kCsPitch], x3 = rd[(2 * q + 6) * kCsPitch];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x3, acc[3], 0, 0, 0);
    }
  };
  auto load_obs = [&](int o, double2& m, int& pt) { const int oc = min(o, last); m = a.im_uv[oc]; pt = a.im_pt[oc]; };
  auto load_pt = [&](int pt, double (&X)[3], double& act) {
    X[0] = a.points[3 * (long long)pt]; X[1] = a.points[3 * (long long)pt + 1]; X[2] = a.points[3 * (long long)pt + 2];
    act = a.pt_active ? (a.pt_active[pt] ? 1.0 : 0.0) : 1.0;
  };
  // look-ahead: pixel / point index two batches ahead, the point one batch ahead
  int o = ch.begin + threadIdx.x;
  const int o_wave = ch.begin + (threadIdx.x & ~63);
  if (o_wave > last) return;
  double2 m, m_n;
  int pt, pt_n;
  double X[3], act;
  load_obs(o, m, pt);
  load_pt(pt, X, act);
  load_obs(o + 256, m_n, pt_n);
  double v[32];
  {
    double Xn[3], act_n;
    load_pt(pt_n, Xn, act_n);
    double2 m_nn; int pt_nn;
    load_obs(o + 512, m_nn, pt_nn);
    rows(o, m, X, act, v);
    to_lds(v);
    X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2]; act = act_n; m = m_n; m_n = m_nn; pt_n = pt_nn;
  }
  for (int ow = o_wave + 256; ow <= last; ow += 256) {
    o += 256;
    double Xn[3], act_n;
    load_pt(pt_n, Xn, act_n);
    double2 m_nn; int pt_nn;
    load_obs(o + 512, m_nn, pt_nn);
    rows(o, m, X, act, v);   // batch j + 1: vector arithmetic ...
    gram();                  // ... under the matrix instructions of batch j
    // (the scheduler would put all the arithmetic first: ask for operand read / 2 matrix instructions / 16 vector
    // instructions in turn - ~250 vector instructions and 32 matrix instructions per batch)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
    }
    to_lds(v);
    X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2]; act = act_n; m = m_n; m_n = m_nn; pt_n = pt_nn;
  }
  gram();
}
template <int K>
__device__ __forceinline__ void camera_sweep_gram_body(const CamSweepArgs& a, int chunk, double* s_rows /* LDS, kCsLdsDoubles, 16-byte aligned */) {
  const SweepChunk ch = a.chunks[chunk];
  const int cam = a.img_cam[ch.image];
  const int model = a.cam_model[cam];
  double rec[9], kin[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) rec[k] = a.camrec[9 * ch.image + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) kin[k] = a.intr[9 * cam + k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* mine = s_rows + (size_t)wv * 64 * kCsPitch;
  double* wr = mine + lane * kCsPitch;
  // operand of matrix instruction q: row (lane >> 4) of the pair (2q, 2q + 1), element lane & 15
  const double* rd = mine + (lane >> 5) * kCsPitch + ((lane >> 4) & 1) * 16 + (lane & 15);
  cs_d4 accs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accs[i] = (cs_d4){0.0, 0.0, 0.0, 0.0};
  if (ch.end > ch.begin) {
    if (model == MAVBA_M_PINHOLE) camera_gram_wave<K, MAVBA_M_PINHOLE>(a, ch, rec, kin, wr, rd, accs);
    else if (model == MAVBA_M_OPENCV) camera_gram_wave<K, MAVBA_M_OPENCV>(a, ch, rec, kin, wr, rd, accs);
    else camera_gram_wave<K, MAVBA_M_CATA>(a, ch, rec, kin, wr, rd, accs);
  }
  // G of the four waves -> the fixed 135-slot layout of the chunk partial (upper triangle of G, element (R, C))
  __syncthreads();
  double* red = s_rows;  // [4][256]
  const cs_d4 acc = (accs[0] + accs[1]) + (accs[2] + accs[3]);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) red[wv * 256 + (4 * rr + (lane >> 4)) * 16 + (lane & 15)] = acc[rr];  // D layout: row 4 r + (lane >> 4), column lane & 15
  __syncthreads();
  const int R = threadIdx.x >> 4, C = threadIdx.x & 15;
  if (R <= C) {
    const double g = (red[threadIdx.x] + red[256 + threadIdx.x]) + (red[512 + threadIdx.x] + red[768 + threadIdx.x]);
    int slot = -1;
    if (R < 6) slot = C < 6 ? sym_idx(R, C, 6) : C < 15 ? 27 + R * 9 + (C - 6) : 21 + R;
    else if (R < 15) slot = C < 15 ? 81 + sym_idx(R - 6, C - 6, 9) : 126 + (R - 6);
    if (slot >= 0) a.partial[(size_t)chunk * kSweepAcc + slot] = g;
  }
}

}  // namespace mavba
#endif  // MAVBA_SWEEP_BODY_H_
