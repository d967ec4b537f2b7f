// kernels.hip — hand-written gfx950 kernels of the bundle-adjustment hot path.
//
// Data layout in HBM (all FP64):
//   observations  point-major (all observations of a 3-D point contiguous), uv as double2
//   Jacobian      structure-of-arrays planes of length Nstride: R[2], Jp[6], Jc[12], Jk[2*KMAX]
//                 -> every wave store is 64 lanes x 16 B contiguous
//   per point     planes of length NPs: Cu[6] (J_p^T J_p), gu[3] (J_p^T r), Gi[6], h[3]
//   Schur entries array-of-records (24 / 36 doubles) because the pair kernel gathers them
//   S             dense (n_pad + 64) x n_pad row-major; row n_pad is the right-hand side
//
// Every reduction is a fixed-shape tree (per-thread -> wave shuffle -> LDS -> per-block
// partial -> single-block final pass), so results are bit-reproducible run to run.
#include "internal.h"
#include <cstdlib>
#include <type_traits>
#include "ba_math.h"
#include "dev_reduce.h"
#include "lm_decide.h"
#include "lm_bodies.h"
#include "sweep_body.h"

namespace mavba {

// ---------------------------------------------------------------------------
// K0: camera records (hoists sin/cos out of the per-observation work)
// ---------------------------------------------------------------------------
__global__ void k_cam_prepare(int NI, const double* __restrict__ poses, double* __restrict__ camrec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NI) return;
  double rec[9];
  cam_prepare(poses + 6 * i, rec);
#pragma unroll
  for (int k = 0; k < 9; ++k) camrec[9 * i + k] = rec[k];
}
void launch_cam_prepare(hipStream_t st, int NI, const double* poses, double* camrec) {
  if (NI <= 0) return;
  hipLaunchKernelGGL(k_cam_prepare, dim3((NI + 127) / 128), dim3(128), 0, st, NI, poses, camrec);
}

// ---------------------------------------------------------------------------
// K1: the Jacobian sweep. Persistent 256-thread blocks; the whole camera table
// (72 B / image + intrinsics) is staged in LDS once per block; each lane owns
// TWO consecutive observations so that every plane store is 16 B per lane.
// Algorithmic bytes per observation: 48 read + 16 + 2*(9+K)*8 written.
// ---------------------------------------------------------------------------
constexpr int kSweepObsPerBlock = 512;
constexpr int kSweepMaxGrid = 1024;
int jacobian_sweep_grid(int N) {
  int g = (N + kSweepObsPerBlock - 1) / kSweepObsPerBlock;
  if (g < 1) g = 1;
  return g > kSweepMaxGrid ? kSweepMaxGrid : g;
}

struct CamTables {
  const double* rec; const double* intr; const int* img_cam; const int* model;
};

template <bool LDS_CAM>
__device__ __forceinline__ CamTables stage_cameras(double* smem, int NI, int NC,
                                                   const double* camrec, const double* intr,
                                                   const int* img_cam, const int* cam_model) {
  CamTables t;
  if constexpr (LDS_CAM) {
    double* s_rec = smem + 4;
    double* s_intr = s_rec + 9 * NI;
    int* s_cam = reinterpret_cast<int*>(s_intr + 9 * NC);
    int* s_model = s_cam + NI;
    for (int i = threadIdx.x; i < 9 * NI; i += blockDim.x) s_rec[i] = camrec[i];
    for (int i = threadIdx.x; i < 9 * NC; i += blockDim.x) s_intr[i] = intr[i];
    for (int i = threadIdx.x; i < NI; i += blockDim.x) s_cam[i] = img_cam[i];
    for (int i = threadIdx.x; i < NC; i += blockDim.x) s_model[i] = cam_model[i];
    __syncthreads();
    t.rec = s_rec; t.intr = s_intr; t.img_cam = s_cam; t.model = s_model;
  } else {
    t.rec = camrec; t.intr = intr; t.img_cam = img_cam; t.model = cam_model;
  }
  return t;
}
static size_t camera_lds_bytes(int NI, int NC) {
  size_t b = (size_t)(4 + 9 * NI + 9 * NC) * 8 + (size_t)(NI + NC) * 4;
  return (b + 15) & ~(size_t)15;
}
// Camera records staged in LDS only below this size. Since the points are renumbered by image list, neighbouring
// observations see the same few images and the records come from L1/L2 just as fast - without the table's LDS
// limiting occupancy (measured: C3 71 -> 73 % of HBM peak, C5 55 -> 74 %, candidate cost at C5 2.4x faster).
// MAVBA_CAM_LDS_KB re-enables the table (up to 150 of the 160 KiB per CU).
static size_t cam_lds_limit() {
  static const size_t v = [] { const char* e = std::getenv("MAVBA_CAM_LDS_KB"); return (size_t)(e ? std::atoi(e) : 0) * 1024; }();
  return v;
}

// MASK: the session has filtered points (pt_active != null); a separate instantiation so that the common path keeps
// its register budget (the extra byte load costs the unmasked kernel a wave of occupancy: 0.106 -> 0.127 ms at C3).
template <int KMAX, bool LDS_CAM, bool MASK>
__global__ void __launch_bounds__(256) k_jacobian_sweep(SweepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const CamTables T = stage_cameras<LDS_CAM>(smem, a.NI, a.NC, a.camrec, a.intr, a.img_cam, a.cam_model);
  constexpr int NOUT = 2 + 6 + 12 + 2 * KMAX;
  const int tid = threadIdx.x;
  const long long N = a.N, S = a.Nstride;
  double cost = 0.0;
  for (long long base = (long long)blockIdx.x * kSweepObsPerBlock; base < N;
       base += (long long)gridDim.x * kSweepObsPerBlock) {
    const long long o0 = base + 2 * tid;
    if (o0 >= N) continue;
    const bool two = (o0 + 1 < N);
    int im[2], pt[2];
    double uo[2], vo[2];
    if (two) {
      const int2 i2 = *reinterpret_cast<const int2*>(a.obs_img + o0);
      const int2 p2 = *reinterpret_cast<const int2*>(a.obs_pt + o0);
      const double2 m0 = a.uv[o0], m1 = a.uv[o0 + 1];
      im[0] = i2.x; im[1] = i2.y; pt[0] = p2.x; pt[1] = p2.y;
      uo[0] = m0.x; vo[0] = m0.y; uo[1] = m1.x; vo[1] = m1.y;
    } else {
      im[0] = im[1] = a.obs_img[o0]; pt[0] = pt[1] = a.obs_pt[o0];
      const double2 m0 = a.uv[o0];
      uo[0] = uo[1] = m0.x; vo[0] = vo[1] = m0.y;
    }
    double out[2][NOUT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cam = T.img_cam[im[j]];
      const int model = T.model[cam];
      double rec[9], kin[9], X[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) rec[k] = T.rec[9 * im[j] + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) kin[k] = T.intr[9 * cam + k];
      X[0] = a.points[3 * (long long)pt[j]]; X[1] = a.points[3 * (long long)pt[j] + 1];
      X[2] = a.points[3 * (long long)pt[j] + 2];
      double r[2], Jc[12], Jp[6], Jk[18];
      obs_jacobian(model, rec, kin, X, uo[j], vo[j], r, Jc, Jp, Jk);
      double w, half_rho;
      cauchy_weight(r[0] * r[0] + r[1] * r[1], a.loss_b, a.loss_inv_b, w, half_rho);
      if constexpr (MASK) { if (!a.pt_active[pt[j]]) { w = 0.0; half_rho = 0.0; } }  // filtered point: no residual block
      if (j == 0 || two) cost += half_rho;
      out[j][0] = w * r[0]; out[j][1] = w * r[1];
#pragma unroll
      for (int e = 0; e < 6; ++e) out[j][2 + e] = w * Jp[e];
#pragma unroll
      for (int e = 0; e < 12; ++e) out[j][8 + e] = w * Jc[e];
#pragma unroll
      for (int row = 0; row < 2; ++row)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) out[j][20 + row * KMAX + k] = w * Jk[row * 9 + k];
    }
    // plane e of the concatenated [R | Jp | Jc | Jk] output
    auto plane = [&](int e) -> double* {
      if (e < 2) return a.R + (long long)e * S;
      if (e < 8) return a.Jp + (long long)(e - 2) * S;
      if (e < 20) return a.Jc + (long long)(e - 8) * S;
      return a.Jk + (long long)(e - 20) * S;
    };
    if (two) {
#pragma unroll
      for (int e = 0; e < NOUT; ++e)
        *reinterpret_cast<double2*>(plane(e) + o0) = make_double2(out[0][e], out[1][e]);
    } else {
#pragma unroll
      for (int e = 0; e < NOUT; ++e) plane(e)[o0] = out[0][e];
    }
  }
  const double tot = block_sum_256(cost, smem);
  if (tid == 0) a.cost_partial[blockIdx.x] = tot;
}

template <bool LDS_CAM, bool MASK>
__global__ void __launch_bounds__(256) k_cost_only(SweepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const CamTables T = stage_cameras<LDS_CAM>(smem, a.NI, a.NC, a.camrec, a.intr, a.img_cam, a.cam_model);
  const long long N = a.N;
  double cost = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < N; o += (long long)gridDim.x * 256) {
    const int im = a.obs_img[o], pt = a.obs_pt[o];
    const double2 m = a.uv[o];
    const int cam = T.img_cam[im];
    double rec[9], kin[9], X[3], r[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) rec[k] = T.rec[9 * im + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) kin[k] = T.intr[9 * cam + k];
    X[0] = a.points[3 * (long long)pt]; X[1] = a.points[3 * (long long)pt + 1]; X[2] = a.points[3 * (long long)pt + 2];
    obs_residual(T.model[cam], rec, kin, X, m.x, m.y, r);
    double w, half_rho;
    cauchy_weight(r[0] * r[0] + r[1] * r[1], a.loss_b, a.loss_inv_b, w, half_rho);
    if constexpr (MASK) { if (!a.pt_active[pt]) half_rho = 0.0; }
    cost += half_rho;
  }
  const double tot = block_sum_256(cost, smem);
  if (threadIdx.x == 0) a.cost_partial[blockIdx.x] = tot;
}

// |r_raw| per observation (trivial loss) for the point-error report.
__global__ void __launch_bounds__(256) k_raw_residual_norm(SweepArgs a, double* __restrict__ out) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= a.N) return;
  const int im = a.obs_img[o], pt = a.obs_pt[o];
  const double2 m = a.uv[o];
  const int cam = a.img_cam[im];
  double r[2];
  obs_residual(a.cam_model[cam], a.camrec + 9 * im, a.intr + 9 * cam, a.points + 3 * (long long)pt, m.x, m.y, r);
  out[o] = (a.pt_active && !a.pt_active[pt]) ? 0.0 : sqrt(r[0] * r[0] + r[1] * r[1]);
}

void launch_jacobian_sweep(hipStream_t st, const SweepArgs& a) {
  if (a.N <= 0) return;
  const int grid = jacobian_sweep_grid(a.N);
  const size_t lds = camera_lds_bytes(a.NI, a.NC);
  const bool use_lds = lds <= cam_lds_limit();
  const size_t shm = use_lds ? lds : 64;
#define MAVBA_SWEEP(K)                                                                                           \
  if (a.pt_active) {                                                                                             \
    if (use_lds) hipLaunchKernelGGL((k_jacobian_sweep<K, true, true>), dim3(grid), dim3(256), shm, st, a);       \
    else hipLaunchKernelGGL((k_jacobian_sweep<K, false, true>), dim3(grid), dim3(256), shm, st, a);              \
  } else if (use_lds) hipLaunchKernelGGL((k_jacobian_sweep<K, true, false>), dim3(grid), dim3(256), shm, st, a); \
  else hipLaunchKernelGGL((k_jacobian_sweep<K, false, false>), dim3(grid), dim3(256), shm, st, a);
  if (a.KMAX <= 4) { MAVBA_SWEEP(4) } else if (a.KMAX <= 8) { MAVBA_SWEEP(8) } else { MAVBA_SWEEP(9) }
#undef MAVBA_SWEEP
}
void launch_cost_only(hipStream_t st, const SweepArgs& a) {
  if (a.N <= 0) return;
  const int grid = jacobian_sweep_grid(a.N);  // same partial count as the sweep
  const size_t lds = camera_lds_bytes(a.NI, a.NC);
  const bool use_lds = lds <= cam_lds_limit();
  if (a.pt_active) {
    if (use_lds) hipLaunchKernelGGL((k_cost_only<true, true>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_cost_only<false, true>), dim3(grid), dim3(256), 64, st, a);
  } else if (use_lds) hipLaunchKernelGGL((k_cost_only<true, false>), dim3(grid), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((k_cost_only<false, false>), dim3(grid), dim3(256), 64, st, a);
}
void launch_raw_residual_norm(hipStream_t st, const SweepArgs& a, double* out_norm) {
  if (a.N <= 0) return;
  hipLaunchKernelGGL(k_raw_residual_norm, dim3((a.N + 255) / 256), dim3(256), 0, st, a, out_norm);
}

// ---------------------------------------------------------------------------
// K3: per-point reductions, once per Jacobian evaluation. 16 lanes per 3-D point (4 points per
// wave): the lanes stride over the point's contiguous observations (coalesced 128-byte runs per
// plane) and combine with DPP row rotations — "wavefront reductions for the 3x3 point blocks".
//   Cu = sum Jp^T Jp (sym 6),  gu = sum Jp^T r,
//   Wk[q] = sum_{obs of camera q_cam[q]} Jk^T Jp  (KMAX x 3, unscaled) for each free camera
//           the point is seen by (the radius-independent part of the intrinsics Schur entries).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double row16_sum(double v) {
  // all-reduce inside each 16-lane DPP row: rotate right by 8, 4, 2, 1
#define MAVBA_ROR_ADD(N)                                                                                   \
  {                                                                                                        \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 | (N), 0xf, 0xf, false);        \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 | (N), 0xf, 0xf, false);        \
    v += __hiloint2double(hi, lo);                                                                         \
  }
  MAVBA_ROR_ADD(8) MAVBA_ROR_ADD(4) MAVBA_ROR_ADD(2) MAVBA_ROR_ADD(1)
#undef MAVBA_ROR_ADD
  return v;
}

template <int KMAX>
__global__ void __launch_bounds__(256) k_point_reduce(
    int NP, int NPs, int Nstride, const int* __restrict__ pt_start, const int* __restrict__ q_start,
    const int* __restrict__ q_cam, const int* __restrict__ obs_img, const int* __restrict__ img_cam,
    const double* __restrict__ R, const double* __restrict__ Jp, const double* __restrict__ Jk,
    double* __restrict__ Cu, double* __restrict__ gu, double* __restrict__ Wk) {
  const int g = threadIdx.x & 15;
  const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (p >= NP) return;  // whole 16-lane row leaves together
  const long long S = Nstride;
  const int b = pt_start[p], e = pt_start[p + 1];
  const int q0 = q_start[p], nq = q_start[p + 1] - q0;
  const int cam0 = nq > 0 ? q_cam[q0] : -1, cam1 = nq > 1 ? q_cam[q0 + 1] : -1;
  // One pass in chunks of 16 observations (one per lane). The chunk's products are reduced over the 16 lanes
  // right away and only the REDUCED values are accumulated (lane k & 15 owns output k): no per-lane accumulators
  // for the 2 x 3 KMAX intrinsics products, i.e. ~100 fewer registers and twice the waves in flight. For tracks
  // of at most 16 observations (one chunk) the sums are bit-identical to accumulating per lane first.
  // (Skipping the second camera's reductions for one-camera points was measured slower: 0.188 vs 0.177 ms at C3.)
  double c[6] = {0, 0, 0, 0, 0, 0}, gg[3] = {0, 0, 0};
  constexpr int NW = KMAX * 3, NOWN = (NW + 15) / 16;
  double own0[NOWN], own1[NOWN];
#pragma unroll
  for (int k = 0; k < NOWN; ++k) { own0[k] = 0.0; own1[k] = 0.0; }
  for (int ob = b; ob < e; ob += 16) {
    const int o = ob + g;
    const bool on = o < e;
    double jp[6] = {0, 0, 0, 0, 0, 0}, r0 = 0.0, r1 = 0.0, s0 = 0.0, s1 = 0.0;
    double jk[2 * KMAX];
#pragma unroll
    for (int k = 0; k < 2 * KMAX; ++k) jk[k] = 0.0;
    if (on) {
#pragma unroll
      for (int t = 0; t < 6; ++t) jp[t] = Jp[t * S + o];
      r0 = R[o]; r1 = R[S + o];
      if (nq > 0) {
        // the Jk loads go out together with the image -> camera lookup (they do not wait for it)
#pragma unroll
        for (int k = 0; k < 2 * KMAX; ++k) jk[k] = Jk[k * S + o];
        const int cam = img_cam[obs_img[o]];
        s0 = cam == cam0 ? 1.0 : 0.0; s1 = cam == cam1 ? 1.0 : 0.0;
      }
    }
    c[0] += jp[0] * jp[0] + jp[3] * jp[3]; c[1] += jp[0] * jp[1] + jp[3] * jp[4]; c[2] += jp[0] * jp[2] + jp[3] * jp[5];
    c[3] += jp[1] * jp[1] + jp[4] * jp[4]; c[4] += jp[1] * jp[2] + jp[4] * jp[5]; c[5] += jp[2] * jp[2] + jp[5] * jp[5];
    gg[0] += jp[0] * r0 + jp[3] * r1; gg[1] += jp[1] * r0 + jp[4] * r1; gg[2] += jp[2] * r0 + jp[5] * r1;
    if (nq > 0) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const double j0 = jk[k], j1 = jk[KMAX + k];
        const double w[3] = {j0 * jp[0] + j1 * jp[3], j0 * jp[1] + j1 * jp[4], j0 * jp[2] + j1 * jp[5]};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int kk = 3 * k + t;
          const double t0 = row16_sum(0.0 + s0 * w[t]), t1 = row16_sum(0.0 + s1 * w[t]);
          if (g == (kk & 15)) { own0[kk >> 4] += t0; own1[kk >> 4] += t1; }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = row16_sum(c[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) gg[k] = row16_sum(gg[k]);
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) Cu[k * NPs + p] = c[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) gu[k * NPs + p] = gg[k];
  }
  auto emit_own = [&](int q, const double* own) {
    double* out = Wk + (size_t)q * 27;
#pragma unroll
    for (int kk = 0; kk < NW; ++kk)
      if (g == (kk & 15)) out[kk] = own[kk >> 4];
    if (KMAX < 9 && g < 27 - KMAX * 3) out[KMAX * 3 + g] = 0.0;
    if (KMAX < 9 && g + 16 < 27 - KMAX * 3) out[KMAX * 3 + g + 16] = 0.0;
  };
  if (nq > 0) emit_own(q0, own0);
  if (nq > 1) emit_own(q0 + 1, own1);
  double W0[KMAX * 3];
  auto emit = [&](int q, double* W) {
    double* out = Wk + (size_t)q * 27;
#pragma unroll
    for (int k = 0; k < KMAX * 3; ++k) {
      const double v = row16_sum(W[k]);
      if (g == (k & 15)) out[k] = v;
    }
    if (KMAX < 9 && g < 27 - KMAX * 3) out[KMAX * 3 + g] = 0.0;
    if (KMAX < 9 && g + 16 < 27 - KMAX * 3) out[KMAX * 3 + g + 16] = 0.0;
  };
  // points seen by more than two free cameras: one extra pass per further camera
  for (int q = q0 + 2; q < q0 + nq; ++q) {
    const int cam = q_cam[q];
#pragma unroll
    for (int k = 0; k < KMAX * 3; ++k) W0[k] = 0.0;
    for (int o = b + g; o < e; o += 16) {
      if (img_cam[obs_img[o]] != cam) continue;
      double jp[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) jp[t] = Jp[t * S + o];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const double j0 = Jk[k * S + o], j1 = Jk[(KMAX + k) * S + o];
        W0[3 * k] += j0 * jp[0] + j1 * jp[3];
        W0[3 * k + 1] += j0 * jp[1] + j1 * jp[4];
        W0[3 * k + 2] += j0 * jp[2] + j1 * jp[5];
      }
    }
    emit(q, W0);
  }
}
void launch_point_reduce(hipStream_t st, int NP, int NPs, int Nstride, int KMAX, const int* pt_start,
                         const int* q_start, const int* q_cam, const int* obs_img, const int* img_cam,
                         const double* R, const double* Jp, const double* Jk, double* Cu, double* gu,
                         double* Wk) {
  if (NP <= 0) return;
  const dim3 g((NP + 15) / 16), b(256);
#define MAVBA_PR(K) hipLaunchKernelGGL((k_point_reduce<K>), g, b, 0, st, NP, NPs, Nstride, pt_start, q_start, q_cam, \
                                       obs_img, img_cam, R, Jp, Jk, Cu, gu, Wk)
  if (KMAX <= 4) MAVBA_PR(4); else if (KMAX <= 8) MAVBA_PR(8); else MAVBA_PR(9);
#undef MAVBA_PR
}

// ---------------------------------------------------------------------------
// K2: the J-free, point-major Schur front end (layout in internal.h). Per tile of consecutive points:
//   phase 1  one observation per lane: residual + Jacobian in registers (never stored), the 9 + 3 K products of the
//            observation parked in LDS (two rounds share the buffer), then one lane per (point, value) / ((point, camera),
//            value) adds the point's observations in order - a fixed sequential sum, independent of the tiling;
//   owner    one lane per point: Cu, gu out; damped block C = S Cu S + D^2, Gi = chol(C)^-1, h = Gi S gu
//            (the Schur eliminator's e-block step, D^2 = clamp(diag) / radius);
//   phase 2  intrinsics entry records from the Wk sums (one lane per (record, parameter)), pose entry records from the
//            Jacobians still in registers (staged through LDS, coalesced stores).
// HBM traffic: 48 B read per observation; 192 B / observation + 288 B / (point, camera) + 144 B / point written.
// The Jacobian planes (272 / 336 B per observation written, 400 - 500 B re-read by three kernels) are gone.
// ---------------------------------------------------------------------------
namespace {
constexpr int kFrPitch = kFrontObs + 1;  // park row pitch: rows v, v + 1 of one observation are one bank pair apart
template <int KMAX>
struct FrontShape {
  static constexpr int K3 = 3 * KMAX, NROWS = 9 + K3;
  static constexpr int ROUNDS = KMAX > 0 ? 2 : 1;
  static constexpr int NR = (NROWS + ROUNDS - 1) / ROUNDS;  // park rows per round; >= 9: the point rows are all in round 0
  static constexpr int PARK = NR * kFrPitch, SQ = kFrontQ * K3;
  static constexpr int REC = 128 * 25;  // half a tile of pose records (pitch 25) is staged at a time
  static constexpr int A = (PARK + SQ) > REC ? (PARK + SQ) : REC;
  static constexpr int DOUBLES = A + kFrontPts * 9 + kFrontPts * 12 + 4;
  static constexpr int INTS = (kFrontPts + 1) + kFrontObs + 2 * kFrontQ + 3;
  static constexpr size_t BYTES = (size_t)DOUBLES * 8 + (size_t)INTS * 4;
  static_assert(NR >= 9, "point rows must fit the first round");
};
}  // namespace

template <int KMAX, bool ENTRIES, bool MASK, bool TRACE = false>
__global__ void __launch_bounds__(256, 2) k_point_front(FrontArgs a) {  // two work-groups per CU (LDS and registers)
  using SH = FrontShape<KMAX>;
  if (!lm_spec_go(a.spec, &a.radius)) return;  // (speculative evaluation: only behind an accepted step, with the radius it leaves)
  long long stamp[TRACE ? 12 : 1];
  int nstamp = 0;
  auto mark = [&]() { if constexpr (TRACE) { if (nstamp < 12) stamp[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); } };
  extern __shared__ __attribute__((aligned(16))) double fr_smem[];
  double* s_park = fr_smem;                         // [NR][kFrPitch] the window's products of this round
  double* s_q = fr_smem + SH::PARK;                 // [kFrontQ][K3]  Wk sums
  double* s_rec = fr_smem;                          // pose records being staged (aliases park | s_q once they are consumed)
  double* s_sum = fr_smem + SH::A;                  // [kFrontPts][9] Cu(6) gu(3) sums
  double* s_g = s_sum + kFrontPts * 9;              // [kFrontPts][12] Gi(6) h(3) scale(3)
  double* s_red = s_g + kFrontPts * 12;             // [4]
  int* s_pb = reinterpret_cast<int*>(s_red + 4);    // [kFrontPts + 1] first observation of the tile's points
  int* s_cam = s_pb + kFrontPts + 1;                // [kFrontObs] camera of the window's observations
  int* s_qcam = s_cam + kFrontObs;                  // [kFrontQ] camera of the tile's intrinsics entries
  int* s_qpt = s_qcam + kFrontQ;                    // [kFrontQ] their point (index within the tile)
  const int tid = threadIdx.x;
  const SweepArgs& w = a.sw;
  const int NPs = a.NPs;
  const int t_begin = (int)((long long)a.num_tiles * blockIdx.x / gridDim.x);
  const int t_end = (int)((long long)a.num_tiles * (blockIdx.x + 1) / gridDim.x);
  double cost = 0.0;
  // A work-group walks a contiguous run of tiles. What a tile needs FIRST - its record, every lane's observation (image,
  // point, pixel) and the owner lanes' per-point inputs - is requested while the PREVIOUS tile is still in its second
  // phase, so the only latency left at the top of a tile is that of the gathers (camera record, point) behind them.
  FrontTile Tn{0, 0, 0, 0, 0, 0};
  int im_n = 0, pt_n = 0;
  double2 m_n = make_double2(0.0, 0.0);
  bool act_n = false, own_free_n = false;
  double own_sp_n[3] = {0.0, 0.0, 0.0};
  auto request_tile = [&](int t) {
    Tn = a.tiles[t];
    act_n = Tn.o0 + tid < Tn.o1;
    if (act_n) { im_n = w.obs_img[Tn.o0 + tid]; pt_n = w.obs_pt[Tn.o0 + tid]; m_n = w.uv[Tn.o0 + tid]; }
    if constexpr (ENTRIES) {
      own_free_n = false;
      if (tid < Tn.p1 - Tn.p0) {
        own_free_n = a.pt_free[Tn.p0 + tid] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) own_sp_n[k] = a.scale_pt[(size_t)k * NPs + Tn.p0 + tid];
      }
    }
  };
  if (t_begin < t_end) request_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    nstamp = 0;
    mark();  // 0: top of the tile (the trace keeps the work-group's LAST tile: steady state, prefetched)
    const FrontTile T = Tn;
    const int np = T.p1 - T.p0;
    const int o0 = T.o0, o1 = T.o1;
    const int q0 = T.q0, nq = KMAX > 0 ? T.q1 - T.q0 : 0;
    int im = im_n, lp = 0, pt0 = pt_n;
    double2 m0 = m_n;
    bool act = act_n;
    const bool own_free = own_free_n;
    const double own_sp[3] = {own_sp_n[0], own_sp_n[1], own_sp_n[2]};
    for (int j = tid; j <= np; j += 256) s_pb[j] = a.pt_start[T.p0 + j];
    for (int i = tid; i < np * 9; i += 256) s_sum[i] = 0.0;
    if constexpr (KMAX > 0) {
      for (int q = tid; q < nq; q += 256) { s_qcam[q] = a.q_cam[q0 + q]; s_qpt[q] = a.q_pt[q0 + q] - T.p0; }
      for (int i = tid; i < nq * SH::K3; i += 256) s_q[i] = 0.0;
    }
    const bool single = o1 - o0 <= kFrontObs;  // (all but tiles made of one very long track)
    // the observation's weighted Jacobian blocks stay in registers from phase 1 to phase 2
    double jc[12], jp[6];
    // residual + Jacobian of an observation (image imv, point pt, pixel m), rows weighted by sqrt(rho'); returns rho / 2
    auto eval_obs = [&](int imv, int pt, double2 m, double (&rr)[2], double (&jk)[18], int& cam) -> double {
      im = imv;
      lp = pt - T.p0;
      cam = w.img_cam[imv];
      const int model = w.cam_model[cam];
      double rec[9], kin[9], X[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) rec[k] = w.camrec[9 * imv + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) kin[k] = w.intr[9 * cam + k];
      X[0] = w.points[3 * (long long)pt]; X[1] = w.points[3 * (long long)pt + 1]; X[2] = w.points[3 * (long long)pt + 2];
      double r[2], Jc[12], Jp[6], Jk[18];
      obs_jacobian(model, rec, kin, X, m.x, m.y, r, Jc, Jp, Jk);
      double wgt, half_rho;
      cauchy_weight(r[0] * r[0] + r[1] * r[1], w.loss_b, w.loss_inv_b, wgt, half_rho);
      if constexpr (MASK) { if (!w.pt_active[pt]) { wgt = 0.0; half_rho = 0.0; } }  // filtered point: no residual block
      rr[0] = wgt * r[0]; rr[1] = wgt * r[1];
#pragma unroll
      for (int e = 0; e < 12; ++e) jc[e] = wgt * Jc[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) jp[e] = wgt * Jp[e];
#pragma unroll
      for (int e = 0; e < 18; ++e) jk[e] = wgt * Jk[e];
      return half_rho;
    };
    mark();  // 1: first loads out, tile bookkeeping written
    // ---- phase 1: per-point sums ----
    for (int base = o0; base < o1; base += kFrontObs) {
      const int o = base + tid;
      if (base != o0) {
        act = o < o1;
        if (act) { im = w.obs_img[o]; pt0 = w.obs_pt[o]; m0 = w.uv[o]; }
      }
      double prod[SH::NROWS];
      int cam = -1;
      if (act) {
        double rr[2], jk[18];
        cost += eval_obs(im, pt0, m0, rr, jk, cam);
        prod[0] = jp[0] * jp[0] + jp[3] * jp[3]; prod[1] = jp[0] * jp[1] + jp[3] * jp[4]; prod[2] = jp[0] * jp[2] + jp[3] * jp[5];
        prod[3] = jp[1] * jp[1] + jp[4] * jp[4]; prod[4] = jp[1] * jp[2] + jp[4] * jp[5]; prod[5] = jp[2] * jp[2] + jp[5] * jp[5];
        prod[6] = jp[0] * rr[0] + jp[3] * rr[1]; prod[7] = jp[1] * rr[0] + jp[4] * rr[1]; prod[8] = jp[2] * rr[0] + jp[5] * rr[1];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
          for (int t = 0; t < 3; ++t) prod[9 + 3 * k + t] = jk[k] * jp[t] + jk[9 + k] * jp[3 + t];
      }
      mark();  // 2: Jacobian + products in registers
      auto round = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        constexpr int lo = R * SH::NR, hi = (lo + SH::NR < SH::NROWS) ? lo + SH::NR : SH::NROWS;
        constexpr int PR = R == 0 ? 9 : 0;                             // point rows parked in this round
        constexpr int wlo = (lo > 9 ? lo : 9) - 9, WR = hi - 9 - wlo;  // Wk values [wlo, wlo + WR) parked in this round
        if (R > 0) lds_barrier();  // (the previous round's sums have been read)
        if (act) {
          if (R == 0) s_cam[tid] = cam;
#pragma unroll
          for (int v = lo; v < hi; ++v) s_park[(v - lo) * kFrPitch + tid] = prod[v];
        }
        lds_barrier();
        // One lane per sum, a point's observations added in order (a fixed sequential sum, independent of the tiling).
        const int nit = np * PR + nq * WR;
        for (int it = tid; it < nit; it += 256) {
          const bool is_pt = it < np * PR;
          int j, slot, c = -1;
          const double* row;
          double* dst;
          if (is_pt) {
            j = it / 9;
            slot = it - 9 * j;
            row = s_park + slot * kFrPitch - base;
            dst = s_sum + it;
          } else {
            const int t2 = it - np * PR;
            const int wr = WR > 0 ? WR : 1;
            const int q = t2 / wr, vv = wlo + (t2 - q * wr);
            j = s_qpt[q]; c = s_qcam[q];
            row = s_park + (vv + 9 - lo) * kFrPitch - base;
            dst = s_q + q * SH::K3 + vv;
          }
          const int b = max(s_pb[j], base), e = min(s_pb[j + 1], base + kFrontObs);
          const int* camv = s_cam - base;
          double acc = *dst;
          for (int i0 = b; i0 < e; i0 += 4) {  // four (clamped, masked) reads in flight, the adds stay in observation order
            const int i1 = min(i0 + 1, e - 1), i2 = min(i0 + 2, e - 1), i3 = min(i0 + 3, e - 1);
            double x0 = row[i0], x1 = row[i1], x2 = row[i2], x3 = row[i3];
            if (WR > 0 && !is_pt) {
              x0 = camv[i0] == c ? x0 : 0.0; x1 = camv[i1] == c ? x1 : 0.0; x2 = camv[i2] == c ? x2 : 0.0; x3 = camv[i3] == c ? x3 : 0.0;
            }
            acc += x0;
            acc += i0 + 1 < e ? x1 : 0.0;   // (adding 0.0 leaves the sum as it is)
            acc += i0 + 2 < e ? x2 : 0.0;
            acc += i0 + 3 < e ? x3 : 0.0;
          }
          *dst = acc;
        }
      };
      round(std::integral_constant<int, 0>{});
      mark();  // 3: round 0
      if constexpr (SH::ROUNDS > 1) round(std::integral_constant<int, 1>{});
      mark();  // 4: round 1
      lds_barrier();  // sums complete; the park buffer is free for the next window
    }
    if (tile + 1 < t_end) request_tile(tile + 1);  // (arrives under the rest of this tile)
    // ---- owner lanes: the point's sums out, its damped block factorised ----
    if (tid < np) {
      const int p = T.p0 + tid;
      double C6[6], g3[3];
#pragma unroll
      for (int k = 0; k < 6; ++k) C6[k] = s_sum[tid * 9 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g3[k] = s_sum[tid * 9 + 6 + k];
#pragma unroll
      for (int k = 0; k < 6; ++k) a.Cu[(size_t)k * NPs + p] = C6[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) a.gu[(size_t)k * NPs + p] = g3[k];
      if constexpr (ENTRIES) {
        double G[6] = {0, 0, 0, 0, 0, 0}, hh[3] = {0, 0, 0}, sp[3] = {0, 0, 0};
        if (own_free) {
#pragma unroll
          for (int k = 0; k < 3; ++k) sp[k] = own_sp[k];
          double C[6];
          C[0] = sp[0] * sp[0] * C6[0]; C[1] = sp[0] * sp[1] * C6[1]; C[2] = sp[0] * sp[2] * C6[2];
          C[3] = sp[1] * sp[1] * C6[3]; C[4] = sp[1] * sp[2] * C6[4]; C[5] = sp[2] * sp[2] * C6[5];
          const double inv_radius = 1.0 / a.radius;
          C[0] = __builtin_fma(clampd(C[0], a.dmin, a.dmax), inv_radius, C[0]);
          C[3] = __builtin_fma(clampd(C[3], a.dmin, a.dmax), inv_radius, C[3]);
          C[5] = __builtin_fma(clampd(C[5], a.dmin, a.dmax), inv_radius, C[5]);
          bool fin = chol3_inv_fast(C, G);
          const double gs[3] = {sp[0] * g3[0], sp[1] * g3[1], sp[2] * g3[2]};
          gi_mul(G, gs, hh);
#pragma unroll
          for (int k = 0; k < 6; ++k) fin = fin && isfinite(G[k]);
          if (!fin) atomicAdd(a.fail, 1.0);
        }
        if (s_pb[tid + 1] > s_pb[tid]) {  // (points without observations keep the zeros of the set-up: nothing reads them)
#pragma unroll
          for (int k = 0; k < 6; ++k) a.Gi[(size_t)k * NPs + p] = G[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) a.h[(size_t)k * NPs + p] = hh[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) s_g[tid * 12 + k] = G[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_g[tid * 12 + 6 + k] = hh[k]; s_g[tid * 12 + 9 + k] = sp[k]; }
      }
    }
    if constexpr (ENTRIES) {
      lds_barrier();
      mark();  // 5: owner lanes done
      // ---- intrinsics entry records: Uk = (s_k Wk s_p) Gi^T (9 x 3), ek = Uk h; one lane per (record, parameter) ----
      if constexpr (KMAX > 0) {
        for (int it = tid; it < nq * 9; it += 256) {
          const int q = it / 9, k = it - 9 * q;
          const double* g = s_g + s_qpt[q] * 12;
          double w0 = 0.0, w1 = 0.0, w2 = 0.0;
          if (k < KMAX) {
            const double sk = a.scale_cam[6 * w.NI + 9 * s_qcam[q] + k];
            const double* W = s_q + q * SH::K3 + 3 * k;
            w0 = W[0] * sk * g[9]; w1 = W[1] * sk * g[10]; w2 = W[2] * sk * g[11];
          }
          const double u0 = w0 * g[0];
          const double u1 = w0 * g[1] + w1 * g[2];
          const double u2 = w0 * g[3] + w1 * g[4] + w2 * g[5];
          double* out = a.Eintr + (size_t)(q0 + q) * kIntrRec;
          out[3 * k] = u0; out[3 * k + 1] = u1; out[3 * k + 2] = u2;
          out[27 + k] = u0 * g[6] + u1 * g[7] + u2 * g[8];
        }
        lds_barrier();  // the Wk sums are consumed: their LDS becomes the record staging buffer
      }
      mark();  // 6: intrinsics records
      // ---- pose entry records: U_a = (Jc' ^T Jp') Gi^T (6 x 3), e_a = U_a h ----
      for (int base = o0; base < o1; base += kFrontObs) {
        const int o = base + tid;
        if (!single) {
          act = o < o1;
          if (act) { double rr[2], jk[18]; int cam; (void)eval_obs(w.obs_img[o], w.obs_pt[o], w.uv[o], rr, jk, cam); }
        }
        double rec[kPoseRec];
#pragma unroll
        for (int k = 0; k < kPoseRec; ++k) rec[k] = 0.0;
        if (act && a.pt_free[T.p0 + lp]) {
          const double* g = s_g + lp * 12;
          double jps[6];
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int k = 0; k < 3; ++k) jps[row * 3 + k] = jp[row * 3 + k] * g[9 + k];
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            const double sc = a.scale_cam[6 * im + e];
            const double j0 = jc[e] * sc, j1 = jc[6 + e] * sc;
            const double w0 = j0 * jps[0] + j1 * jps[3], w1 = j0 * jps[1] + j1 * jps[4], w2 = j0 * jps[2] + j1 * jps[5];
            const double u0 = w0 * g[0];
            const double u1 = w0 * g[1] + w1 * g[2];
            const double u2 = w0 * g[3] + w1 * g[4] + w2 * g[5];
            rec[3 * e] = u0; rec[3 * e + 1] = u1; rec[3 * e + 2] = u2;
            rec[18 + e] = u0 * g[6] + u1 * g[7] + u2 * g[8];
          }
        }
        mark();  // 7: pose records in registers
        const long long lim = (long long)o1 * kPoseRec;
        // staged through LDS (pitch 25), coalesced stores: the whole window at once where the buffer holds 256 records,
        // else in two halves
        constexpr int kStage = SH::A >= 256 * 25 ? 256 : 128;
#pragma unroll
        for (int part = 0; part < 256 / kStage; ++part) {
          if (tid / kStage == part) {
#pragma unroll
            for (int k = 0; k < kPoseRec; ++k) s_rec[(tid % kStage) * 25 + k] = rec[k];
          }
          lds_barrier();
          const long long gbase = ((long long)base + part * kStage) * kPoseRec;
#pragma unroll 4
          for (int i = tid; i < kStage * kPoseRec; i += 256) {
            const int t = i / kPoseRec, k = i - t * kPoseRec;
            if (gbase + i < lim) a.Epose[gbase + i] = s_rec[t * 25 + k];
          }
          lds_barrier();
        }
      }
    } else {
      lds_barrier();  // (the next tile re-initialises the sums)
    }
    mark();  // 8: records stored
  }
  if constexpr (TRACE) {
    if (a.trace && (tid & 63) == 0 && blockIdx.x < 16384) {  // one line per wave: [n, stamps...] (single-tile work-groups)
      long long* out = a.trace + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 16;
      out[0] = nstamp;
      for (int i = 0; i < nstamp; ++i) out[1 + i] = stamp[i];
    }
  }
  const double tot = block_sum_256(cost, s_red);
  if (tid == 0) w.cost_partial[blockIdx.x] = tot;
}

// Grid: one tile per work-group (dynamic balance beat a resident grid walking runs of tiles with prefetch: 0.275 vs 0.291 ms
// at C3); beyond kFrontMaxGrid tiles a work-group walks a contiguous run. MAVBA_FRONT_GRID overrides (tuning).
int point_front_grid(int num_tiles) {
  static const int cap = [] { const char* e = std::getenv("MAVBA_FRONT_GRID"); const int v = e ? std::atoi(e) : 0; return v > 0 && v < kFrontMaxGrid ? v : kFrontMaxGrid; }();
  return num_tiles < 1 ? 0 : (num_tiles > cap ? cap : num_tiles);
}
void launch_point_front(hipStream_t st, const FrontArgs& a, int kmax_intr, bool entries) {
  const int grid = point_front_grid(a.num_tiles);
  if (grid <= 0) return;
  const bool mask = a.sw.pt_active != nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  // MAVBA_FRONT_TRACE=<file>: the 5th entries launch (widest model, unmasked) records s_memtime stamps per wave
  static const char* trace_file = std::getenv("MAVBA_FRONT_TRACE");
  static int trace_calls = 0;
  if (trace_file && entries && !mask && kmax_intr > 4 && kmax_intr <= 8 && ++trace_calls == 5) {
    const size_t n = (size_t)16384 * 4 * 16;
    long long* tr = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&tr), n * 8);
    (void)hipMemsetAsync(tr, 0, n * 8, st);
    FrontArgs b = a;
    b.trace = tr;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_point_front<8, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)FrontShape<8>::BYTES);
    hipLaunchKernelGGL((k_point_front<8, true, false, true>), dim3(grid), dim3(256), FrontShape<8>::BYTES, st, b);
    std::vector<long long> h(n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), tr, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(tr);
    if (FILE* fp = std::fopen(trace_file, "w")) {
      for (int b2 = 0; b2 < std::min(grid, 16384); ++b2)
        for (int wv = 0; wv < 4; ++wv) {
          const long long* r = &h[((size_t)b2 * 4 + wv) * 16];
          std::fprintf(fp, "%d %d", b2, wv);
          for (int i = 0; i < (int)r[0]; ++i) std::fprintf(fp, " %lld", r[1 + i]);
          std::fprintf(fp, "\n");
        }
      std::fclose(fp);
    }
    return;
  }
#define MAVBA_FRONT_LAUNCH(K, E, M)                                                                                     \
  {                                                                                                                     \
    static bool configured[64] = {};  /* per device: more than 64 KiB of dynamic LDS has to be asked for */            \
    if (dev >= 0 && dev < 64 && !configured[dev]) {                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_point_front<K, E, M>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)FrontShape<K>::BYTES);                                                             \
      configured[dev] = true;                                                                                           \
    }                                                                                                                   \
    hipLaunchKernelGGL((k_point_front<K, E, M>), dim3(grid), dim3(256), FrontShape<K>::BYTES, st, a);                   \
  }
#define MAVBA_FRONT_K(K)                                                                       \
  {                                                                                            \
    if (entries) { if (mask) MAVBA_FRONT_LAUNCH(K, true, true) else MAVBA_FRONT_LAUNCH(K, true, false) }   \
    else { if (mask) MAVBA_FRONT_LAUNCH(K, false, true) else MAVBA_FRONT_LAUNCH(K, false, false) }         \
  }
  if (kmax_intr <= 0) MAVBA_FRONT_K(0)
  else if (kmax_intr <= 4) MAVBA_FRONT_K(4)
  else if (kmax_intr <= 8) MAVBA_FRONT_K(8)
  else MAVBA_FRONT_K(9)
#undef MAVBA_FRONT_K
#undef MAVBA_FRONT_LAUNCH
}

// ---------------------------------------------------------------------------
// K4: camera sweep. Image-major pass that RECOMPUTES each observation's camera-
// side Jacobian (cheaper than gathering 2x(6+K) doubles through a permutation:
// 28 B of input per observation) and reduces, per image chunk,
//   PP = Jc^T Jc, Pg = Jc^T r, PI = Jc^T Jk, II = Jk^T Jk, Ig = Jk^T r.
// One block per chunk; the image's camera record is block-uniform.
// ---------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) k_camera_sweep(CamSweepArgs a) {
  if (!lm_spec_go(a.spec, nullptr)) return;  // (speculative evaluation: only behind an accepted step)
  __shared__ double s_red[4 * kSweepAcc];
  camera_sweep_body<K>(a, blockIdx.x, s_red);  // (sweep_body.h)
}

// K4 with free intrinsics: the same sums as ONE 16 x 16 Gram matrix per chunk on the matrix cores. Every observation
// gives two weighted rows  v = w [ Jc(6) | Jk(9) | r ]  (16 entries); G = sum v^T v holds PP, PI, II in its pose /
// intrinsics blocks and Pg, Ig in its last column. The vector kernel above keeps ~120 FP64 accumulators per lane (one
// wave per SIMD, nothing hides the Jacobian arithmetic, and 135 wave reductions at the end); here the accumulators are
// 4 registers, already summed over the wave. A lane's rows go through LDS (pitch 34 doubles: 16-byte writes, and
// the operand reads of a matrix instruction - element l & 15 of row l >> 4 of two neighbouring observations - are two
// contiguous 128-byte runs), v_mfma_f64_16x16x4_f64 takes 4 rows (2 observations) per issue; both operands are the
// same register. Two work-groups per CU: one wave's Jacobian arithmetic runs under another's matrix instructions.
template <int K>
__global__ void __launch_bounds__(256, 2) k_camera_sweep_gram(CamSweepArgs a) {
  if (!lm_spec_go(a.spec, nullptr)) return;  // (speculative evaluation: only behind an accepted step)
  __shared__ __attribute__((aligned(16))) double s_rows[kCsLdsDoubles];
  camera_sweep_gram_body<K>(a, blockIdx.x, s_rows);  // (sweep_body.h)
}
void launch_camera_sweep(hipStream_t st, const CamSweepArgs& a, int kmax, bool any_intr_free) {
  if (a.num_chunks <= 0) return;
  const dim3 g(a.num_chunks), b(256);
  // (with free intrinsics the vector form - k_camera_sweep<4 / 8 / 9>, ~120 accumulators per lane - was 0.133 against the Gram
  // form's 0.074 ms at C3, round 2; its switch MAVBA_CAMSWEEP_VECTOR went in round 6, the template keeps the code)
  if (!any_intr_free) hipLaunchKernelGGL((k_camera_sweep<0>), g, b, 0, st, a);  // 27 sums: the vector kernel is the cheaper one
  else if (kmax <= 4) hipLaunchKernelGGL((k_camera_sweep_gram<4>), g, b, 0, st, a);
  else if (kmax <= 8) hipLaunchKernelGGL((k_camera_sweep_gram<8>), g, b, 0, st, a);
  else hipLaunchKernelGGL((k_camera_sweep_gram<9>), g, b, 0, st, a);
}

// Rotation priors: one lane per prior (reference bundle_adjustment.cc:72-111, :428-444).
__global__ void k_rot_prior(int n, const int* __restrict__ prior_img, const double* __restrict__ R0,
                            double w, const double* __restrict__ poses, double* __restrict__ res,
                            double* __restrict__ jac, double* __restrict__ cost_partial, LmSpec spec) {
  if (!lm_spec_go(spec, nullptr)) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  double r, j[3];
  rot_prior_eval(poses + 6 * prior_img[q], R0 + 9 * q, w, r, j);
  res[q] = r; jac[3 * q] = j[0]; jac[3 * q + 1] = j[1]; jac[3 * q + 2] = j[2];
  cost_partial[q] = 0.5 * r * r;
}
void launch_rot_prior(hipStream_t st, int n, const int* prior_img, const double* prior_R0, double w,
                      const double* poses, double* res, double* jac, double* cost_partial, const LmSpec& spec) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rot_prior, dim3((n + 127) / 128), dim3(128), 0, st, n, prior_img, prior_R0, w, poses, res, jac, cost_partial, spec);
}

// Per image: sum its chunks (fixed order) + its rotation priors -> img_rec[81] and
// the image's intrinsics part -> img_intr_tmp[54]. One 64-lane group per image (camera_reduce_img_body, lm_bodies.h).
__device__ void partial_reduce_task(const PartialReduce T, int lane, double* part_pp, double* part_ip, double* part_ii);  // (below, with k_partial_reduce)
__global__ void __launch_bounds__(64) k_camera_reduce_img(
    int NI, const int* __restrict__ img_chunk_start, const double* __restrict__ partial,
    const int* __restrict__ prior_start, const double* __restrict__ prior_res,
    const double* __restrict__ prior_jac, double* __restrict__ img_rec, double* __restrict__ img_intr_tmp, LmSpec spec, PartialRide ride) {
  if (!lm_spec_go(spec, nullptr)) return;
  if ((int)blockIdx.x >= NI) { partial_reduce_task(ride.tasks[blockIdx.x - NI], threadIdx.x, ride.pp, ride.ip, ride.ii); return; }
  camera_reduce_img_body(blockIdx.x, threadIdx.x, img_chunk_start, partial, prior_start, prior_res, prior_jac, img_rec, img_intr_tmp);
}
// Per camera: sum the intrinsics parts of its images (fixed order: 16 interleaved partial sums
// per element, then a fixed tree) — one 1024-thread block per camera.
__global__ void __launch_bounds__(1024) k_camera_reduce_cam(
    const int* __restrict__ cam_img_start, const int* __restrict__ cam_imgs,
    const double* __restrict__ img_intr_tmp, double* __restrict__ cam_rec, LmSpec spec) {
  __shared__ double s_part[16][64];
  if (!lm_spec_go(spec, nullptr)) return;
  const int c = blockIdx.x;
  const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
  double s = 0.0;
  if (e < kCamRec) {
    // (four images per trip, their loads in flight together: the dependent index -> value round trips were this kernel's
    // whole time, 10 us for 250 images per camera; the order of the additions is unchanged)
    const int t1 = cam_img_start[c + 1];
    for (int t = cam_img_start[c] + part; t < t1; t += 64) {
      const int i0 = cam_imgs[t], i1 = cam_imgs[min(t + 16, t1 - 1)], i2 = cam_imgs[min(t + 32, t1 - 1)], i3 = cam_imgs[min(t + 48, t1 - 1)];
      const double x0 = img_intr_tmp[(size_t)i0 * kCamRec + e], x1 = img_intr_tmp[(size_t)i1 * kCamRec + e];
      const double x2 = img_intr_tmp[(size_t)i2 * kCamRec + e], x3 = img_intr_tmp[(size_t)i3 * kCamRec + e];
      s += x0;
      if (t + 16 < t1) s += x1;
      if (t + 32 < t1) s += x2;
      if (t + 48 < t1) s += x3;
    }
  }
  s_part[part][e] = s;
  __syncthreads();
  if (part == 0 && e < kCamRec) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += s_part[k][e];
    cam_rec[(size_t)c * kCamRec + e] = tot;
  }
}
void launch_camera_reduce(hipStream_t st, int NI, int NC, const int* img_chunk_start,
                          const double* partial, const int* prior_start, const double* prior_res,
                          const double* prior_jac, const int* cam_img_start, const int* cam_imgs,
                          double* img_rec, double* cam_rec, double* img_intr_tmp, bool with_cams, const LmSpec& spec, const PartialRide& ride) {
  if (NI > 0 || ride.n > 0)
    hipLaunchKernelGGL(k_camera_reduce_img, dim3(NI + ride.n), dim3(64), 0, st, NI, img_chunk_start, partial,
                       prior_start, prior_res, prior_jac, img_rec, img_intr_tmp, spec, ride);
  if (NC > 0 && with_cams)  // (no free intrinsics: nobody reads the per-camera sums, they stay zero)
    hipLaunchKernelGGL(k_camera_reduce_cam, dim3(NC), dim3(1024), 0, st, cam_img_start, cam_imgs, img_intr_tmp, cam_rec, spec);
}

// ---------------------------------------------------------------------------
// Jacobi scaling (ceres EstimateScale): s_j = 1 / (1 + |J_:j|), 0 for constant columns.
// scale_cam is indexed like S: 6*i+e for poses, 6*NI + 9*c + k for intrinsics.
// ---------------------------------------------------------------------------
__global__ void k_scales(int NI, int NC, int NP, int NPs, int jacobi,
                         const unsigned char* __restrict__ pose_free,
                         const unsigned char* __restrict__ intr_free,
                         const unsigned char* __restrict__ pt_free, const double* __restrict__ img_rec,
                         const double* __restrict__ cam_rec, const double* __restrict__ Cu,
                         double* __restrict__ scale_cam, double* __restrict__ scale_pt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ncam = 6 * NI + 9 * NC;
  if (t < ncam) {
    double n2; bool fr;
    if (t < 6 * NI) {
      const int i = t / 6, e = t % 6;
      n2 = img_rec[(size_t)i * kImgRec + sym_idx(e, e, 6)];
      fr = pose_free[t] != 0;
    } else {
      const int c = (t - 6 * NI) / 9, k = (t - 6 * NI) % 9;
      n2 = cam_rec[(size_t)c * kCamRec + sym_idx(k, k, 9)];
      fr = intr_free[9 * c + k] != 0;
    }
    scale_cam[t] = fr ? (jacobi ? 1.0 / (1.0 + sqrt(n2)) : 1.0) : 0.0;
  }
  if (t < NP) {
    const bool fr = pt_free[t] != 0;
    const int d[3] = {0, 3, 5};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      scale_pt[k * NPs + t] = fr ? (jacobi ? 1.0 / (1.0 + sqrt(Cu[d[k] * NPs + t])) : 1.0) : 0.0;
  }
}
void launch_scales(hipStream_t st, int NI, int NC, int NP, int NPs, int jacobi,
                   const unsigned char* pose_free, const unsigned char* intr_free,
                   const unsigned char* pt_free, const double* img_rec, const double* cam_rec,
                   const double* Cu, double* scale_cam, double* scale_pt) {
  const int n = max(6 * NI + 9 * NC, NP);
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scales, dim3((n + 255) / 256), dim3(256), 0, st, NI, NC, NP, NPs, jacobi, pose_free,
                     intr_free, pt_free, img_rec, cam_rec, Cu, scale_cam, scale_pt);
}

// max |g| and |x|^2 over the free parameters. partial[b][0] = max, [b][1] = sum.
// Blocks [0, gp) cover points; the blocks behind them the camera columns.
__global__ void __launch_bounds__(256) k_state_norms(
    int NI, int NC, int NP, int NPs, int gp, int cam_part, const unsigned char* __restrict__ pose_free,
    const unsigned char* __restrict__ intr_free, const unsigned char* __restrict__ pt_free,
    const double* __restrict__ poses, const double* __restrict__ intr, const double* __restrict__ points,
    const double* __restrict__ img_rec, const double* __restrict__ cam_rec, const double* __restrict__ gu,
    double* __restrict__ partial, LmSpec spec) {
  __shared__ double s_red[4];
  if (!lm_spec_go(spec, nullptr)) return;
  // (camera blocks behind the point blocks: one block for 3 000 columns was this kernel's long pole, 11 -> 5 us at C3)
  state_norms_body(blockIdx.x, gp, gridDim.x - gp, NI, NC, NP, NPs, cam_part, pose_free, intr_free, pt_free, poses, intr, points,
                   img_rec, cam_rec, gu, partial, s_red);
}
void state_norms_grid(int NI, int NC, int NP, int* gp, int* gc) {
  *gp = std::min(512, (NP + 255) / 256);
  *gc = std::max(1, std::min(kStateNormsCamBlocks, (6 * NI + 9 * NC + 255) / 256));
}
void launch_state_norms(hipStream_t st, int NI, int NC, int NP, int NPs, bool cam_part,
                        const unsigned char* pose_free, const unsigned char* intr_free,
                        const unsigned char* pt_free, const double* poses, const double* intr,
                        const double* points, const double* img_rec, const double* cam_rec,
                        const double* gu, double* partial, int* grid_out, const LmSpec& spec) {
  int gp, gc;
  state_norms_grid(NI, NC, NP, &gp, &gc);
  hipLaunchKernelGGL(k_state_norms, dim3(gp + gc), dim3(256), 0, st, NI, NC, NP, NPs, gp, cam_part ? 1 : 0,
                     pose_free, intr_free, pt_free, poses, intr, points, img_rec, cam_rec, gu, partial, spec);
  *grid_out = gp + gc;
}


// ---------------------------------------------------------------------------
// Pose entries: U_a = (Jc' ^T Jp') Gi^T (6x3), e_a = U_a h. One lane per
// observation; records are transposed through LDS so the 192-byte records leave
// the block as one contiguous, fully coalesced 48 KB store.
// ---------------------------------------------------------------------------
// FACTOR: the point's damped 3x3 block - C_p = S_p Cu S_p + D_p^2, C_p = G G^T, Gi = G^-1, h = Gi (S_p gu); the Schur
// eliminator's e-block step, D^2 = clamp(diag(Js^T Js)) / radius - is factorised HERE (by every observation of the point
// - 9 + 3 values in instead of 6 + 3, ~100 instructions in an HBM-bound kernel) and its first observation stores Gi and h
// for the later kernels: one launch less per linear solve.
template <bool FACTOR>
__global__ void __launch_bounds__(256) k_entries_pose(
    int N, int Nstride, int NPs, const int* __restrict__ obs_img, const int* __restrict__ obs_pt,
    const unsigned char* __restrict__ pt_free, const double* __restrict__ Jc, const double* __restrict__ Jp,
    const double* __restrict__ scale_cam, const double* __restrict__ scale_pt, double* __restrict__ Gi,
    double* __restrict__ h, double* __restrict__ Epose, const int* __restrict__ pt_start, const double* __restrict__ Cu,
    const double* __restrict__ gu, double radius, double dmin, double dmax, double* __restrict__ fail) {
  __shared__ double s_rec[256 * 25];
  const long long S = Nstride;
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  double rec[kPoseRec];
#pragma unroll
  for (int k = 0; k < kPoseRec; ++k) rec[k] = 0.0;
  if (o < N) {
    const int p = obs_pt[o];
    const bool fr = pt_free[p] != 0;
    if constexpr (FACTOR) {
      if (!fr && o == pt_start[p]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Gi[k * NPs + p] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) h[k * NPs + p] = 0.0;
      }
    }
    if (fr) {
      const int im = obs_img[o];
      double sp[3], G[6], hh[3], jp[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) sp[k] = scale_pt[k * NPs + p];
      if constexpr (FACTOR) {
        double C[6];
        C[0] = sp[0] * sp[0] * Cu[p];           C[1] = sp[0] * sp[1] * Cu[NPs + p];     C[2] = sp[0] * sp[2] * Cu[2 * NPs + p];
        C[3] = sp[1] * sp[1] * Cu[3 * NPs + p]; C[4] = sp[1] * sp[2] * Cu[4 * NPs + p]; C[5] = sp[2] * sp[2] * Cu[5 * NPs + p];
        C[0] += clampd(C[0], dmin, dmax) / radius;
        C[3] += clampd(C[3], dmin, dmax) / radius;
        C[5] += clampd(C[5], dmin, dmax) / radius;
        const bool ok = chol3_inv(C, G);
        const double gs[3] = {sp[0] * gu[p], sp[1] * gu[NPs + p], sp[2] * gu[2 * NPs + p]};
        gi_mul(G, gs, hh);
        if (o == pt_start[p]) {
          bool fin = ok;
#pragma unroll
          for (int k = 0; k < 6; ++k) fin = fin && isfinite(G[k]);
          if (!fin) atomicAdd(fail, 1.0);
#pragma unroll
          for (int k = 0; k < 6; ++k) Gi[k * NPs + p] = G[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) h[k * NPs + p] = hh[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) hh[k] = h[k * NPs + p];
#pragma unroll
        for (int k = 0; k < 6; ++k) G[k] = Gi[k * NPs + p];
      }
#pragma unroll
      for (int row = 0; row < 2; ++row)
#pragma unroll
        for (int k = 0; k < 3; ++k) jp[row * 3 + k] = Jp[(row * 3 + k) * S + o] * sp[k];
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const double sc = scale_cam[6 * im + e];
        const double j0 = Jc[e * S + o] * sc, j1 = Jc[(6 + e) * S + o] * sc;
        const double w0 = j0 * jp[0] + j1 * jp[3], w1 = j0 * jp[1] + j1 * jp[4], w2 = j0 * jp[2] + j1 * jp[5];
        // U = W Gi^T : U[k] = sum_{m<=k} W[m] Gi[k][m]
        const double u0 = w0 * G[0];
        const double u1 = w0 * G[1] + w1 * G[2];
        const double u2 = w0 * G[3] + w1 * G[4] + w2 * G[5];
        rec[3 * e] = u0; rec[3 * e + 1] = u1; rec[3 * e + 2] = u2;
        rec[18 + e] = u0 * hh[0] + u1 * hh[1] + u2 * hh[2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kPoseRec; ++k) s_rec[threadIdx.x * 25 + k] = rec[k];
  __syncthreads();
  const long long base = (long long)blockIdx.x * 256 * kPoseRec;
  const long long lim = (long long)N * kPoseRec;
  for (int i = threadIdx.x; i < 256 * kPoseRec; i += 256) {
    const int t = i / kPoseRec, k = i - t * kPoseRec;
    if (base + i < lim) Epose[base + i] = s_rec[t * 25 + k];
  }
}
void launch_entries_pose(hipStream_t st, int N, int Nstride, int NPs, const int* obs_img,
                         const int* obs_pt, const unsigned char* pt_free, const double* Jc,
                         const double* Jp, const double* scale_cam, const double* scale_pt,
                         double* Gi, double* h, double* Epose) {
  if (N <= 0) return;
  hipLaunchKernelGGL((k_entries_pose<false>), dim3((N + 255) / 256), dim3(256), 0, st, N, Nstride, NPs, obs_img, obs_pt,
                     pt_free, Jc, Jp, scale_cam, scale_pt, Gi, h, Epose, nullptr, nullptr, nullptr, 1.0, 0.0, 0.0, nullptr);
}
// k_point_factor + k_entries_pose in one launch (points without observations keep whatever Gi / h hold: nothing reads them)
void launch_factor_entries_pose(hipStream_t st, int N, int Nstride, int NPs, const int* obs_img, const int* obs_pt,
                                const unsigned char* pt_free, const double* Jc, const double* Jp, const double* scale_cam,
                                const double* scale_pt, double* Gi, double* h, double* Epose, const int* pt_start,
                                const double* Cu, const double* gu, double radius, double dmin, double dmax, double* fail) {
  if (N <= 0) return;
  hipLaunchKernelGGL((k_entries_pose<true>), dim3((N + 255) / 256), dim3(256), 0, st, N, Nstride, NPs, obs_img, obs_pt,
                     pt_free, Jc, Jp, scale_cam, scale_pt, Gi, h, Epose, pt_start, Cu, gu, radius, dmin, dmax, fail);
}

// Intrinsics entries, per linear solve: one lane per (free point p, free camera c seen by p):
//   Uk = (s_k * Wk * s_p) Gi^T  (9x3),  ek = Uk h,   Wk from k_point_reduce.
__global__ void __launch_bounds__(256) k_entries_intr(
    int Q, int NI, int NPs, const int* __restrict__ q_pt, const int* __restrict__ q_cam,
    const double* __restrict__ Wk, const double* __restrict__ scale_cam, const double* __restrict__ scale_pt,
    const double* __restrict__ Gi, const double* __restrict__ h, double* __restrict__ Eintr) {
  // One lane per (record q, intrinsics parameter k): neighbouring lanes read neighbouring 24-byte rows of Wk and write
  // neighbouring rows of the entry record - every access is a coalesced run without staging (one lane per RECORD touched
  // 64 cache lines per load, 0.097 ms at C3; through LDS 0.086 ms at two waves per SIMD). The point's 12 values are read
  // by its 18 lanes from the same cache lines.
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(t / 9), k = (int)(t - 9ll * q);
  if (q >= Q) return;
  const int p = q_pt[q], c = q_cam[q];
  const double* W = Wk + (size_t)q * 27 + 3 * k;
  const double sk = scale_cam[6 * NI + 9 * c + k];
  const double w0 = W[0] * sk * scale_pt[p], w1 = W[1] * sk * scale_pt[NPs + p], w2 = W[2] * sk * scale_pt[2 * NPs + p];
  const double u0 = w0 * Gi[p];
  const double u1 = w0 * Gi[NPs + p] + w1 * Gi[2 * NPs + p];
  const double u2 = w0 * Gi[3 * (size_t)NPs + p] + w1 * Gi[4 * (size_t)NPs + p] + w2 * Gi[5 * (size_t)NPs + p];
  double* out = Eintr + (size_t)q * kIntrRec;
  out[3 * k] = u0; out[3 * k + 1] = u1; out[3 * k + 2] = u2;
  out[27 + k] = u0 * h[p] + u1 * h[NPs + p] + u2 * h[2 * NPs + p];
}
void launch_entries_intr(hipStream_t st, int Q, int NI, int NPs, const int* q_pt, const int* q_cam,
                         const double* Wk, const double* scale_cam, const double* scale_pt,
                         const double* Gi, const double* h, double* Eintr) {
  if (Q <= 0) return;
  hipLaunchKernelGGL(k_entries_intr, dim3((unsigned)((9ll * Q + 255) / 256)), dim3(256), 0, st, Q, NI, NPs, q_pt, q_cam, Wk, scale_cam,
                     scale_pt, Gi, h, Eintr);
}

// ---------------------------------------------------------------------------
// Schur chunks: one wave per chunk of (x, y) entry pairs of one block of S;
// partial[chunk] = sum U_x U_y^T  (+ sum e_x over x == y terms for diagonal kinds).
// Every block of S is produced by exactly one finalize group from its chunk
// partials in order: no atomics, bit-reproducible.
// ---------------------------------------------------------------------------
template <int RX, int RY, int SX, int SY, bool DIAG>
__global__ void __launch_bounds__(256) k_schur_chunks(int num_chunks, const SchurChunk* __restrict__ chunks,
                                                      const int2* __restrict__ terms,
                                                      const double* __restrict__ EX,
                                                      const double* __restrict__ EY,
                                                      double* __restrict__ partial) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware order: work-group b runs on XCD b % 8 (observed dispatch, used for speed only), so
  // give every XCD one CONTIGUOUS run of chunks — neighbouring blocks share entry records and then
  // find them in the same L2.
  const int per_xcd = (gridDim.x + 7) >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int cid = wg * 4 + wv;
  if (cid >= num_chunks) return;
  const SchurChunk ch = chunks[cid];
  constexpr int NA = RX * RY;
  constexpr int PS = NA + (DIAG ? RX : 0);
  double acc[NA], ev[RX];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < RX; ++i) ev[i] = 0.0;
  for (int t = ch.begin + lane; t < ch.end; t += 64) {
    const int2 xy = terms[t];
    const double* px = EX + (size_t)xy.x * SX;
    const double* py = EY + (size_t)xy.y * SY;
    double ux[RX * 3], uy[RY * 3];
#pragma unroll
    for (int i = 0; i < RX * 3; i += 2) {  // RX*3 is even for RX = 6; odd tail handled below
      if (i + 1 < RX * 3) { const double2 v = *reinterpret_cast<const double2*>(px + i); ux[i] = v.x; ux[i + 1] = v.y; }
      else ux[i] = px[i];
    }
#pragma unroll
    for (int i = 0; i < RY * 3; i += 2) {
      if (i + 1 < RY * 3) { const double2 v = *reinterpret_cast<const double2*>(py + i); uy[i] = v.x; uy[i + 1] = v.y; }
      else uy[i] = py[i];
    }
#pragma unroll
    for (int r = 0; r < RX; ++r)
#pragma unroll
      for (int c = 0; c < RY; ++c)
        acc[r * RY + c] += ux[3 * r] * uy[3 * c] + ux[3 * r + 1] * uy[3 * c + 1] + ux[3 * r + 2] * uy[3 * c + 2];
    if (DIAG && xy.x == xy.y) {
#pragma unroll
      for (int r = 0; r < RX; ++r) ev[r] += px[RX * 3 + r];
    }
  }
  double* out = partial + (size_t)ch.slot * PS;
#pragma unroll
  for (int i = 0; i < NA; ++i) { const double v = wave_sum(acc[i]); if (lane == 0) out[i] = v; }
  if (DIAG) {
#pragma unroll
    for (int r = 0; r < RX; ++r) { const double v = wave_sum(ev[r]); if (lane == 0) out[NA + r] = v; }
  }
}

// ---------------------------------------------------------------------------
// Schur clusters (layout in internal.h): one work-group per cluster. The stacked entry matrix E of a batch
// of kClBatch points is built in LDS (rows x 96 columns, pitch 100 -> conflict-free operand reads), and
// S_cl += E E^T runs on v_mfma_f64_16x16x4_f64: the lower 16x16 tiles (36 for 128 rows, 21 for 96) are dealt
// round-robin to the 4 waves; both operands of a tile are rows of E, so a k-step costs NT LDS reads and <= 9
// matrix instructions per wave. 53 KB of LDS -> three work-groups per CU: one fetches its next batch while
// the others keep the matrix cores busy. Every entry record is read from HBM exactly once per linear solve.
// ---------------------------------------------------------------------------
namespace {
typedef double cl_d4 __attribute__((ext_vector_type(4)));
constexpr int kClK = 3 * kClBatch, kClPitch = kClK + 2;  // (2 * pitch) mod 64 dwords = 4: the 32 lanes of a half-wave's operand read hit 64 different banks
constexpr int kClThreads = 512, kClWaves = kClThreads / 64;  // two waves per SIMD: with one, every instruction of the clear / scatter / fetch phases shows its full latency (~10 cycles per instruction measured)
template <int I, int C>
struct ClShape {
  static constexpr int images = I, cams = C, cam_row0 = 6 * I, hrow = 6 * I + 9 * C, rows = (hrow + 16) / 16 * 16;
  static constexpr int NT = rows / 16, tiles = NT * (NT + 1) / 2, acc = (tiles + kClWaves - 1) / kClWaves;
  static constexpr int tab_ip = I * (I + 1) / 2, tab_ii = tab_ip + C * I, tab = tab_ii + C * (C + 1) / 2;
};
// false: small batches, two work-groups per CU that hide each other's latencies; true: one work-group per CU that
// prefetches its next batch and double-buffers the operand reads itself
constexpr bool kClPipelined = kClBatch >= 32;
// does wave W (tiles t = W, W + kClWaves, ... of the row-major lower triangle) read row block `row` as an operand?
template <int NT>
constexpr bool cluster_row_used(int W, int row) {
  int t = 0;
  for (int i = 0; i < NT; ++i)
    for (int j = 0; j <= i; ++j) {
      if (t % kClWaves == W && (i == row || j == row)) return true;
      ++t;
    }
  return false;
}
template <class SH, int W>
__device__ __forceinline__ void cluster_mfma(const double* __restrict__ E, int lane, cl_d4 (&acc)[SH::acc]) {
  constexpr int NT = SH::NT;
  const int li = lane & 15, lk = lane >> 4;
  const double* base = E + li * kClPitch + lk;
  // Two operand register sets, no copies: the LDS reads of k-step kk + 4 are issued before the matrix
  // instructions of k-step kk, so their latency hides behind those (copies between the sets made the
  // compiler wait for the reads at the top of every step).
  auto load = [&](double (&x)[NT], int kk) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (cluster_row_used<NT>(W, i)) x[i] = base[16 * i * kClPitch + kk];  // (only the row blocks of this wave's tiles)
  };
  auto mma = [&](const double (&x)[NT]) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (t % kClWaves == W) acc[t / kClWaves] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[j], acc[t / kClWaves], 0, 0, 0);
        ++t;
      }
  };
  if constexpr (!kClPipelined) {
    // two work-groups share the CU: the other one's matrix instructions cover this wave's LDS reads
    double a[NT];
#pragma unroll 1
    for (int kk = 0; kk < kClK; kk += 4) { load(a, kk); mma(a); }
  } else {
    static_assert(kClK % 8 == 0, "two k-steps per trip");
    double a[NT], b[NT];
    load(a, 0);
#pragma unroll 1
    for (int kk = 0; kk < kClK; kk += 8) {
      load(b, kk + 4);
      __builtin_amdgcn_sched_barrier(0);  // reads first, then the matrix instructions they hide behind
      mma(a);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 8 < kClK) load(a, kk + 8);
      __builtin_amdgcn_sched_barrier(0);
      mma(b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// element (R, C), R >= C, of the cluster's product -> partial slot
template <class SH>
__device__ __forceinline__ void cluster_store(int R, int C, double v, const int* __restrict__ tab,
                                              double* __restrict__ part_pp, double* __restrict__ part_ip,
                                              double* __restrict__ part_ii) {
  constexpr int P0 = SH::cam_row0, H = SH::hrow;
  if (R < P0) {                       // pose x pose
    const int la = R / 6, r = R - 6 * la, lb = C / 6, c = C - 6 * lb;
    if (la == lb && r < c) return;    // diagonal blocks: the finalize pass only reads r >= c
    const int slot = tab[la * (la + 1) / 2 + lb];
    if (slot >= 0) part_pp[(size_t)slot * 42 + r * 6 + c] = v;
  } else if (R < H) {
    const int lc = (R - P0) / 9, r = (R - P0) - 9 * lc;
    if (C < P0) {                     // intrinsics x pose
      const int la = C / 6, c = C - 6 * la;
      const int slot = tab[SH::tab_ip + lc * SH::images + la];
      if (slot >= 0) part_ip[(size_t)slot * 54 + r * 6 + c] = v;
    } else {                          // intrinsics x intrinsics
      const int lc2 = (C - P0) / 9, c = (C - P0) - 9 * lc2;
      if (lc == lc2 && r < c) return;
      const int slot = tab[SH::tab_ii + lc * (lc + 1) / 2 + lc2];
      if (slot >= 0) part_ii[(size_t)slot * 90 + r * 9 + c] = v;
    }
  } else if (R == H) {                // h row: the right-hand-side parts of the diagonal blocks
    if (C < P0) {
      const int la = C / 6, r = C - 6 * la;
      const int slot = tab[la * (la + 1) / 2 + la];
      if (slot >= 0) part_pp[(size_t)slot * 42 + 36 + r] = v;
    } else if (C < H) {
      const int lc = (C - P0) / 9, r = (C - P0) - 9 * lc;
      const int slot = tab[SH::tab_ii + lc * (lc + 1) / 2 + lc];
      if (slot >= 0) part_ii[(size_t)slot * 90 + 81 + r] = v;
    }
  }
}
template <class SH, int W>
__device__ __forceinline__ void cluster_emit(int lane, const cl_d4 (&acc)[SH::acc], const int* __restrict__ tab,
                                             double* __restrict__ part_pp, double* __restrict__ part_ip,
                                             double* __restrict__ part_ii) {
  const int li = lane & 15, lk = lane >> 4;
  int t = 0;
#pragma unroll
  for (int i = 0; i < SH::NT; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      if (t % kClWaves == W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int R = 16 * i + lk + 4 * r, C = 16 * j + li;  // D layout of the matrix instruction
          if (R >= C) cluster_store<SH>(R, C, acc[t / kClWaves][r], tab, part_pp, part_ip, part_ii);
        }
      }
      ++t;
    }
}
}  // namespace

namespace {
constexpr int kClChunk = (kClImagesMax * kClBatch * 2 / 3 * 9 + kClThreads - 1) / kClThreads, kClQChunk = (2 * kClBatch * 14 + kClThreads - 1) / kClThreads + 1;  // value loads a thread has in flight per batch (pose / intrinsics records)
// Records of one batch, HBM -> LDS matrix. Element e of a record sits at (row e / 3, column e % 3) relative to
// the record's base (row 6 la or 96 + 9 lc, column 3 * point-in-batch). Thread tid takes the double2 number
// f = u * 256 + tid of the batch's contiguous record range. The value loads do not wait for the record's local
// index (only the scatter does), so a batch costs ONE memory latency - and that one is spent while the matrix
// cores work on the previous batch.
template <int REC, int USED, int NU>
struct ClusterRegs { double2 v[NU]; unsigned meta[NU]; };  // meta = local index << 8 | point in batch (host-packed), 0xFFFF = not in the cluster (full registers: 16-bit fields make the compiler merge - and wait)
// What thread tid does with its u-th double2 of a batch never changes: record number within the batch, offset in the
// record range, offset in E relative to the record's base. Computed once per work-group (the divisions per element and
// per batch were ~800 instructions of the scatter and ~350 of the fetch, with the matrix cores idle). Only the USED
// double2s of a record are numbered (9 of 12 for a pose record, 14 of 18 for an intrinsics record).
template <int REC, int USED, int NU>
struct ClusterPlan {
  static constexpr int D2 = (USED + 1) / 2;
  int src[NU];               // double offset of the double2 from the batch's first record
  int rec_no[NU];            // record number within the batch
  unsigned dst[NU];          // offset of .x in E from the record's base (row, column); bit 15: .y starts the next row, bit 14: .y is used
  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int f = u * kClThreads + tid, oo = f / D2, e2 = (f - oo * D2) * 2;
      rec_no[u] = oo;
      src[u] = oo * REC + e2;
      dst[u] = (unsigned)((e2 / 3) * kClPitch + e2 % 3) | (e2 % 3 == 2 ? 0x8000u : 0u) | (e2 + 1 < USED ? 0x4000u : 0u);
    }
  }
};
// The loads are UNCONDITIONAL (a slot beyond the batch's records reads the last record again and is ignored by the
// scatter): one basic block, nothing depends on a loaded value, so all of them go out back to back - with a branch per
// load the compiler guarded re-used registers with s_waitcnt vmcnt(0) and the fetch phase waited out the HBM latency.
template <int REC, int USED, int NU>
__device__ __forceinline__ void cluster_fetch(ClusterRegs<REC, USED, NU>& R, const ClusterPlan<REC, USED, NU>& P, int first,
                                              int count, const unsigned short* __restrict__ rec_meta,
                                              const double* __restrict__ rec) {
  if (count <= 0) return;  // (wave-uniform)
  const double* base = rec + (size_t)first * REC;
  const unsigned short* mbase = rec_meta + first;
  const int last = count - 1;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int over = max(P.rec_no[u] - last, 0);
    R.v[u] = *reinterpret_cast<const double2*>(base + (P.src[u] - over * REC));
    R.meta[u] = mbase[P.rec_no[u] - over];
  }
}
template <int REC, int USED, int NU>
__device__ __forceinline__ void cluster_scatter(const ClusterRegs<REC, USED, NU>& R, const ClusterPlan<REC, USED, NU>& P,
                                                double* __restrict__ E, int row0, int row_step, int count) {
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (P.rec_no[u] < count && R.meta[u] != 0xFFFFu) {
      const int at = (row0 + row_step * (int)(R.meta[u] >> 8)) * kClPitch + 3 * (int)(R.meta[u] & 255u) + (int)(P.dst[u] & 0x3FFFu);
      E[at] = R.v[u].x;
      if (P.dst[u] & 0x4000u) E[at + ((P.dst[u] & 0x8000u) ? kClPitch - 2 : 1)] = R.v[u].y;
    }
}
// the generic forms (any u0) for the overflow path
template <int REC, int USED, int NU>
__device__ __forceinline__ void cluster_fetch_any(ClusterRegs<REC, USED, NU>& R, int tid, int u0, int first, int count,
                                                  const unsigned short* __restrict__ rec_meta, const double* __restrict__ rec) {
  constexpr int D2 = (USED + 1) / 2;
  const int ntot = count * D2;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int f = (u0 + u) * kClThreads + tid;
    R.meta[u] = 0xFFFFu;
    if (f < ntot) {
      const int oo = f / D2, e2 = (f - oo * D2) * 2, o = first + oo;
      R.v[u] = *reinterpret_cast<const double2*>(rec + (size_t)o * REC + e2);
      R.meta[u] = rec_meta[o];
    }
  }
}
template <int REC, int USED, int NU>
__device__ __forceinline__ void cluster_scatter_any(const ClusterRegs<REC, USED, NU>& R, double* __restrict__ E, int tid, int u0,
                                                    int row0, int row_step) {
  constexpr int D2 = (USED + 1) / 2;
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (R.meta[u] != 0xFFFFu) {
      const int f = (u0 + u) * kClThreads + tid, e2 = (f % D2) * 2;
      const int at = (row0 + row_step * (int)(R.meta[u] >> 8) + e2 / 3) * kClPitch + 3 * (int)(R.meta[u] & 255u) + e2 % 3;
      E[at] = R.v[u].x;
      if (e2 + 1 < USED) E[at + ((e2 % 3 == 2) ? kClPitch - 2 : 1)] = R.v[u].y;
    }
}
// what does not fit the prefetch registers (the range also holds the records of points that are not in the
// cluster - long tracks between clustered points - so its length is not bounded by the cluster's capacity)
template <int REC, int USED, int NU>
__device__ __forceinline__ void cluster_overflow(double* __restrict__ E, int tid, int first, int count, int row0,
                                                 int row_step, const unsigned short* __restrict__ rec_meta,
                                                 const double* __restrict__ rec) {
  const int u_end = (count * ((USED + 1) / 2) + kClThreads - 1) / kClThreads;
  for (int uc = NU; uc < u_end; uc += NU) {
    ClusterRegs<REC, USED, NU> R;
    cluster_fetch_any(R, tid, uc, first, count, rec_meta, rec);
    cluster_scatter_any(R, E, tid, uc, row0, row_step);
  }
}
}  // namespace

template <class SH, bool TRACE>
__global__ void __launch_bounds__(kClThreads, kClPipelined ? 1 : 2) k_schur_clusters(
    const SchurCluster* __restrict__ clusters, const int* __restrict__ tabs, const int* __restrict__ pt_start,
    const int* __restrict__ q_start, const unsigned short* __restrict__ obs_meta,
    const unsigned short* __restrict__ q_meta, const unsigned char* __restrict__ pt_clustered,
    const double* __restrict__ Epose, const double* __restrict__ Eintr, const double* __restrict__ h, int NPs,
    double* __restrict__ part_pp, double* __restrict__ part_ip, double* __restrict__ part_ii,
    long long* __restrict__ trace) {
  __shared__ __attribute__((aligned(16))) double E[SH::rows * kClPitch];
  // (TRACE: a separate instantiation - the 24 stamps are 48 registers the production kernel needs for itself)
  long long stamp[TRACE ? 24 : 1];
  int nstamp = 0;
  auto mark = [&]() { if constexpr (TRACE) { if (nstamp < 24) stamp[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); } };
  mark();
  __shared__ int s_bounds[2][kClMaxBatches + 1];  // first observation / intrinsics entry of every batch
  __shared__ int s_tab[SH::tab];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const SchurCluster cl = clusters[blockIdx.x];
  const int nbatch = (cl.p1 - cl.p0 + kClBatch - 1) / kClBatch;
  for (int i = tid; i <= nbatch; i += kClThreads) {
    const int p = min(cl.p0 + i * kClBatch, cl.p1);
    s_bounds[0][i] = pt_start[p];
    s_bounds[1][i] = q_start[p];
  }
  for (int i = tid; i < SH::tab; i += kClThreads) s_tab[i] = tabs[(size_t)blockIdx.x * SH::tab + i];
  cl_d4 acc[SH::acc];
#pragma unroll
  for (int i = 0; i < SH::acc; ++i) acc[i] = (cl_d4){0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  ClusterRegs<kPoseRec, 18, kClChunk> RP;
  ClusterRegs<kIntrRec, 27, kClQChunk> RQ;
  ClusterPlan<kPoseRec, 18, kClChunk> PP;
  ClusterPlan<kIntrRec, 27, kClQChunk> PQ;
  PP.init(tid);
  PQ.init(tid);
  double hv = 0.0;
  unsigned char hon = 0;
  auto fetch = [&](int bi) {
    const int b0 = cl.p0 + bi * kClBatch, b1 = min(b0 + kClBatch, cl.p1);
    cluster_fetch(RP, PP, s_bounds[0][bi], s_bounds[0][bi + 1] - s_bounds[0][bi], obs_meta, Epose);
    cluster_fetch(RQ, PQ, s_bounds[1][bi], s_bounds[1][bi + 1] - s_bounds[1][bi], q_meta, Eintr);
    hon = 0;
    if (tid < (b1 - b0) * 3) {
      const int pp = tid / 3, t = tid - 3 * pp;
      hon = pt_clustered[b0 + pp];
      hv = h[(size_t)t * NPs + b0 + pp];
    }
  };
  if constexpr (kClPipelined) fetch(0);
  for (int bi = 0; bi < nbatch; ++bi) {
    if constexpr (!kClPipelined) fetch(bi);  // all loads of the batch are in flight while E is cleared
    for (int i = tid; i < SH::rows * kClPitch / 2; i += kClThreads) reinterpret_cast<double2*>(E)[i] = make_double2(0.0, 0.0);
    __syncthreads();
    mark();
    cluster_scatter(RP, PP, E, 0, 6, s_bounds[0][bi + 1] - s_bounds[0][bi]);
    cluster_scatter(RQ, PQ, E, SH::cam_row0, 9, s_bounds[1][bi + 1] - s_bounds[1][bi]);
    if (hon) E[SH::hrow * kClPitch + tid] = hv;
    cluster_overflow<kPoseRec, 18, kClChunk>(E, tid, s_bounds[0][bi], s_bounds[0][bi + 1] - s_bounds[0][bi], 0, 6, obs_meta, Epose);
    cluster_overflow<kIntrRec, 27, kClQChunk>(E, tid, s_bounds[1][bi], s_bounds[1][bi + 1] - s_bounds[1][bi], SH::cam_row0, 9, q_meta, Eintr);
    __syncthreads();
    mark();
    if constexpr (kClPipelined) {
      if (bi + 1 < nbatch) fetch(bi + 1);  // travels while the matrix cores work on this batch
      __builtin_amdgcn_sched_barrier(0);   // (keep the loads here: the scheduler would sink them to their use)
    }
    mark();
    switch (wv) {
      case 0: cluster_mfma<SH, 0>(E, lane, acc); break;
      case 1: cluster_mfma<SH, 1>(E, lane, acc); break;
      case 2: cluster_mfma<SH, 2>(E, lane, acc); break;
      case 3: cluster_mfma<SH, 3>(E, lane, acc); break;
      case 4: cluster_mfma<SH, 4>(E, lane, acc); break;
      case 5: cluster_mfma<SH, 5>(E, lane, acc); break;
      case 6: cluster_mfma<SH, 6>(E, lane, acc); break;
      default: cluster_mfma<SH, 7>(E, lane, acc); break;
    }
    mark();
    __syncthreads();
    mark();
  }
  const int* tab = s_tab;
  switch (wv) {
    case 0: cluster_emit<SH, 0>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 1: cluster_emit<SH, 1>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 2: cluster_emit<SH, 2>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 3: cluster_emit<SH, 3>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 4: cluster_emit<SH, 4>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 5: cluster_emit<SH, 5>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    case 6: cluster_emit<SH, 6>(lane, acc, tab, part_pp, part_ip, part_ii); break;
    default: cluster_emit<SH, 7>(lane, acc, tab, part_pp, part_ip, part_ii); break;
  }
  mark();
  if constexpr (TRACE) {
    if (trace && (tid & 63) == 0 && wv < 4 && blockIdx.x < 4096) {  // one line per wave (the first four): [n, stamps...]
      long long* out = trace + ((size_t)blockIdx.x * 4 + wv) * 32;
      out[0] = nstamp;
      for (int i = 0; i < nstamp; ++i) out[1 + i] = stamp[i];
    }
  }
}
#define MAVBA_CL_S12 ClShape<12, 2>
#define MAVBA_CL_S16 ClShape<16, 3>
void launch_schur_clusters(hipStream_t st, ClusterShape shape, int num_clusters, const SchurCluster* clusters, const int* tab,
                           const int* pt_start, const int* q_start, const unsigned short* obs_meta,
                           const unsigned short* q_meta, const unsigned char* pt_clustered, const double* Epose,
                           const double* Eintr, const double* h, int NPs, double* part_pp, double* part_ip,
                           double* part_ii) {
  if (num_clusters <= 0) return;
  // MAVBA_CLUSTER_TRACE=<file>: the 5th launch of the process records s_memtime stamps per wave (debugging aid)
  long long* trace = nullptr;
  static int calls = 0;
  static const char* trace_file = std::getenv("MAVBA_CLUSTER_TRACE");
  const size_t trace_n = (size_t)4096 * 4 * 32;
  if (trace_file && ++calls == 5) { (void)hipMalloc(reinterpret_cast<void**>(&trace), trace_n * 8); (void)hipMemsetAsync(trace, 0, trace_n * 8, st); }
#define MAVBA_CL_LAUNCH(SHAPE, TR)                                                                                                    \
  hipLaunchKernelGGL((k_schur_clusters<SHAPE, TR>), dim3(num_clusters), dim3(kClThreads), 0, st, clusters, tab, pt_start, q_start, obs_meta, \
                     q_meta, pt_clustered, Epose, Eintr, h, NPs, part_pp, part_ip, part_ii, trace)
  if (shape.images == 12) { if (trace) MAVBA_CL_LAUNCH(MAVBA_CL_S12, true); else MAVBA_CL_LAUNCH(MAVBA_CL_S12, false); }
  else { if (trace) MAVBA_CL_LAUNCH(MAVBA_CL_S16, true); else MAVBA_CL_LAUNCH(MAVBA_CL_S16, false); }
#undef MAVBA_CL_LAUNCH
  if (trace) {
    std::vector<long long> hst(trace_n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hst.data(), trace, trace_n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(trace);
    if (FILE* f = std::fopen(trace_file, "w")) {
      for (int b = 0; b < std::min(num_clusters, 4096); ++b)
        for (int w = 0; w < 4; ++w) {
          const long long* r = &hst[((size_t)b * 4 + w) * 32];
          std::fprintf(f, "%d %d", b, w);
          for (int i = 0; i < (int)r[0]; ++i) std::fprintf(f, " %lld", r[1 + i]);
          std::fprintf(f, "\n");
        }
      std::fclose(f);
    }
  }
}

int schur_partial_stride(int kind) { return kind == BLK_PP ? 42 : kind == BLK_IP ? 54 : 90; }
void launch_schur_chunks(hipStream_t st, int kind, int num_chunks, const SchurChunk* chunks,
                         const int2* terms, const double* Epose, const double* Eintr, double* partial) {
  if (num_chunks <= 0) return;
  const dim3 g((((num_chunks + 3) / 4) + 7) / 8 * 8), b(256);  // multiple of 8: one contiguous run per XCD
  if (kind == BLK_PP)
    hipLaunchKernelGGL((k_schur_chunks<6, 6, kPoseRec, kPoseRec, true>), g, b, 0, st, num_chunks, chunks, terms, Epose, Epose, partial);
  else if (kind == BLK_IP)
    hipLaunchKernelGGL((k_schur_chunks<9, 6, kIntrRec, kPoseRec, false>), g, b, 0, st, num_chunks, chunks, terms, Eintr, Epose, partial);
  else
    hipLaunchKernelGGL((k_schur_chunks<9, 9, kIntrRec, kIntrRec, true>), g, b, 0, st, num_chunks, chunks, terms, Eintr, Eintr, partial);
}

// Blocks that many clusters touch (every cluster touches the intrinsics blocks) would make one finalize group add
// thousands of partials one after the other: runs of 32 are summed here first, one wave per run, same fixed order.
// Element idx of the records [begin, end) of `part` (PS doubles each) summed in a fixed order: sixteen running sums over the
// records begin + u, begin + 16 + u, ... and a fixed tree over them. The sixteen loads of a trip are independent and the last,
// partial trip is predicated instead of walked one record at a time: the pass is a chain of load latencies (~0.5 us each) -
// round 4: 8 sums + a sequential remainder took 25 such steps for the 157 partials of a local window's block, this takes 10.
__device__ __forceinline__ double strided_sum16(const double* __restrict__ part, int PS, int idx, int begin, int end) {
  double s[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) s[u] = 0.0;
  for (int c = begin; c < end; c += 16) {
    double x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = part[(size_t)min(c + u, end - 1) * PS + idx];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u] += (c + u < end) ? x[u] : 0.0;
  }
  return (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) + (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
}
// one task = one 64-lane group (also run as extra work-groups of k_camera_reduce_img / k_eval_head: PartialRide)
__device__ void partial_reduce_task(const PartialReduce T, int lane, double* part_pp, double* part_ip, double* part_ii) {
  double* part = T.kind == BLK_PP ? part_pp : T.kind == BLK_IP ? part_ip : part_ii;
  const int PS = T.kind == BLK_PP ? 42 : T.kind == BLK_IP ? 54 : 90;
  for (int idx = lane; idx < PS; idx += 64) part[(size_t)T.dst * PS + idx] = strided_sum16(part, PS, idx, T.src_begin, T.src_end);
}
__global__ void __launch_bounds__(256) k_partial_reduce(int num_tasks, const PartialReduce* __restrict__ tasks,
                                                        double* __restrict__ part_pp, double* __restrict__ part_ip,
                                                        double* __restrict__ part_ii) {
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (task >= num_tasks) return;
  partial_reduce_task(tasks[task], lane, part_pp, part_ip, part_ii);
}
void launch_partial_reduce(hipStream_t st, int num_tasks, const PartialReduce* tasks, double* part_pp, double* part_ip,
                           double* part_ii) {
  if (num_tasks <= 0) return;
  hipLaunchKernelGGL(k_partial_reduce, dim3((num_tasks + 3) / 4), dim3(256), 0, st, num_tasks, tasks, part_pp, part_ip, part_ii);
}

// Finalize: one 64-lane group per block of S.
//   S_blk = base - sum_chunks partial,   v_rows = base_g - sum_chunks e   (diagonal kinds)
// base (only when add_base, i.e. on one rank): scaled F^T F + D^2 from the camera sums.
__global__ void __launch_bounds__(256) k_schur_finalize(
    int num_blocks, const SchurBlock* __restrict__ blocks, const double* __restrict__ part_pp,
    const double* __restrict__ part_ip, const double* __restrict__ part_ii, int NI, int NC, const int* __restrict__ slot, int nb,
    int add_base, double radius, double dmin, double dmax, const int* __restrict__ img_cam,
    const double* __restrict__ img_rec, const double* __restrict__ cam_rec,
    const double* __restrict__ scale_cam, const int* __restrict__ off_img, const int* __restrict__ off_cam,
    double* __restrict__ S) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bid = blockIdx.x * 4 + wv;
  if (bid >= num_blocks) return;
  const SchurBlock B = blocks[bid];
  const int RX = B.kind == BLK_PP ? 6 : 9, RY = B.kind == BLK_II ? 9 : 6;
  const int NA = RX * RY;
  const bool diag_kind = B.kind != BLK_IP;
  const int PS = NA + (diag_kind ? RX : 0);
  const double* part = B.kind == BLK_PP ? part_pp : B.kind == BLK_IP ? part_ip : part_ii;
  // row0 / col0: the block in the variables' own order (scales, camera sums); prow0 / pcol0: where it
  // goes in the matrix, whose columns follow the elimination order chosen for the factorisation
  const int row0 = B.kind == BLK_PP ? 6 * B.row_ent : 6 * NI + 9 * B.row_ent;
  const int col0 = B.kind == BLK_II ? 6 * NI + 9 * B.col_ent : 6 * B.col_ent;
  const int prow0 = B.kind == BLK_PP ? off_img[B.row_ent] : off_cam[B.row_ent];
  const int pcol0 = B.kind == BLK_II ? off_cam[B.col_ent] : off_img[B.col_ent];
  const bool is_diag = diag_kind && B.row_ent == B.col_ent;
  auto apply = [&](int idx, double s) {
    if (idx < NA) {
      const int r = idx / RY, c = idx - r * RY;
      const int gr = row0 + r, gc = col0 + c;
      double base = 0.0;
      if (add_base) {
        const double sr = scale_cam[gr], sc = scale_cam[gc];
        if (B.kind == BLK_PP && is_diag) {
          const int x = r < c ? r : c, y = r < c ? c : r;
          base = sr * sc * img_rec[(size_t)B.row_ent * kImgRec + sym_idx(x, y, 6)];
          if (r == c && sr != 0.0) base += clampd(base, dmin, dmax) / radius;
        } else if (B.kind == BLK_IP && img_cam[B.col_ent] == B.row_ent) {
          base = sr * sc * img_rec[(size_t)B.col_ent * kImgRec + 27 + c * 9 + r];  // PI[pose c][intr r]
        } else if (B.kind == BLK_II && is_diag) {
          const int x = r < c ? r : c, y = r < c ? c : r;
          base = sr * sc * cam_rec[(size_t)B.row_ent * kCamRec + sym_idx(x, y, 9)];
          if (r == c && sr != 0.0) base += clampd(base, dmin, dmax) / radius;
        }
      }
      double val = base - s;
      // a constant parameter's diagonal entry (its row and column are zero): 1 on the rank that adds the base terms
      // (what k_fix_diag does for the columns no block covers)
      if (is_diag && r == c && scale_cam[gr] == 0.0) val = add_base ? 1.0 : 0.0;
      if (!is_diag || r >= c) {
        // (tile store: only the lower tiles exist; an element of a diagonal tile is written on both sides of the diagonal)
        const int R = prow0 + r, C = pcol0 + c;
        if ((R >> 6) >= (C >> 6)) S[((size_t)slot[(R >> 6) * nb + (C >> 6)] << 12) + (R & 63) * 64 + (C & 63)] = val;
        if ((C >> 6) >= (R >> 6)) S[((size_t)slot[(C >> 6) * nb + (R >> 6)] << 12) + (C & 63) * 64 + (R & 63)] = val;
      }
    } else if (is_diag) {
      const int r = idx - NA;
      const int gr = row0 + r;
      double base = 0.0;
      if (add_base) {
        const double g = B.kind == BLK_PP ? img_rec[(size_t)B.row_ent * kImgRec + 21 + r]
                                          : cam_rec[(size_t)B.row_ent * kCamRec + 45 + r];
        base = scale_cam[gr] * g;
      }
      const int C = prow0 + r;  // right-hand side: first row of tile (nb, C / 64)
      S[((size_t)slot[nb * nb + (C >> 6)] << 12) + (C & 63)] = base - s;
    }
  };
  // one lane per element; the block's chunk partials are added in a fixed order (strided_sum16)
  for (int idx = lane; idx < PS; idx += 64) apply(idx, strided_sum16(part, PS, idx, B.chunk_begin, B.chunk_end));
}
void launch_schur_finalize(hipStream_t st, int num_blocks, const SchurBlock* blocks,
                           const double* part_pp, const double* part_ip, const double* part_ii,
                           int NI, int NC, const int* slot, int nb, bool add_base, double radius, double dmin,
                           double dmax, const int* img_cam, const double* img_rec,
                           const double* cam_rec, const double* scale_cam, const int* off_img, const int* off_cam,
                           double* S) {
  if (num_blocks <= 0) return;
  hipLaunchKernelGGL(k_schur_finalize, dim3((num_blocks + 3) / 4), dim3(256), 0, st, num_blocks, blocks, part_pp,
                     part_ip, part_ii, NI, NC, slot, nb, add_base ? 1 : 0, radius, dmin, dmax, img_cam, img_rec, cam_rec,
                     scale_cam, off_img, off_cam, S);
}
// Multi-rank exchange of the reduced system: only the structurally non-zero lower 64x64 tiles and the right-hand side
// travel. pack: tile tiles[t] of the store -> buf[t * 4096 ...], right-hand side (first rows of the tiles of tile row nb)
// -> buf[num_tiles * 4096 + c]; unpack: the reverse.
__global__ void __launch_bounds__(256) k_tiles_copy(int num_tiles, const int* __restrict__ tiles, const int* __restrict__ slot, int nb,
                                                    double* __restrict__ M, double* __restrict__ buf, int to_buf) {
  const int t = blockIdx.x;
  if (t < num_tiles) {
    double2* a = reinterpret_cast<double2*>(M + ((size_t)tiles[t] << 12));
    double2* b = reinterpret_cast<double2*>(buf + ((size_t)t << 12));
    for (int e = threadIdx.x; e < 2048; e += 256) { if (to_buf) b[e] = a[e]; else a[e] = b[e]; }
  } else {
    double* rhs = buf + ((size_t)num_tiles << 12);
    for (int c = (t - num_tiles) * 256 + threadIdx.x; c < nb * 64; c += (gridDim.x - num_tiles) * 256) {
      double* a = M + ((size_t)slot[nb * nb + (c >> 6)] << 12) + (c & 63);
      if (to_buf) rhs[c] = *a; else *a = rhs[c];
    }
  }
}
void launch_tiles_copy(hipStream_t st, int num_tiles, const int* tiles, const int* slot, int nb, double* M, double* buf, bool to_buf) {
  const int extra = std::max(1, std::min(16, (nb * 64 + 1023) / 1024));
  hipLaunchKernelGGL(k_tiles_copy, dim3(num_tiles + extra), dim3(256), 0, st, num_tiles, tiles, slot, nb, M, buf, to_buf ? 1 : 0);
}

// Constant / unused / padding columns: unit diagonal (their rows and columns are zero).
__global__ void k_fix_diag(int n_mat, const int* __restrict__ slot, int add_one, const int* __restrict__ col_var,
                           const double* __restrict__ scale_cam, double* __restrict__ S) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_mat) return;
  const int j = col_var[t];  // the variable in matrix column t, -1 for padding
  if (j < 0 || scale_cam[j] == 0.0) S[((size_t)slot[(t >> 6) * (n_mat >> 6) + (t >> 6)] << 12) + (t & 63) * 65] = add_one ? 1.0 : 0.0;
}
void launch_fix_diag(hipStream_t st, int n_mat, const int* slot, bool add_one, const int* col_var, const double* scale_cam, double* S) {
  hipLaunchKernelGGL(k_fix_diag, dim3((n_mat + 255) / 256), dim3(256), 0, st, n_mat, slot, add_one ? 1 : 0, col_var, scale_cam, S);
}

// ---------------------------------------------------------------------------
// Back-substitution and candidate point update (one lane per point):
//   y_p = Gi^T ( h - sum_a U_a^T y_cam(a) - sum_q Uk_q^T y_intr(q) ),  step = -y,
//   delta = s_p * step,  X_cand = X + delta.
// partial[b] = { |delta|^2, model-change part, |X_cand|^2 } over free points, where the
// model cost change is 1/2 sum_j y_j (g_j + D_j^2 y_j)  (== -(J step).(r + J step/2) when
// (J^T J + D^2) y = g, which the direct solve guarantees to round-off).
// WITHOUT entry records (the round-1 kernel that read them - 144 B per observation + the intrinsics records - lives in
// scripts/_dbg/pruned_r06.patch): for every observation of the point the Jacobian is recomputed (20 bytes of input) and
//   sum_a U_a^T y_a + sum_q Uk_q^T y_q = Gi ( s_p o sum_obs Jp^T (Jc (s y)_cam + Jk (s y)_intr) ),
// with (s y) = -delta_cam from k_update_cameras (which therefore runs first).
// ONE observation per lane: a work-group takes `ppb` consecutive points (~224 observations) at a time, every lane computes its
// observation's Jp^T (J_cam delta_cam), the three values go through LDS and the point's owner lane adds its observations
// in order (a 16-lanes-per-point version kept 10 of 16 lanes busy at 10 observations per point: 0.126 vs 0.102 ms at C3).
// Round 4: the SQ counters showed the waves parked on memory 70 % of the time (SQ_WAIT_ANY) - with two waves per SIMD a block
// was a chain of five dependent trips to memory (block bounds -> observation -> point / image / camera index -> camera
// tables -> the owner's point data). Now a work-group walks CONSECUTIVE blocks: the next block's bounds and its lanes'
// observations are requested before this block's arithmetic, the owner lanes' point data before the Jacobians, and the
// per-camera tables (intrinsics, their step, the model) sit in LDS - three trips, two of them hidden.
constexpr int kBsCamsLds = 16;  // cameras whose intrinsics tables are staged in LDS (more: read from memory)
__global__ void __launch_bounds__(256) k_backsub_points_packed(
    int NP, int NPs, int NI, int NC, int N, int ppb, int nblocks, double radius, double dmin, double dmax, double loss_b, double loss_inv_b,
    const int* __restrict__ pt_start, const int* __restrict__ obs_img, const int* __restrict__ obs_pt,
    const double2* __restrict__ uv, const int* __restrict__ img_cam, const int* __restrict__ cam_model,
    const double* __restrict__ camrec, const double* __restrict__ intr, const double* __restrict__ delta_cam,
    const unsigned char* __restrict__ pt_free, const double* __restrict__ Gi, const double* __restrict__ h,
    const double* __restrict__ Cu, const double* __restrict__ gu, const double* __restrict__ scale_pt,
    const double* __restrict__ points, double* __restrict__ cand_points, double* __restrict__ delta_points,
    double* __restrict__ partial, const double* __restrict__ cand_camrec, const double* __restrict__ cand_intr,
    const unsigned char* __restrict__ pt_active, double* __restrict__ cost_partial) {
  __shared__ double s_t[3][256];
  __shared__ double s_new[3][128];  // the block's candidate points (ppb <= 128)
  __shared__ double s_red[4];
  __shared__ double s_kin[kBsCamsLds][9], s_dk[kBsCamsLds][9];
  __shared__ int s_model[kBsCamsLds];
  const int tid = threadIdx.x;
  const bool cams_lds = NC <= kBsCamsLds;
  double a_step = 0.0, a_model = 0.0, a_x2 = 0.0, a_cost = 0.0;
  // blocks blockIdx.x, + gridDim.x, ... (work-groups that run side by side walk neighbouring blocks: the points are ordered by
  // their image lists, so they share camera records in the caches - consecutive blocks per work-group lost 8 % at C5)
  const int G = (int)gridDim.x;
  struct Obs { int pt, im; double2 m; };
  auto load_obs = [&](int o) {
    Obs r{0, 0, make_double2(0.0, 0.0)};
    if (N <= 0) return r;
    const int oc = min(o, N - 1);  // (a lane beyond the chunk reads the last observation and drops the result)
    r.pt = obs_pt[oc]; r.im = obs_img[oc]; r.m = uv[oc];
    return r;
  };
  // one observation's three values (zero for a lane beyond the chunk or a point that is not free)
  auto obs_term = [&](const Obs& q, bool valid, double (&t)[3]) {
    t[0] = t[1] = t[2] = 0.0;
    if (!valid || !pt_free[q.pt]) return;
    const int cam = img_cam[q.im];
    const double X[3] = {points[3 * (size_t)q.pt], points[3 * (size_t)q.pt + 1], points[3 * (size_t)q.pt + 2]};
    double rec[9], kin[9], dc[6], dk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rec[k] = camrec[9 * q.im + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) dc[k] = delta_cam[6 * q.im + k];
    int model;
    if (cams_lds) {
      model = s_model[cam];
#pragma unroll
      for (int k = 0; k < 9; ++k) { kin[k] = s_kin[cam][k]; dk[k] = s_dk[cam][k]; }
    } else {
      model = cam_model[cam];
#pragma unroll
      for (int k = 0; k < 9; ++k) { kin[k] = intr[9 * cam + k]; dk[k] = delta_cam[6 * NI + 9 * cam + k]; }
    }
    // (round 5: directional derivatives instead of the full Jacobian - obs_backsub_term, ba_math.h)
    double r[2], tt[3];
#ifndef MAVBA_BS_SKIP  // (timing-only builds, scripts/_dbg/backsub_variants.sh: 1 = no arithmetic, 2 = no owner summation)
#define MAVBA_BS_SKIP 0
#endif
    if (MAVBA_BS_SKIP & 1) { r[0] = rec[0] + kin[0] + dc[0] + dk[0]; r[1] = X[0] + q.m.x; tt[0] = rec[3]; tt[1] = X[1]; tt[2] = q.m.y; }
    else
    obs_backsub_term(model, rec, kin, X, q.m.x, q.m.y, dc, dk, r, tt);
    double w, half_rho;
    cauchy_weight(r[0] * r[0] + r[1] * r[1], loss_b, loss_inv_b, w, half_rho);
    const double w2 = w * w;
    t[0] = w2 * tt[0]; t[1] = w2 * tt[1]; t[2] = w2 * tt[2];
  };
  // bounds of this block and of the next one, observations of this block: requested one / two blocks ahead
  int o0 = 0, o1 = 0, o0n = 0, o1n = 0;
  Obs cur{0, 0, make_double2(0.0, 0.0)};
  if ((int)blockIdx.x < nblocks) {
    const int b0 = (int)blockIdx.x;
    o0 = pt_start[b0 * ppb]; o1 = pt_start[min(b0 * ppb + ppb, NP)];
    if (b0 + G < nblocks) { o0n = pt_start[(b0 + G) * ppb]; o1n = pt_start[min((b0 + G) * ppb + ppb, NP)]; }
    cur = load_obs(o0 + tid);
  }
  if (cams_lds) {  // (behind the first requests: they travel while the tables are staged)
    for (int e = tid; e < 9 * NC; e += 256) { s_kin[e / 9][e % 9] = intr[e]; s_dk[e / 9][e % 9] = delta_cam[6 * NI + e]; }
    for (int c = tid; c < NC; c += 256) s_model[c] = cam_model[c];
    __syncthreads();
  }
  for (int blk = blockIdx.x; blk < nblocks; blk += G) {
    const int p0 = blk * ppb, p1 = min(p0 + ppb, NP);
    const int p = p0 + tid;
    const bool owner = p < p1;
    // requests that do not depend on this block's arithmetic: the next block's observations, the bounds of the one after it,
    // the owner lanes' point data
    int o0nn = 0, o1nn = 0;
    Obs nxt = cur;
    if (blk + G < nblocks) {
      nxt = load_obs(o0n + tid);
      if (blk + 2 * G < nblocks) { o0nn = pt_start[(blk + 2 * G) * ppb]; o1nn = pt_start[min((blk + 2 * G) * ppb + ppb, NP)]; }
    }
    int mb = 0, me = 0;
    bool fr = false;
    double G[6] = {0, 0, 0, 0, 0, 0}, sp[3] = {0, 0, 0}, hh[3] = {0, 0, 0}, cu[3] = {0, 0, 0}, gg[3] = {0, 0, 0}, Xo[3] = {0, 0, 0};
    if (owner) {
      mb = pt_start[p]; me = pt_start[p + 1];
      fr = pt_free[p] != 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) Xo[k] = points[3 * (size_t)p + k];
      if (fr) {
        const int dg[3] = {0, 3, 5};
#pragma unroll
        for (int k = 0; k < 6; ++k) G[k] = Gi[k * NPs + p];
#pragma unroll
        for (int k = 0; k < 3; ++k) { sp[k] = scale_pt[k * NPs + p]; hh[k] = h[k * NPs + p]; cu[k] = Cu[dg[k] * NPs + p]; gg[k] = gu[k * NPs + p]; }
      }
    }
    double T[3] = {0.0, 0.0, 0.0};
    for (int base = o0; base < o1; base += 256) {
      const int o = base + tid;
      const Obs q = base == o0 ? cur : load_obs(o);  // (a block of more than 256 observations - long tracks - reads its further chunks here)
      double t[3];
      obs_term(q, o < o1, t);
      s_t[0][tid] = t[0]; s_t[1][tid] = t[1]; s_t[2][tid] = t[2];
      __syncthreads();
      if (owner) {
        const int b = max(mb, base), e = (MAVBA_BS_SKIP & 2) ? min(me, max(mb, base) + 1) : min(me, base + 256);
        for (int i = b; i < e; ++i) { T[0] += s_t[0][i - base]; T[1] += s_t[1][i - base]; T[2] += s_t[2][i - base]; }
      }
      __syncthreads();
    }
    if (owner) {
      double d[3] = {0, 0, 0};
      if (fr) {
        const double q0 = -sp[0] * T[0], q1 = -sp[1] * T[1], q2 = -sp[2] * T[2];
        const double z[3] = {q0 * G[0], q0 * G[1] + q1 * G[2], q0 * G[3] + q1 * G[4] + q2 * G[5]};
        const double tt[3] = {hh[0] - z[0], hh[1] - z[1], hh[2] - z[2]};
        double yp[3];
        git_mul(G, tt, yp);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double s = sp[k];
          const double D2 = clampd(s * s * cu[k], dmin, dmax) / radius;
          const double gs = s * gg[k];
          a_model += 0.5 * yp[k] * (gs + D2 * yp[k]);
          d[k] = -yp[k] * s;
          a_step += d[k] * d[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double xn = Xo[k] + d[k];
        cand_points[3 * (size_t)p + k] = xn;
        delta_points[3 * (size_t)p + k] = d[k];
        if (fr) a_x2 += xn * xn;
        if (cost_partial) s_new[k][tid] = xn;
      }
    }
    // ---- the candidate's cost over the block's observations (what k_cost_only did in a launch of its own: the pixel and
    // the image index are re-read from the caches, the candidate point comes from LDS, the candidate cameras from memory) ----
    if (cost_partial) {
      __syncthreads();
      for (int base = o0; base < o1; base += 256) {
        const int o = base + tid;
        if (o < o1) {
          const int pt = obs_pt[o], im = obs_img[o];
          const double2 m = uv[o];
          const int cam = img_cam[im];
          double rec[9], kin[9], r[2];
#pragma unroll
          for (int k = 0; k < 9; ++k) rec[k] = cand_camrec[9 * im + k];
#pragma unroll
          for (int k = 0; k < 9; ++k) kin[k] = cand_intr[9 * cam + k];
          const double Xn[3] = {s_new[0][pt - p0], s_new[1][pt - p0], s_new[2][pt - p0]};
          obs_residual(cam_model[cam], rec, kin, Xn, m.x, m.y, r);
          double w, half_rho;
          cauchy_weight(r[0] * r[0] + r[1] * r[1], loss_b, loss_inv_b, w, half_rho);
          if (pt_active && !pt_active[pt]) half_rho = 0.0;  // filtered point: no residual block
          a_cost += half_rho;
        }
      }
      __syncthreads();  // s_new is rewritten by the next block
    }
    o0 = o0n; o1 = o1n; o0n = o0nn; o1n = o1nn; cur = nxt;
  }
  const double s0 = block_sum_256(a_step, s_red);
  const double s1 = block_sum_256(a_model, s_red);
  const double s2 = block_sum_256(a_x2, s_red);
  if (threadIdx.x == 0) { partial[3 * blockIdx.x] = s0; partial[3 * blockIdx.x + 1] = s1; partial[3 * blockIdx.x + 2] = s2; }
  if (cost_partial) {
    const double s3 = block_sum_256(a_cost, s_red);
    if (threadIdx.x == 0) cost_partial[blockIdx.x] = s3;
  }
}
int backsub_points_grid(int NP) {
  int gp = (NP + 15) / 16;
  if (gp > 1024) gp = 1024;
  return gp < 1 ? 1 : gp;
}
void launch_backsub_points_jvp(hipStream_t st, int NP, int NPs, int NI, double radius, double dmin, double dmax,
                               const SweepArgs& a, const int* pt_start, const double* delta_cam,
                               const unsigned char* pt_free, const double* Gi, const double* h, const double* Cu,
                               const double* gu, const double* scale_pt, double* cand_points, double* delta_points,
                               double* partial, const double* cand_camrec, const double* cand_intr, double* cost_partial) {
  const int gp = backsub_points_grid(NP);
  // points per work-group: ~224 observations at the problem's average track length (the owner lanes are the first ppb)
  const double track = NP > 0 ? (double)a.N / NP : 1.0;
  const int ppb = std::max(1, std::min(128, (int)(224.0 / std::max(track, 1.0))));
  const int nblocks = (NP + ppb - 1) / ppb;
  hipLaunchKernelGGL(k_backsub_points_packed, dim3(gp), dim3(256), 0, st, NP, NPs, NI, a.NC, a.N, ppb, nblocks, radius, dmin, dmax, a.loss_b,
                     a.loss_inv_b, pt_start, a.obs_img, a.obs_pt, a.uv, a.img_cam, a.cam_model, a.camrec, a.intr, delta_cam, pt_free,
                     Gi, h, Cu, gu, scale_pt, a.points, cand_points, delta_points, partial, cand_camrec, cand_intr, a.pt_active,
                     cost_partial);
}

// Camera columns: step = -y, delta = s * step, candidate parameters (update_cameras_body, lm_bodies.h: work-group b takes
// 42 images - one work-group looping over all 6 NI + 9 NC parameters took 22 us at C3).
int update_cameras_groups(int NI) { return std::max(1, (NI + kUpdImagesPerGroup - 1) / kUpdImagesPerGroup); }
__global__ void __launch_bounds__(256) k_update_cameras(
    int NI, int NC, int cam_part, double radius, double dmin, double dmax, const double* __restrict__ y,
    const double* __restrict__ scale_cam, const double* __restrict__ img_rec, const double* __restrict__ cam_rec,
    const double* __restrict__ poses, const double* __restrict__ intr, double* cand_poses,
    double* __restrict__ cand_intr, double* __restrict__ delta_cam, double* __restrict__ partial3,
    double* __restrict__ cand_camrec) {
  __shared__ double s_red[4];
  update_cameras_body(blockIdx.x, NI, NC, cam_part, radius, dmin, dmax, y, scale_cam, img_rec, cam_rec, poses, intr, cand_poses, cand_intr,
                      delta_cam, partial3, cand_camrec, s_red);
}
void launch_update_cameras(hipStream_t st, int NI, int NC, bool cam_part, double radius, double dmin,
                           double dmax, const double* y, const double* scale_cam,
                           const double* img_rec, const double* cam_rec, const double* poses,
                           const double* intr, double* cand_poses, double* cand_intr,
                           double* delta_cam, double* partial3, double* cand_camrec) {
  hipLaunchKernelGGL(k_update_cameras, dim3(update_cameras_groups(NI)), dim3(256), 0, st, NI, NC, cam_part ? 1 : 0, radius, dmin,
                     dmax, y, scale_cam, img_rec, cam_rec, poses, intr, cand_poses, cand_intr, delta_cam, partial3, cand_camrec);
}

// out[c] (op)= reduce over rows of partial[row*stride + c]; single block, fixed order.
__global__ void __launch_bounds__(256) k_reduce_cols(const double* __restrict__ partial, int rows, int cols,
                                                     int stride, unsigned max_mask, double* __restrict__ out,
                                                     int accumulate) {
  __shared__ double s_red[4];
  for (int c = 0; c < cols; ++c) {
    const bool mx = (max_mask >> c) & 1u;
    double v = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
      const double x = partial[(size_t)r * stride + c];
      v = mx ? fmax(v, x) : v + x;
    }
    const double t = mx ? block_max_256(v, s_red) : block_sum_256(v, s_red);
    if (threadIdx.x == 0) {
      if (accumulate) out[c] = mx ? fmax(out[c], t) : out[c] + t;
      else out[c] = t;
    }
    __syncthreads();
  }
}
void launch_reduce_cols(hipStream_t st, const double* partial, int rows, int cols, int stride,
                        unsigned max_mask, double* out, bool accumulate) {
  hipLaunchKernelGGL(k_reduce_cols, dim3(1), dim3(256), 0, st, partial, rows, cols, stride, max_mask, out, accumulate ? 1 : 0);
}

// Several independent single-block reductions in ONE launch (the scalars of an LM phase): block b handles task b,
//   out = op(src[r * stride], r < rows)  (+ sum of src2[r], r < rows2, for the cost = observations + priors).
__global__ void __launch_bounds__(256) k_reduce_tasks(ReduceTasks T, LmSpec spec) {
  __shared__ double s_red[4];
  if (!lm_spec_go(spec, nullptr)) return;
  reduce_task_body(T.t[blockIdx.x], s_red);
}
// One wave: the LM decision on the device and the scalars to the host (lm_snapshot_body, lm_bodies.h).
__global__ void __launch_bounds__(64) k_lm_snapshot(LmSpec spec, double* dec, double* host_pub, double seq, double* fail_slots) {
  lm_snapshot_body(threadIdx.x, spec, dec, host_pub, seq, fail_slots);
}
// The end of a candidate step in ONE work-group: its (up to four) scalar reductions side by side, then (first wave) the
// decision and the publication - k_reduce_tasks + k_lm_snapshot without the launch in between.
__global__ void __launch_bounds__(1024) k_lm_tail(ReduceTasks T, int n, LmSpec spec, double* dec, double* host_pub, double seq, double* fail_slots) {
  __shared__ double s_t[4][2][4];
  reduce_tasks_grouped(T, n, s_t);
  if (threadIdx.x < 64) lm_snapshot_body(threadIdx.x, spec, dec, host_pub, seq, fail_slots);
}
// The tail of an evaluation for a small problem in ONE work-group of 1024 lanes: per-image sums (one lane per element),
// per-camera sums, norms (four of k_state_norms' groups at a time), the three scalar reductions (side by side) -
// k_camera_reduce_img + k_camera_reduce_cam + k_state_norms + k_reduce_tasks: 17 us of launches for ~4 us of work.
constexpr int kEvalSmallMaxGroups = 40;  // 8192 points / 256 + camera groups
__global__ void __launch_bounds__(1024) k_eval_small(EvalSmallArgs a) {
  __shared__ double s_part[16][64];
  __shared__ double s_w[kEvalSmallMaxGroups][2][4];
  __shared__ double s_t[4][2][4];
  if (!lm_spec_go(a.spec, nullptr)) return;
  const int tid = threadIdx.x, g = tid >> 8, lt = tid & 255, wv = (tid >> 6) & 3, lane = tid & 63;
  for (int q = tid; q < a.NI * kSweepAcc; q += 1024)
    camera_reduce_img_elem(q / kSweepAcc, q % kSweepAcc, a.img_chunk_start, a.cam_partial, a.prior_start, a.prior_res, a.prior_jac, a.img_rec,
                           a.img_intr_tmp);
  __syncthreads();
  if (a.with_cams) {
    for (int c = 0; c < a.NC; ++c) {
      if (lane < kCamRec) s_part[tid >> 6][lane] = camera_reduce_cam_part(c, tid >> 6, lane, a.cam_img_start, a.cam_imgs, a.img_intr_tmp);
      __syncthreads();
      if (tid < kCamRec) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_part[k][tid];
        a.cam_rec[(size_t)c * kCamRec + tid] = tot;
      }
      __syncthreads();
    }
  }
  const int nvb = a.gp + a.gc;
  for (int vb0 = 0; vb0 < nvb; vb0 += 4) {
    const int vb = vb0 + g;
    if (vb < nvb) {
      double gmax, x2;
      state_norms_local(vb, lt, a.gp, a.gc, a.NI, a.NC, a.NP, a.NPs, a.cam_part, a.pose_free, a.intr_free, a.pt_free, a.poses, a.intr, a.points,
                        a.img_rec, a.cam_rec, a.gu, gmax, x2);
      const double m = wave_max(gmax), sum = wave_sum(x2);
      if (lane == 0) { s_w[vb][0][wv] = m; s_w[vb][1][wv] = sum; }
    }
  }
  __syncthreads();
  if (tid < nvb) { a.norm_partial[2 * tid] = group4_max(s_w[tid][0]); a.norm_partial[2 * tid + 1] = group4_sum(s_w[tid][1]); }
  __syncthreads();
  reduce_tasks_grouped(a.T, a.num_tasks, s_t);
}
void launch_eval_small(hipStream_t st, const EvalSmallArgs& a) { hipLaunchKernelGGL(k_eval_small, dim3(1), dim3(1024), 0, st, a); }
// Round 5: the tail of an evaluation of a LARGE problem was four dependent launches of 5-7 us each (k_camera_reduce_img,
// k_camera_reduce_cam, k_state_norms, k_reduce_tasks) for a few microseconds of work. Two launches: k_eval_head - the
// per-image sums (four images per 256-lane work-group) and the POINTS' norm groups, independent of each other, in one
// grid -, k_eval_tail - one 1024-lane work-group: per-camera sums, the CAMERAS' norm groups (they need those sums),
// the three reductions over the partials of both kernels. The same device functions in the same order as the launches they
// replace (lm_bodies.h): bit-identical results. Not with shards (the camera sums are all-reduced between the two halves).
__global__ void __launch_bounds__(256) k_eval_head(EvalSmallArgs a, int img_blocks) {
  __shared__ double s_red[4];
  if (!lm_spec_go(a.spec, nullptr)) return;
  if ((int)blockIdx.x < img_blocks) {
    const int i = 4 * blockIdx.x + (threadIdx.x >> 6);
    if (i < a.NI) camera_reduce_img_body(i, threadIdx.x & 63, a.img_chunk_start, a.cam_partial, a.prior_start, a.prior_res, a.prior_jac, a.img_rec, a.img_intr_tmp);
    return;
  }
  if ((int)blockIdx.x >= img_blocks + a.gp) {  // (the pre-reduction's tasks riding along, four per work-group)
    const int task = 4 * ((int)blockIdx.x - img_blocks - a.gp) + (threadIdx.x >> 6);
    if (task < a.ride.n) partial_reduce_task(a.ride.tasks[task], threadIdx.x & 63, a.ride.pp, a.ride.ip, a.ride.ii);
    return;
  }
  state_norms_body(blockIdx.x - img_blocks, a.gp, a.gc, a.NI, a.NC, a.NP, a.NPs, a.cam_part, a.pose_free, a.intr_free, a.pt_free, a.poses, a.intr,
                   a.points, a.img_rec, a.cam_rec, a.gu, a.norm_partial, s_red);
}
__global__ void __launch_bounds__(1024) k_eval_tail(EvalSmallArgs a) {
  __shared__ double s_part[16][64];
  __shared__ double s_w[kStateNormsCamBlocks][2][4];
  __shared__ double s_t[4][2][4];
  if (!lm_spec_go(a.spec, nullptr)) return;
  const int tid = threadIdx.x, g = tid >> 8, lt = tid & 255, wv = (tid >> 6) & 3, lane = tid & 63;
  if (a.with_cams) {
    for (int c = 0; c < a.NC; ++c) {
      if (lane < kCamRec) s_part[tid >> 6][lane] = camera_reduce_cam_part(c, tid >> 6, lane, a.cam_img_start, a.cam_imgs, a.img_intr_tmp);
      __syncthreads();
      if (tid < kCamRec) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_part[k][tid];
        a.cam_rec[(size_t)c * kCamRec + tid] = tot;
      }
      __syncthreads();
    }
  }
  const int nvb = a.gp + a.gc;
  for (int vb0 = a.first_group; vb0 < nvb; vb0 += 4) {
    const int vb = vb0 + g;
    if (vb < nvb) {
      double gmax, x2;
      state_norms_local(vb, lt, a.gp, a.gc, a.NI, a.NC, a.NP, a.NPs, a.cam_part, a.pose_free, a.intr_free, a.pt_free, a.poses, a.intr, a.points,
                        a.img_rec, a.cam_rec, a.gu, gmax, x2);
      const double m = wave_max(gmax), sum = wave_sum(x2);
      if (lane == 0) { s_w[vb - a.first_group][0][wv] = m; s_w[vb - a.first_group][1][wv] = sum; }
    }
  }
  __syncthreads();
  if (tid < nvb - a.first_group) {
    a.norm_partial[2 * (a.first_group + tid)] = group4_max(s_w[tid][0]);
    a.norm_partial[2 * (a.first_group + tid) + 1] = group4_sum(s_w[tid][1]);
  }
  __syncthreads();
  reduce_tasks_grouped(a.T, a.num_tasks, s_t);
}
void launch_eval_head_tail(hipStream_t st, const EvalSmallArgs& a) {
  const int img_blocks = (a.NI + 3) / 4;
  hipLaunchKernelGGL(k_eval_head, dim3(img_blocks + a.gp + (a.ride.n + 3) / 4), dim3(256), 0, st, a, img_blocks);
  EvalSmallArgs b = a;
  b.first_group = a.gp;
  hipLaunchKernelGGL(k_eval_tail, dim3(1), dim3(1024), 0, st, b);
}
bool eval_head_tail_fits(int gc) { return gc <= kStateNormsCamBlocks; }
void launch_lm_tail(hipStream_t st, const ReduceTasks& T, int n, const LmSpec& spec, double* dec, double* host_pub, double seq, double* fail_slots) {
  if (n > 4) {  // (the merged kernel runs at most four reductions side by side)
    launch_reduce_tasks(st, T, n);
    launch_lm_snapshot(st, spec, dec, host_pub, seq, fail_slots);
    return;
  }
  hipLaunchKernelGGL(k_lm_tail, dim3(1), dim3(1024), 0, st, T, n, spec, dec, host_pub, seq, fail_slots);
}
// Test entry (mavba_debug_lm_decide): `n` decisions by the device build of lm_decide. in: SC_COUNT scalars + 8 parameters
// (radius, decrease_factor, ptol, ftol, min_rel_dec, max_radius, abs_gtol, pending_eval) per case; out: 6 doubles per case.
__global__ void k_lm_decide_cases(int n, const double* __restrict__ in, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const double* q = in + (size_t)c * (SC_COUNT + 8);
  LmSpec sp = lm_spec_off();
  sp.radius = q[SC_COUNT]; sp.decrease_factor = q[SC_COUNT + 1]; sp.ptol = q[SC_COUNT + 2]; sp.ftol = q[SC_COUNT + 3];
  sp.min_rel_dec = q[SC_COUNT + 4]; sp.max_radius = q[SC_COUNT + 5]; sp.abs_gtol = q[SC_COUNT + 6]; sp.pending_eval = q[SC_COUNT + 7] != 0.0;
  const LmDecision d = lm_decide(q, sp);
  double* o = out + (size_t)c * 6;
  o[0] = (double)d.code; o[1] = d.radius; o[2] = d.decrease_factor; o[3] = d.rel; o[4] = d.step_norm; o[5] = d.cost_change;
}
void launch_lm_decide_cases(hipStream_t st, int n, const double* in, double* out) {
  if (n > 0) hipLaunchKernelGGL(k_lm_decide_cases, dim3((n + 255) / 256), dim3(256), 0, st, n, in, out);
}
void launch_lm_snapshot(hipStream_t st, const LmSpec& spec, double* dec, double* host_pub, double seq, double* fail_slots) {
  hipLaunchKernelGGL(k_lm_snapshot, dim3(1), dim3(64), 0, st, spec, dec, host_pub, seq, fail_slots);
}
void launch_reduce_tasks(hipStream_t st, const ReduceTasks& T, int n, const LmSpec& spec) {
  if (n > 0) hipLaunchKernelGGL(k_reduce_tasks, dim3(n), dim3(256), 0, st, T, spec);
}

// read-back of per-point values in the caller's order: out[orig[q]][e] = in[q][e]
__global__ void k_points_to_caller(int NP, int width, const int* __restrict__ orig, const double* __restrict__ in, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NP * width) return;
  const int q = t / width, e = t - q * width;
  out[(size_t)orig[q] * width + e] = in[t];
}
void launch_points_to_caller(hipStream_t st, int NP, int width, const int* orig, const double* in, double* out) {
  if (NP <= 0) return;
  hipLaunchKernelGGL(k_points_to_caller, dim3((NP * width + 255) / 256), dim3(256), 0, st, NP, width, orig, in, out);
}

// point3D_errors: sum |r_raw| / count over a point's observations (bundle_adjustment.cc:590-596)
__global__ void k_point_errors(int NP, const int* __restrict__ pt_start, const double* __restrict__ rnorm,
                               const int* __restrict__ pt_count, double* __restrict__ perr) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NP) return;
  const double cnt = (double)pt_count[p];
  double s = 0.0;
  for (int o = pt_start[p]; o < pt_start[p + 1]; ++o) s += rnorm[o] / cnt;
  perr[p] = s;
}
void launch_point_errors(hipStream_t st, int NP, const int* pt_start, const double* rnorm,
                         const int* pt_count, double* perr) {
  if (NP <= 0) return;
  hipLaunchKernelGGL(k_point_errors, dim3((NP + 255) / 256), dim3(256), 0, st, NP, pt_start, rnorm, pt_count, perr);
}

}  // namespace mavba
