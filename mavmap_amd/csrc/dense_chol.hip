// dense_chol.hip — dense SPD solve of the reduced camera system on gfx950.
//
// Replaces what the reference gets from Ceres' SPARSE_SCHUR back end (CHOLMOD
// factorisation of the reduced camera matrix, reference
// src/base3d/bundle_adjustment.cc:555). Right-looking blocked Cholesky, NB = 64:
//
//   diag    a 64x64 diagonal tile is factorised AND inverted by one wave with its row held
//           in registers (fully unrolled, the other row broadcast from LDS);
//   trsm    A_ik <- A_ik L_kk^-T is then a 64x64x64 product on the FP64 matrix cores
//           (v_mfma_f64_16x16x4_f64) against the explicit inverse;
//   update  trailing tiles C_ij -= A_ik A_jk^T, also FP64 MFMA; the work-group that owns
//           tile (k+1, k+1) goes on to factorise + invert it, so the latency-bound tile
//           factorisation of the NEXT panel overlaps the rest of this panel's update.
//
// The right-hand side rides along as an extra row block below the matrix, so the forward
// substitution is free; the backward substitution is one small launch per tile.
#include "internal.h"

namespace mavba {

namespace {
constexpr int NB = 64;
constexpr int GLD = 66;  // LDS pitch (doubles): (2*row + k) % 32 distinct per 32-lane group, 16 B rows

typedef double d4 __attribute__((ext_vector_type(4)));

// 1/sqrt(d) to full FP64 precision: v_rsq_f64 seed + two Newton steps (no divide, no sqrt
// expansion on the factorisation's critical path).
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * (1.5 - hd * y * y);
  y = y * (1.5 - hd * y * y);
  return y;
}

// Factorise the SPD tile in T (LDS, pitch GLD) and invert the factor:
//   T    <- L   (lower; strict upper zeroed),   Ti <- L^-1 (lower).
// Executed by wave 0 of a 256-thread work-group; every thread must call it (uniform
// barriers). scr = 192 doubles of LDS scratch (two column buffers + reciprocal diagonal).
// Right-looking with the lane's row in registers: per column one pivot broadcast, one
// rsqrt, one LDS column exchange, then 63-j INDEPENDENT fused multiply-adds.
// Returns false in wave 0 if a pivot is not positive.
__device__ __forceinline__ bool tile_potrf_inv(double* T, double* Ti, double* scr, int tid) {
  const bool w0 = tid < 64;
  const int lane = tid & 63;
  double* rd = scr + 128;
  bool ok = true;
  double reg[NB];  // wave 0: row `lane` of the (partially updated) tile, later column `lane` of L^-1
  if (w0) {
#pragma unroll
    for (int k = 0; k < NB; k += 2) {
      const double2 v = *reinterpret_cast<const double2*>(T + lane * GLD + k);
      reg[k] = v.x; reg[k + 1] = v.y;
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double* col = scr + (j & 1) * NB;
    if (w0) {
      double piv = __shfl(reg[j], j, 64);
      if (!(piv > 0.0) || !isfinite(piv)) { ok = false; piv = 1.0; }
      const double rs = rsqrt_nr(piv);
      const double l = lane == j ? piv * rs : (lane > j ? reg[j] * rs : 0.0);
      reg[j] = l;
      col[lane] = l;
      if (lane == j) rd[j] = rs;
    }
    __syncthreads();
    if (w0) {
      const double l = reg[j];
#pragma unroll
      for (int k = j + 1; k < NB; ++k) {
        reg[k] -= l * col[k];
      }
    }
  }
  if (w0) {
#pragma unroll
    for (int k = 0; k < NB; ++k) T[lane * GLD + k] = k <= lane ? reg[k] : 0.0;
  }
  __syncthreads();
  // column `lane` of X = L^-1:  x_r = (delta_rc - sum_{m<r} L[r][m] x_m) / L[r][r]
  if (w0) {
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const double* rr = T + r * GLD;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int m = 0; m + 3 < r; m += 4) {
        const double2 b0 = *reinterpret_cast<const double2*>(rr + m);
        const double2 b1 = *reinterpret_cast<const double2*>(rr + m + 2);
        s0 += b0.x * reg[m]; s1 += b0.y * reg[m + 1];
        s2 += b1.x * reg[m + 2]; s3 += b1.y * reg[m + 3];
      }
#pragma unroll
      for (int m = (r / 4) * 4; m < r; ++m) s0 += rr[m] * reg[m];
      reg[r] = ((r == lane ? 1.0 : 0.0) - ((s0 + s1) + (s2 + s3))) * rd[r];
      Ti[r * GLD + lane] = reg[r];
    }
  }
  __syncthreads();
  return ok;
}

// acc (2x2 MFMA tiles of the wave's 32x32 quadrant) = As(rows wr..) * Bs(rows wc..)^T, K = 64.
__device__ __forceinline__ void mfma_quadrant_nt(const double* As, const double* Bs, int wr, int wc,
                                                 int lane, d4 acc[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < NB; kk += 4) {
    const double a0 = As[(wr + li) * GLD + kk + lk];
    const double a1 = As[(wr + 16 + li) * GLD + kk + lk];
    const double b0 = Bs[(wc + li) * GLD + kk + lk];
    const double b1 = Bs[(wc + 16 + li) * GLD + kk + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// 64x64 tile, global (leading dimension ld) -> LDS (pitch GLD), 256 threads, 16 B accesses.
__device__ __forceinline__ void load_tile(const double* __restrict__ G, size_t ld, double* S, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    const double2 v = *reinterpret_cast<const double2*>(G + (size_t)row * ld + c2);
    *reinterpret_cast<double2*>(S + row * GLD + c2) = v;
  }
}
__device__ __forceinline__ void store_tile(double* __restrict__ G, size_t ld, const double* S, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(G + (size_t)row * ld + c2) = *reinterpret_cast<const double2*>(S + row * GLD + c2);
  }
}
}  // namespace

// Factor + invert diagonal tile 0 (one work-group).
__global__ void __launch_bounds__(256) k_chol_diag0(const double* __restrict__ M, int ld,
                                                    double* __restrict__ diag, double* __restrict__ inv,
                                                    double* __restrict__ fail) {
  __shared__ __attribute__((aligned(16))) double T[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ti[NB * GLD];
  __shared__ __attribute__((aligned(16))) double rd[3 * NB];
  const int tid = threadIdx.x;
  load_tile(M, ld, T, tid);
  for (int i = tid; i < NB * GLD; i += 256) Ti[i] = 0.0;
  __syncthreads();
  const bool ok = tile_potrf_inv(T, Ti, rd, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(diag, NB, T, tid);
  store_tile(inv, NB, Ti, tid);
}

// Panel k: row block k+1+blockIdx.x (the last one is the right-hand-side block) of panel k
//   A_ik <- A_ik L_kk^-T = A_ik (L_kk^-1)^T
__global__ void __launch_bounds__(256) k_chol_trsm(double* __restrict__ M, int ld, int k,
                                                   const double* __restrict__ inv) {
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  const int tid = threadIdx.x;
  const int i = k + 1 + blockIdx.x;
  double* A = M + (size_t)i * NB * ld + (size_t)k * NB;
  load_tile(A, ld, As, tid);
  load_tile(inv + (size_t)k * NB * NB, NB, Bs, tid);
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  d4 acc[2][2];
  mfma_quadrant_nt(As, Bs, wr, wc, lane, acc);
  // D layout of v_mfma_f64_16x16x4_f64: reg r of lane l holds D[(l >> 4) + 4 r][l & 15]
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        A[(size_t)(wr + 16 * m + lk + 4 * r) * ld + wc + 16 * n + li] = acc[m][n][r];
}

// Trailing update, panel k: tile (i, j) -= A_ik A_jk^T for k < j <= i (i may be the
// right-hand-side block). The owner of tile (k+1, k+1) then factorises + inverts it.
__global__ void __launch_bounds__(256) k_chol_update(double* __restrict__ M, int ld, int k,
                                                     double* __restrict__ diag, double* __restrict__ inv,
                                                     double* __restrict__ fail) {
  const int j = k + 1 + blockIdx.x, i = k + 1 + blockIdx.y;
  if (j > i) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double rd[3 * NB];
  const int tid = threadIdx.x;
  load_tile(M + (size_t)i * NB * ld + (size_t)k * NB, ld, As, tid);
  load_tile(M + (size_t)j * NB * ld + (size_t)k * NB, ld, Bs, tid);
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  d4 acc[2][2];
  mfma_quadrant_nt(As, Bs, wr, wc, lane, acc);
  const int li = lane & 15, lk = lane >> 4;
  double* C = M + (size_t)i * NB * ld + (size_t)j * NB;
  const bool next_diag = (blockIdx.x == 0 && blockIdx.y == 0);
  if (!next_diag) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t off = (size_t)(wr + 16 * m + lk + 4 * r) * ld + wc + 16 * n + li;
          C[off] -= acc[m][n][r];
        }
    return;
  }
  // tile (k+1, k+1): keep the updated tile in LDS, factorise + invert it for the next panel
  __syncthreads();  // everyone is done reading As / Bs
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * m + lk + 4 * r, col = wc + 16 * n + li;
        As[row * GLD + col] = C[(size_t)row * ld + col] - acc[m][n][r];
      }
  for (int t = tid; t < NB * GLD; t += 256) Bs[t] = 0.0;
  __syncthreads();
  const bool ok = tile_potrf_inv(As, Bs, rd, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(diag + (size_t)(k + 1) * NB * NB, NB, As, tid);
  store_tile(inv + (size_t)(k + 1) * NB * NB, NB, Bs, tid);
}

// Backward substitution, tile k: y_k = L_kk^-T z_k; then z_j -= L_kj^T y_k for j < k.
// grid = k + 1: block k stores y_k, block j < k updates z_j (disjoint segments).
__global__ void __launch_bounds__(64) k_chol_backsolve(const double* __restrict__ M, int ld, int k,
                                                       const double* __restrict__ inv,
                                                       double* __restrict__ z, double* __restrict__ y) {
  __shared__ double zk[NB];
  __shared__ double yk[NB];
  const int lane = threadIdx.x;
  zk[lane] = z[k * NB + lane];
  __syncthreads();
  const double* Li = inv + (size_t)k * NB * NB;  // L_kk^-1, lower, compact 64x64
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; m += 2) {
    acc0 += Li[m * NB + lane] * zk[m];
    acc1 += Li[(m + 1) * NB + lane] * zk[m + 1];
  }
  const double mine = acc0 + acc1;  // (L^-T z)_lane = sum_m Linv[m][lane] z_m
  if ((int)blockIdx.x == k) { y[k * NB + lane] = mine; return; }
  yk[lane] = mine;
  __syncthreads();
  const int j = blockIdx.x;
  const double* L = M + (size_t)k * NB * ld + (size_t)j * NB;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; m += 2) {
    a0 += L[(size_t)m * ld + lane] * yk[m];
    a1 += L[(size_t)(m + 1) * ld + lane] * yk[m + 1];
  }
  z[j * NB + lane] -= a0 + a1;
}

// diag_ws: 2 * n_pad * 64 doubles (factor tiles, then their inverses).
void dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* diag_ws) {
  const int nb = n_pad / NB, ld = n_pad;
  double* diag = diag_ws;
  double* inv = diag_ws + (size_t)n_pad * NB;
  hipLaunchKernelGGL(k_chol_diag0, dim3(1), dim3(256), 0, st, M, ld, diag, inv, fail);
  for (int k = 0; k < nb; ++k) {
    const int below = nb - 1 - k;  // real row blocks below tile k
    hipLaunchKernelGGL(k_chol_trsm, dim3(below + 1), dim3(256), 0, st, M, ld, k, inv);
    if (below > 0)
      hipLaunchKernelGGL(k_chol_update, dim3(below, below + 1), dim3(256), 0, st, M, ld, k, diag, inv, fail);
  }
  double* z = M + (size_t)n_pad * ld;
  for (int k = nb - 1; k >= 0; --k)
    hipLaunchKernelGGL(k_chol_backsolve, dim3(k + 1), dim3(64), 0, st, M, ld, k, inv, z, y);
}

}  // namespace mavba
