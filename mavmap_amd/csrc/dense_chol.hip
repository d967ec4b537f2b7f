// dense_chol.hip — dense SPD solve of the reduced camera system on gfx950.
//
// Replaces what the reference gets from Ceres' SPARSE_SCHUR back end (CHOLMOD
// factorisation of the reduced camera matrix, reference
// src/base3d/bundle_adjustment.cc:555). Right-looking blocked Cholesky, NB = 64:
//   panel   every work-group re-factorises the 64x64 diagonal tile in LDS (one wave,
//           left-looking) and then solves its own 64-row block of the panel against it;
//   update  trailing tiles C_ij -= A_ik A_jk^T on the FP64 matrix cores
//           (v_mfma_f64_16x16x4_f64), operands staged in LDS with a conflict-free pitch.
// The right-hand side rides along as an extra row block below the matrix, so the
// forward substitution is free; the backward substitution is one small launch per tile.
#include "internal.h"

namespace mavba {

namespace {
constexpr int NB = 64;
constexpr int PLD = 65;  // LDS pitch (doubles) for the lane-per-row kernels: (i*65 + m) % 32 distinct
constexpr int GLD = 66;  // LDS pitch for MFMA operand tiles: (2*row + k) % 32 distinct per 32 lanes

typedef double d4 __attribute__((ext_vector_type(4)));

// Factorise the 64x64 tile held in T (pitch PLD) in place: lower triangle <- L.
// One wave; lane = row. Returns false (wave-uniform) if a pivot is not positive.
__device__ __forceinline__ bool tile_potrf(double* T, int lane) {
  bool ok = true;
  for (int j = 0; j < NB; ++j) {
    double s = T[lane * PLD + j];
    const double* ri = T + lane * PLD;
    const double* rj = T + j * PLD;
    int m = 0;
    for (; m + 4 <= j; m += 4)
      s -= ri[m] * rj[m] + ri[m + 1] * rj[m + 1] + ri[m + 2] * rj[m + 2] + ri[m + 3] * rj[m + 3];
    for (; m < j; ++m) s -= ri[m] * rj[m];
    double d = __shfl(s, j, 64);
    if (!(d > 0.0) || !isfinite(d)) { ok = false; d = 1.0; }
    const double rs = 1.0 / sqrt(d);
    if (lane == j) T[lane * PLD + j] = sqrt(d);
    else if (lane > j) T[lane * PLD + j] = s * rs;
    __syncthreads();
  }
  return ok;
}
}  // namespace

// grid = 1 + (number of row blocks below tile k, including the right-hand-side block).
// Block 0 stores L_kk into diag[k]; block b >= 1 overwrites row block k + b of panel k with
// A_ik L_kk^-T.
__global__ void __launch_bounds__(64) k_chol_panel(double* __restrict__ M, int ld, int k,
                                                   double* __restrict__ diag, double* __restrict__ fail) {
  __shared__ double T[NB * PLD];
  __shared__ double X[NB * PLD];
  const int lane = threadIdx.x;
  const double* A = M + (size_t)k * NB * ld + (size_t)k * NB;
  for (int r = 0; r < NB; ++r) T[r * PLD + lane] = A[(size_t)r * ld + lane];
  __syncthreads();
  const bool ok = tile_potrf(T, lane);
  if (blockIdx.x == 0) {
    if (!ok && lane == 0) atomicAdd(fail, 1.0);
    double* D = diag + (size_t)k * NB * NB;
    for (int r = 0; r < NB; ++r) D[r * NB + lane] = (lane <= r) ? T[r * PLD + lane] : 0.0;
    return;
  }
  const int rb = k + blockIdx.x;
  double* B = M + (size_t)rb * NB * ld + (size_t)k * NB;
  for (int r = 0; r < NB; ++r) X[r * PLD + lane] = B[(size_t)r * ld + lane];
  __syncthreads();
  // row `lane`: x_c = (a_c - sum_{m<c} x_m L[c][m]) / L[c][c]
  double* xr = X + lane * PLD;
  for (int c = 0; c < NB; ++c) {
    const double* lc = T + c * PLD;
    double s = xr[c];
    int m = 0;
    for (; m + 4 <= c; m += 4)
      s -= xr[m] * lc[m] + xr[m + 1] * lc[m + 1] + xr[m + 2] * lc[m + 2] + xr[m + 3] * lc[m + 3];
    for (; m < c; ++m) s -= xr[m] * lc[m];
    xr[c] = s / lc[c];
  }
  __syncthreads();
  for (int r = 0; r < NB; ++r) B[(size_t)r * ld + lane] = X[r * PLD + lane];
}

// Trailing update with FP64 MFMA: tile (i, j) -= A_ik A_jk^T for k < j <= i.
// grid.x = column tiles (j = k+1+x), grid.y = row tiles (i = k+1+y; the last one is the
// right-hand-side block). 4 waves, each a 32x32 quadrant = 2x2 MFMA tiles.
__global__ void __launch_bounds__(256) k_chol_update(double* __restrict__ M, int ld, int k) {
  const int j = k + 1 + blockIdx.x, i = k + 1 + blockIdx.y;
  if (j > i) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  const int tid = threadIdx.x;
  const double* Ai = M + (size_t)i * NB * ld + (size_t)k * NB;
  const double* Aj = M + (size_t)j * NB * ld + (size_t)k * NB;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    const double2 va = *reinterpret_cast<const double2*>(Ai + (size_t)row * ld + c2);
    const double2 vb = *reinterpret_cast<const double2*>(Aj + (size_t)row * ld + c2);
    As[row * GLD + c2] = va.x; As[row * GLD + c2 + 1] = va.y;
    Bs[row * GLD + c2] = vb.x; Bs[row * GLD + c2 + 1] = vb.y;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  d4 acc[2][2];
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
  // D layout of v_mfma_f64_16x16x4_f64: reg r of lane l holds D[(l >> 4) + 4 r][l & 15]
  double* C = M + (size_t)i * NB * ld + (size_t)j * NB;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * m + lk + 4 * r, col = wc + 16 * n + li;
        C[(size_t)row * ld + col] -= acc[m][n][r];
      }
}

// Backward substitution, tile k: y_k = L_kk^-T z_k; then z_j -= L_kj^T y_k for j < k.
// grid = k + 1: block k stores y_k, block j < k updates z_j (disjoint segments).
__global__ void __launch_bounds__(64) k_chol_backsolve(const double* __restrict__ M, int ld, int k,
                                                       const double* __restrict__ diag,
                                                       double* __restrict__ z, double* __restrict__ y) {
  __shared__ double T[NB * PLD];
  __shared__ double yk[NB];
  const int lane = threadIdx.x;
  const double* D = diag + (size_t)k * NB * NB;
  for (int r = 0; r < NB; ++r) T[r * PLD + lane] = D[r * NB + lane];
  double zc = z[k * NB + lane];
  __syncthreads();
  double mine = 0.0;
  for (int c = NB - 1; c >= 0; --c) {
    const double num = __shfl(zc, c, 64);
    const double yc = num / T[c * PLD + c];
    if (lane == c) mine = yc;
    if (lane < c) zc -= T[c * PLD + lane] * yc;
  }
  if ((int)blockIdx.x == k) { y[k * NB + lane] = mine; return; }
  yk[lane] = mine;
  __syncthreads();
  const int j = blockIdx.x;
  const double* L = M + (size_t)k * NB * ld + (size_t)j * NB;
  double acc = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; ++m) acc += L[(size_t)m * ld + lane] * yk[m];
  z[j * NB + lane] -= acc;
}

void dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* g_diag) {
  const int nb = n_pad / NB, ld = n_pad;
  for (int k = 0; k < nb; ++k) {
    const int below = nb - 1 - k;                 // real row blocks below tile k
    hipLaunchKernelGGL(k_chol_panel, dim3(1 + below + 1), dim3(64), 0, st, M, ld, k, g_diag, fail);
    if (below > 0)
      hipLaunchKernelGGL(k_chol_update, dim3(below, below + 1), dim3(256), 0, st, M, ld, k);
  }
  double* z = M + (size_t)n_pad * ld;
  for (int k = nb - 1; k >= 0; --k)
    hipLaunchKernelGGL(k_chol_backsolve, dim3(k + 1), dim3(64), 0, st, M, ld, k, g_diag, z, y);
}

}  // namespace mavba
