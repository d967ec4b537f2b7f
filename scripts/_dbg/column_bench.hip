// One chain column with a panel tile, standalone (round 5): the fused form of tile_potrf_inv_sys + Panel - panel solve
// P = A L^-T, diagonal update D -= P P^T, tile factor + inverse - with every wave's way points, against a host reference.
// (In the persistent launch the same column took 10.3 us = 24 700 ticks; the tile alone 13 000: where is the rest?)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I mavmap_amd/csrc -I include scripts/_dbg/column_bench.hip -o scripts/_dbg/column_bench
#include "../../mavmap_amd/csrc/dense_chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace mavba {
hipError_t device_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }
void device_free(void* p) { (void)hipFree(p); }
hipError_t copy_h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t st) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); }
void release_staged(hipStream_t) {}
namespace {
struct BenchPanel {
  static constexpr bool enabled = true;
  const double* As; double* Cs; bool sub;
  double* Pg;  // P goes here (plain stores: nobody polls)
  __device__ __forceinline__ void store_rows(int wv, int lane) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = lane + 64 * q;
      const int row = 16 * wv + (idx >> 5), c2 = (idx & 31) * 2;
      *reinterpret_cast<double2*>(Pg + (size_t)row * NB + c2) = *reinterpret_cast<const double2*>(Cs + row * GLD + c2);
    }
  }
  __device__ __forceinline__ void publish(int) const {}
  __device__ __forceinline__ int prefetch_poll() const { return 0; }
  __device__ __forceinline__ void prefetch(int, int) const {}
};
__global__ void __launch_bounds__(256) k_column(const double* D, const double* Asub, const double* Linv, double* Xout, double* Pout, long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Cs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ds[NB * GLD];
  __shared__ long long wmarks[32];
  const int tid = threadIdx.x;
  if (tid < 32) wmarks[tid] = 0;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    load_tile(D, NB, Ds, tid); load_tile(Asub, NB, As, tid); load_tile(Linv, NB, Bs, tid);
    __syncthreads();
    const long long t0 = clock64();
    auto mark = [&](int id) { if ((tid & 63) == 0) wmarks[(tid >> 6) * 8 + id] += clock64() - t0; };
    BenchPanel bp{As, Cs, true, Pout};
    tile_potrf_inv_sys<decltype(mark), 0, BenchPanel>(Ds, Bs, tid, mark, bp);
    __syncthreads();
    total += clock64() - t0;
  }
  __syncthreads();
  if (tid == 0) { cyc[0] = total / reps; for (int i = 0; i < 32; ++i) cyc[1 + i] = wmarks[i] / reps; }
  store_tile(Xout, NB, Bs, tid);
}
}}
int main() {
  using namespace mavba;
  const int n = 64;
  auto chol_inv = [&](const std::vector<double>& A, std::vector<double>& X) {
    std::vector<double> L(A);
    for (int j = 0; j < n; ++j) { for (int k = 0; k < j; ++k) for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
      const double d = std::sqrt(L[j * n + j]); for (int i = j; i < n; ++i) L[i * n + j] /= d; }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) L[i * n + j] = 0.0;
    X.assign(n * n, 0.0);
    for (int c = 0; c < n; ++c) for (int i = c; i < n; ++i) { double v = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) v -= L[i * n + k] * X[k * n + c]; X[i * n + c] = v / L[i * n + i]; }
  };
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  std::vector<double> G(n * n), D(n * n), Dp(n * n), Asub(n * n), Linv, P(n * n), Dn(n * n), X;
  for (auto& g : G) g = rnd();
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += G[i * n + k] * G[j * n + k]; Dp[i * n + j] = a + (i == j ? 4.0 : 0.0); }
  chol_inv(Dp, Linv);                                   // the "previous column's" inverse
  for (auto& g : G) g = rnd();
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += G[i * n + k] * G[j * n + k]; D[i * n + j] = a + (i == j ? 40.0 : 0.0); }
  for (auto& a : Asub) a = rnd();
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += Asub[i * n + k] * Linv[j * n + k]; P[i * n + j] = a; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = D[i * n + j]; for (int k = 0; k < n; ++k) a -= P[i * n + k] * P[j * n + k]; Dn[i * n + j] = a; }
  chol_inv(Dn, X);
  double *dD, *dA, *dL, *dX, *dP; long long* dc;
  hipMalloc(&dD, n * n * 8); hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dP, n * n * 8); hipMalloc(&dc, 8 * 40);
  hipMemcpy(dD, D.data(), n * n * 8, hipMemcpyHostToDevice); hipMemcpy(dA, Asub.data(), n * n * 8, hipMemcpyHostToDevice); hipMemcpy(dL, Linv.data(), n * n * 8, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(k_column, dim3(1), dim3(256), 0, 0, dD, dA, dL, dX, dP, dc, 20); hipDeviceSynchronize(); }
  std::vector<double> Xd(n * n), Pd(n * n); long long m[33];
  hipMemcpy(Xd.data(), dX, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(Pd.data(), dP, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(m, dc, 8 * 33, hipMemcpyDeviceToHost);
  double ex = 0, ep = 0, up = 0, mx = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { ep = std::fmax(ep, std::fabs(Pd[i * n + j] - P[i * n + j])); mx = std::fmax(mx, std::fabs(X[i * n + j]));
    if (j <= i) ex = std::fmax(ex, std::fabs(Xd[i * n + j] - X[i * n + j])); else up = std::fmax(up, std::fabs(Xd[i * n + j])); }
  printf("fused column: %lld ticks, inverse max err %.2e (scale %.2e), upper max %.2e, panel tile max err %.2e\n", m[0], ex, mx, up, ep);
  for (int w = 0; w < 4; ++w) { printf("  wave %d way points:", w); for (int i = 0; i < 8; ++i) printf(" %d:%lld", i, m[1 + 8 * w + i]); printf("\n"); }
  return 0;
}
