// Cycle counts + correctness of the 64x64 tile factor/inverse variants (debug harness, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -I mavmap_amd/csrc -I include scripts/_dbg/tile_bench.hip -o scripts/_dbg/tile_bench
#include "../../mavmap_amd/csrc/dense_chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace mavba {
hipError_t device_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }  // (the product's pool lives in host_util.hip)
void device_free(void* p) { (void)hipFree(p); }
hipError_t copy_h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t st) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); }
void release_staged(hipStream_t) {}
namespace {
// (the first blocked variant, kept here for comparison: 18 barriers, per-wave scratch)
// Factorise the SPD tile in T (LDS, pitch GLD) and invert the factor, blocked 4 x 4 in 16x16:
//   T  <- L below the diagonal blocks (the diagonal blocks themselves are consumed),
//   Ti <- L^-1 (lower; Ti must come in zeroed).
// Per 16-column block: the diagonal block is factored + inverted in registers by 16 lanes
// (potrf_inv16, one wave); the panel below it, the trailing update inside the tile and the assembly of
// the off-diagonal blocks of L^-1 are 16x16x16 FP64-MFMA products spread over the 4 waves.
// All 256 threads must call it (uniform barriers). scr: 4 * 16 * 18 doubles of LDS.
// Returns false (in thread 0) if a pivot is not positive.
constexpr int MLD = 18;
__device__ __forceinline__ bool tile_potrf_inv(double* T, double* Ti, double* scr, int tid) {
  const int wv = tid >> 6, lane = tid & 63;
  bool ok = true;
  for (int cb = 0; cb < 4; ++cb) {
    double* D = T + (16 * cb) * GLD + 16 * cb;
    double* Di = Ti + (16 * cb) * GLD + 16 * cb;
    if (wv == 0) {
      d4 acc = load_d16(D, GLD, lane), xacc;
      ok = potrf_inv16(acc, xacc, lane) && ok;
      store_d16(Di, GLD, xacc, lane);  // the diagonal block of L itself is not needed again
    }
    __syncthreads();
    const int nt = 3 - cb;  // 16-row blocks below the diagonal block
    if (wv < nt) {          // panel: P = T[rt, cb] * Dinv^T
      double* P = T + (16 * (cb + 1 + wv)) * GLD + 16 * cb;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      acc = gemm16(P, GLD, Di, GLD, true, 1.0, acc, lane);
      store_d16(P, GLD, acc, lane);
    }
    __syncthreads();
    // trailing update inside the tile: T[i, j] -= P_i P_j^T for cb < j <= i <= 3
    const int npairs = nt * (nt + 1) / 2;
    for (int p = wv; p < npairs; p += 4) {
      int i = 0, q = p;
      while (q > i) { q -= i + 1; ++i; }   // p -> (i, q) with q <= i
      const int bi = cb + 1 + i, bj = cb + 1 + q;
      double* C = T + (16 * bi) * GLD + 16 * bj;
      d4 acc = load_d16(C, GLD, lane);
      acc = gemm16(T + (16 * bi) * GLD + 16 * cb, GLD, T + (16 * bj) * GLD + 16 * cb, GLD, true, -1.0, acc, lane);
      store_d16(C, GLD, acc, lane);
    }
    __syncthreads();
  }
  // off-diagonal blocks of X = L^-1:  X_ij = -Dinv_i * sum_{k=j}^{i-1} L_ik X_kj,  by distance d = i - j
  double* Ms = scr + wv * 16 * MLD;
  for (int d = 1; d < 4; ++d) {
    const bool mine = wv < 4 - d;
    const int i = d + wv, j = wv;
    if (mine) {
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; ++k)
        acc = gemm16(T + (16 * i) * GLD + 16 * k, GLD, Ti + (16 * k) * GLD + 16 * j, GLD, false, 1.0, acc, lane);
      store_d16(Ms, MLD, acc, lane);
    }
    __syncthreads();
    if (mine) {
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      acc = gemm16(Ti + (16 * i) * GLD + 16 * i, GLD, Ms, MLD, false, -1.0, acc, lane);
      store_d16(Ti + (16 * i) * GLD + 16 * j, GLD, acc, lane);
    }
    __syncthreads();
  }
  // thread 0 is in wave 0 lanes < 16: it carries the pivot verdict
  return ok;
}


__global__ void __launch_bounds__(256) k_bench(const double* A, double* Xout, long long* cyc, int variant, int reps) {
  __shared__ __attribute__((aligned(16))) double T[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ti[NB * GLD];
  __shared__ __attribute__((aligned(16))) double rd[4 * 16 * MLD];
  const int tid = threadIdx.x;
  long long total = 0;
  long long marks[20];
  __shared__ long long wmarks[32];
  if (tid < 32) wmarks[tid] = 0;
  for (int i = 0; i < 20; ++i) marks[i] = 0;
  for (int rep = 0; rep < reps; ++rep) {
    load_tile(A, NB, T, tid);
    for (int i = tid; i < NB * GLD; i += 256) Ti[i] = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    if (variant == 0) tile_potrf_inv(T, Ti, rd, tid);
    else if (variant == 1) tile_potrf_inv_la(T, Ti, tid);
    else if (variant == 3) {  // 4 x potrf_inv16_b4 alone (timing only)
      if (tid < 64) for (int cb = 0; cb < 4; ++cb) {
        d4 a = load_d16(T + 16 * cb * GLD + 16 * cb, GLD, tid), x;
        potrf_inv16_b4(a, x, tid);
        store_d16(Ti + 16 * cb * GLD + 16 * cb, GLD, x, tid);
      }
    } else if (variant == 4) {  // look-ahead variant with wave 0's way points
      auto mark = [&](int id) { if (tid == 0) marks[id] += clock64() - t0; };
      tile_potrf_inv_la(T, Ti, tid, NoPhaseHook(), mark);
    }

    else if (variant == 2) {  // 4 x potrf_inv16 alone (timing only)
      if (tid < 64) for (int cb = 0; cb < 4; ++cb) {
        d4 a = load_d16(T + 16 * cb * GLD + 16 * cb, GLD, tid), x;
        potrf_inv16(a, x, tid);
        store_d16(Ti + 16 * cb * GLD + 16 * cb, GLD, x, tid);
      }
    }
    __syncthreads();
    total += clock64() - t0;
  }
  __syncthreads();
  if (tid == 0) { cyc[0] = total / reps; for (int i = 0; i < 20; ++i) cyc[1 + i] = marks[i] / reps; if (variant == 8 || variant == 9) for (int i = 0; i < 32; ++i) cyc[1 + i] = wmarks[i] / reps; }
  store_tile(Xout, NB, Ti, tid);
}
// the systolic variants in a kernel of their own (register allocation as in the product's callers)
template <int VAR>
__global__ void __launch_bounds__(256) k_bench_sys(const double* A, double* Xout, long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double T[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ti[NB * GLD];
  __shared__ long long wmarks[32];
  const int tid = threadIdx.x;
  if (tid < 32) wmarks[tid] = 0;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    load_tile(A, NB, T, tid);
    for (int i = tid; i < NB * GLD; i += 256) Ti[i] = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    auto mark = [&](int id) { if ((tid & 63) == 0) wmarks[(tid >> 6) * 8 + id] += clock64() - t0; };
    if (VAR == 6) tile_potrf_inv_sys(T, Ti, tid);
    else if (VAR == 8) tile_potrf_inv_sys(T, Ti, tid, mark);
    else tile_potrf_inv_sys<decltype(mark), 1>(T, Ti, tid, mark);
    __syncthreads();
    total += clock64() - t0;
  }
  __syncthreads();
  if (tid == 0) { cyc[0] = total / reps; for (int i = 0; i < 32; ++i) cyc[1 + i] = wmarks[i] / reps; }
  store_tile(Xout, NB, Ti, tid);
}
__global__ void k_rsq(const double* d, double* out_nr, double* out_h, double* out_seed, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out_nr[i] = rsqrt_nr(d[i]); out_h[i] = rsqrt_halley(d[i]); out_seed[i] = __builtin_amdgcn_rsq(d[i]); }
}
}}
int main() {
  using namespace mavba;
  const int n = 64;
  std::vector<double> G(n * n), A(n * n, 0.0);
  unsigned s = 12345;
  for (auto& g : G) { s = s * 1664525u + 1013904223u; g = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += G[i * n + k] * G[j * n + k]; A[i * n + j] = a + (i == j ? 4.0 : 0.0); }
  // host reference: L, then X = L^-1
  std::vector<double> L(A), X(n * n, 0.0);
  for (int j = 0; j < n; ++j) { for (int k = 0; k < j; ++k) for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
    const double d = std::sqrt(L[j * n + j]); for (int i = j; i < n; ++i) L[i * n + j] /= d; }
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) L[i * n + j] = 0.0;
  for (int c = 0; c < n; ++c) for (int i = c; i < n; ++i) { double v = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) v -= L[i * n + k] * X[k * n + c]; X[i * n + c] = v / L[i * n + i]; }
  double *dA, *dX; long long* dc;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dc, 8 * 40);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  const int order[10] = {0, 1, 2, 3, 4, 6, 8, 9, 5, 7};  // (5 and 7: variants 1 and 6 on a tile whose last 32 columns are padding)
  for (int oi = 0; oi < 10; ++oi) {
    const int v = order[oi];
    if (v == 5) {  // the last 32 columns are padding (identity): the look-ahead variant skips their pivots
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (i >= 32 || j >= 32) A[i * n + j] = i == j ? 1.0 : 0.0;
      hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
      L = A; X.assign(n * n, 0.0);
      for (int j = 0; j < n; ++j) { for (int k = 0; k < j; ++k) for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
        const double d = std::sqrt(L[j * n + j]); for (int i = j; i < n; ++i) L[i * n + j] /= d; }
      for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) L[i * n + j] = 0.0;
      for (int c = 0; c < n; ++c) for (int i = c; i < n; ++i) { double vv = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) vv -= L[i * n + k] * X[k * n + c]; X[i * n + c] = vv / L[i * n + i]; }
    }
    const int kv = v == 5 ? 1 : (v == 7 ? 6 : v);
    if (kv == 6) hipLaunchKernelGGL(k_bench_sys<6>, dim3(1), dim3(256), 0, 0, dA, dX, dc, 20);
    else if (kv == 8) hipLaunchKernelGGL(k_bench_sys<8>, dim3(1), dim3(256), 0, 0, dA, dX, dc, 20);
    else if (kv == 9) hipLaunchKernelGGL(k_bench_sys<9>, dim3(1), dim3(256), 0, 0, dA, dX, dc, 20);
    else hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, dA, dX, dc, kv, 20);
    hipDeviceSynchronize();
    std::vector<double> Xd(n * n); long long c;
    hipMemcpy(Xd.data(), dX, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double err = 0, mx = 0, up = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { if ((v == 2 || v == 3) && (i / 16 != j / 16)) continue; mx = std::fmax(mx, std::fabs(X[i * n + j])); if (j <= i) err = std::fmax(err, std::fabs(Xd[i * n + j] - X[i * n + j])); else up = std::fmax(up, std::fabs(Xd[i * n + j])); }
    printf("variant %d: %lld cycles/tile, inverse max err %.2e (scale %.2e), upper max %.2e\n", v, c, err, mx, up);
    if (v == 8 || v == 9) { long long m[33]; hipMemcpy(m, dc, 8 * 33, hipMemcpyDeviceToHost); for (int w = 0; w < 4; ++w) { printf("  wave %d way points:", w); for (int i = 0; i < 6; ++i) printf(" %d:%lld", i, m[1 + 8 * w + i]); printf("\n"); } }
    if (v == 4) { long long m[21]; hipMemcpy(m, dc, 8 * 21, hipMemcpyDeviceToHost); printf("  way points of wave 0 (cycles from start):"); for (int i = 1; i <= 16; ++i) printf(" %d:%lld", i, m[1 + i]); printf("\n"); }
  }
  {
    const int m = 1 << 20; std::vector<double> d(m), o1(m), o2(m), o3(m);
    for (int i = 0; i < m; ++i) { s = s * 1664525u + 1013904223u; const double f = 1.0 + (s >> 8) / 16777216.0; s = s * 1664525u + 1013904223u; d[i] = std::ldexp(f, (int)((s >> 8) % 600) - 300); }
    double *dd, *d1, *d2, *d3; hipMalloc(&dd, m * 8); hipMalloc(&d1, m * 8); hipMalloc(&d2, m * 8); hipMalloc(&d3, m * 8);
    hipMemcpy(dd, d.data(), m * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rsq, dim3(m / 256), dim3(256), 0, 0, dd, d1, d2, d3, m); hipDeviceSynchronize();
    hipMemcpy(o1.data(), d1, m * 8, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, m * 8, hipMemcpyDeviceToHost); hipMemcpy(o3.data(), d3, m * 8, hipMemcpyDeviceToHost);
    long double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < m; ++i) { const long double r = 1.0L / sqrtl((long double)d[i]);
      e1 = fmaxl(e1, fabsl(o1[i] - r) / r); e2 = fmaxl(e2, fabsl(o2[i] - r) / r); e3 = fmaxl(e3, fabsl(o3[i] - r) / r); }
    printf("rsqrt max rel err: 2xNewton %.3Le  Halley %.3Le  seed %.3Le (eps %.3e)\n", e1, e2, e3, 2.22e-16);
  }
  return 0;
}
