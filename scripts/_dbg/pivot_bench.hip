// Where do the ticks of ONE 16-pivot block (four 4-pivot steps, potrf_inv16_block4 of dense_chol.hip) go? (round 5)
// Round 4 left "the 4-pivot step takes ~740-875 ticks, the latency model says 430" unexplained. Hypothesis tested here: the
// wave is ISSUE-bound, not latency-bound - ~110 instructions per step at ~5 ticks each, and every v_mfma_f64_16x16x4_f64
// holds the SIMD's FP64 pipe for ~64 ticks whether or not anything depends on it (profiles/r04_pipe_bench_fp64.txt), so
// the two matrix instructions that only carry the running INVERSE sit on the critical wave's issue path.
// Variants (one wave, everything in registers, `REPS` blocks back to back):
//   0  the product's block (U, Xn, acc update, xacc update: 4 matrix instructions per step)
//   1  without the inverse (U, acc update: 2 per step)                                  - timing + factor check
//   2  without the inverse, carrying the transposed block below (Lt = X4 * Bt rows, Bt -= u^T Lt) and the next diagonal
//      block (Dn -= Lt^T Lt): 5 per step, no LDS round trip to reach the next diagonal block  - timing + check
//   3  the scalar part only (readlanes + 4x4 factor + operand build), matrix instructions replaced by moves
//   4  the four matrix instructions only (operands fixed)
//   5  variant 0 with the inverse's two instructions issued AFTER the factor's (order pinned)
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -I mavmap_amd/csrc -I include scripts/_dbg/pivot_bench.hip -o scripts/_dbg/pivot_bench
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
__device__ __forceinline__ long long tick() {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

struct Carry { d4 acc, xacc, xfin, bt, ltfin, dn; };

template <int B, int MODE>
__device__ __forceinline__ void blk4(Carry& c, int lane, const Pivot4Masks& mk) {
  const int li = lane & 15, lk = lane >> 4;
  constexpr int c0 = 4 * B;
  d4& acc = c.acc;
  double xa;
  if (MODE != 4) {
    const double d00 = readlane_d(acc[B], 0 * 16 + c0 + 0);
    const double y00 = __builtin_amdgcn_rsq(d00);
    int alo = __double2loint(acc[B]), ahi = __double2hiint(acc[B]);
    asm volatile("" : "+v"(alo), "+v"(ahi) : "v"(y00));
    auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(ahi, l), __builtin_amdgcn_readlane(alo, l)); };
    const double d10 = rl(1 * 16 + c0 + 0), d11 = rl(1 * 16 + c0 + 1);
    const double d20 = rl(2 * 16 + c0 + 0), d21 = rl(2 * 16 + c0 + 1), d22 = rl(2 * 16 + c0 + 2);
    const double d30 = rl(3 * 16 + c0 + 0), d31 = rl(3 * 16 + c0 + 1), d32 = rl(3 * 16 + c0 + 2), d33 = rl(3 * 16 + c0 + 3);
    const double r0 = rsqrt_finish(y00, __builtin_fma(-(d00 * y00), y00, 1.0));
    const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
    const double t11 = __builtin_fma(-l10, l10, d11);
    const double r1 = rsqrt_halley(t11);
    const double l21 = __builtin_fma(-l20, l10, d21) * r1, l31 = __builtin_fma(-l30, l10, d31) * r1;
    const double t22 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, d22));
    const double r2 = rsqrt_halley(t22);
    const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, d32)) * r2;
    const double t33 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, d33)));
    const double r3 = rsqrt_halley(t33);
    const double x10 = -(l10 * r0) * r1;
    const double x21 = -(l21 * r1) * r2;
    const double x32 = -(l32 * r2) * r3;
    const double x20 = -__builtin_fma(l21, x10, l20 * r0) * r2;
    const double x31 = -__builtin_fma(l32, x21, l31 * r1) * r3;
    const double x30 = -__builtin_fma(l32, x20, __builtin_fma(l31, x10, l30 * r0)) * r3;
    xa = mk.m[0] * r0;
    xa = __builtin_fma(mk.m[1], x10, xa); xa = __builtin_fma(mk.m[2], r1, xa);
    xa = __builtin_fma(mk.m[3], x20, xa); xa = __builtin_fma(mk.m[4], x21, xa); xa = __builtin_fma(mk.m[5], r2, xa);
    xa = __builtin_fma(mk.m[6], x30, xa); xa = __builtin_fma(mk.m[7], x31, xa); xa = __builtin_fma(mk.m[8], x32, xa);
    xa = __builtin_fma(mk.m[9], r3, xa);
  } else {
    xa = mk.m[0] + mk.m[2] + mk.m[5] + mk.m[9];  // (a fixed operand: the identity's 4x4 block)
  }
  const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
  const double ba = li >= c0 ? acc[B] : 0.0;
  if (MODE == 3) {  // scalar part only: the "results" are moves that keep the dependency chain alive
    acc[(B + 1) & 3] += xa * 1e-30;
    c.xfin[B] = xa;
    return;
  }
  if (MODE == 0 || MODE == 4) {
    const d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, ba, zero, 0, 0, 0);
    const d4 Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, c.xacc[B], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const double u = li >= c0 + lk ? U[0] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
    c.xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, Xn[0], c.xacc, 0, 0, 0);
    c.xfin[B] = Xn[0];
  } else if (MODE == 5) {
    const d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, ba, zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const double u = li >= c0 + lk ? U[0] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const d4 Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, c.xacc[B], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    c.xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, Xn[0], c.xacc, 0, 0, 0);
    c.xfin[B] = Xn[0];
  } else if (MODE == 1) {
    const d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, ba, zero, 0, 0, 0);
    const double u = li >= c0 + lk ? U[0] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
    c.xfin[B] = u;  // rows 4B..4B+3 of L^T
  } else if (MODE == 2) {
    const d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, ba, zero, 0, 0, 0);
    const d4 Lt = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, c.bt[B], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const double u = li >= c0 + lk ? U[0] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
    c.bt = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, Lt[0], c.bt, 0, 0, 0);
    c.dn = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lt[0], Lt[0], c.dn, 0, 0, 0);
    c.xfin[B] = u;
    c.ltfin[B] = Lt[0];
  }
}

template <int MODE>
__device__ __forceinline__ void block16(Carry& c, int lane, const Pivot4Masks& mk) {
  blk4<0, MODE>(c, lane, mk); blk4<1, MODE>(c, lane, mk); blk4<2, MODE>(c, lane, mk); blk4<3, MODE>(c, lane, mk);
}

// A: 32 x 32 SPD, row-major. out: [0] xfin (16x16 in the accumulator layout's natural order), [1] ltfin, [2] dn
template <int MODE>
__global__ void __launch_bounds__(64) k_pivot(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc, int reps) {
  const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const Pivot4Masks mk = pivot4_masks(lane);
  Carry c0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c0.acc[r] = A[(lk + 4 * r) * 32 + li];
    c0.xacc[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
    c0.xfin[r] = 0.0;
    c0.bt[r] = A[(16 + li) * 32 + lk + 4 * r];         // Bt[m][n] = A[16 + n][m]
    c0.ltfin[r] = 0.0;
    c0.dn[r] = A[(16 + lk + 4 * r) * 32 + 16 + li];
  }
  Carry c = c0;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    // (same input every time; the tiny dependence on the previous result keeps the compiler from hoisting anything)
    const double eps = c.xfin[3] * 1e-300;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c.acc[r] = c0.acc[r] + eps; c.xacc[r] = c0.xacc[r]; c.bt[r] = c0.bt[r]; c.dn[r] = c0.dn[r]; }
    const long long t0 = tick();
    block16<MODE>(c, lane, mk);
    asm volatile("" :: "v"(c.acc[3]), "v"(c.xfin[3]), "v"(c.xacc[3]), "v"(c.dn[3]), "v"(c.bt[3]));
    total += tick() - t0;
  }
  if (lane == 0) cyc[0] = total / reps;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    out[(lk + 4 * r) * 16 + li] = c.xfin[r];
    out[256 + (lk + 4 * r) * 16 + li] = c.ltfin[r];
    out[512 + (lk + 4 * r) * 16 + li] = c.dn[r];
  }
}

// Issue rates / latencies on ONE wave alone on its SIMD (64-lane work-group).
__global__ void __launch_bounds__(64) k_rates(double* out, long long* cyc, double seed) {
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0000001;
  long long t0; int n = 0;
  constexpr int R = 64;
  // 0: dependent fma
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i) x = __builtin_fma(x, y, 0.5);
  asm volatile("" :: "v"(x)); cyc[n++] = tick() - t0;
  // 1: 8 independent fma chains (issue rate)
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = x + j;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], y, 0.5);
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(a[j]));
  cyc[n++] = tick() - t0;
  // 2: readlane pairs, independent (issue rate of v_readlane_b32)
  int acc_i = 0;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i) { acc_i ^= __builtin_amdgcn_readlane(__double2loint(a[i & 7]), (i * 7) & 63); }
  asm volatile("" :: "s"(acc_i)); cyc[n++] = tick() - t0;
  // 3: dependent mfma chain through the accumulator
  d4 m0 = (d4){x, y, x, y}, m1 = m0, m2 = m0, m3 = m0;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i) m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
  asm volatile("" :: "v"(m0[0])); cyc[n++] = tick() - t0;
  // 4: four independent mfma accumulators (issue rate)
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R / 4; ++i) {
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, m1, 0, 0, 0);
    m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, m3, 0, 0, 0);
  }
  asm volatile("" :: "v"(m0[0]), "v"(m1[0]), "v"(m2[0]), "v"(m3[0])); cyc[n++] = tick() - t0;
  // 5: mfma result -> operand of the next mfma (A/B chain: what U -> acc update is)
  double op = x;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i) { const d4 z = __builtin_amdgcn_mfma_f64_16x16x4f64(op, y, (d4){0, 0, 0, 0}, 0, 0, 0); op = z[0]; }
  asm volatile("" :: "v"(op)); cyc[n++] = tick() - t0;
  // 6: one mfma followed by 16 independent fmas, repeated (does the vector work hide behind the matrix instruction?)
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R / 4; ++i) {
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], y, 0.5);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(a[j]));
  asm volatile("" :: "v"(m0[0])); cyc[n++] = tick() - t0;
  // 7: one mfma followed by 16 independent 32-bit integer VALU instructions (a different pipe?)
  int q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = lane + j;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R / 4; ++i) {
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = q[j] * 3 + 1;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(q[j]));
  asm volatile("" :: "v"(m0[0])); cyc[n++] = tick() - t0;
  // 8: 16 x (16 independent int VALU) alone
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R / 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = q[j] * 3 + 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(q[j]));
  cyc[n++] = tick() - t0;
  // 9: one mfma followed by 16 readlanes
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R / 4; ++i) {
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc_i ^= __builtin_amdgcn_readlane(q[j & 7], (i * 16 + j) & 63);
  }
  asm volatile("" :: "s"(acc_i)); asm volatile("" :: "v"(m0[0])); cyc[n++] = tick() - t0;
  // 10: v_rsq_f64 dependent
  x = fabs(x) + 1.0;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < R; ++i) x = __builtin_amdgcn_rsq(x);
  asm volatile("" :: "v"(x)); cyc[n++] = tick() - t0;
  // 11/12: calibration against the 100 MHz wall clock
  { const long long w0 = wall_clock64(); t0 = tick();
#pragma unroll 1
    for (int i = 0; i < 40000; ++i) x = __builtin_fma(x, y, 0.5);
    asm volatile("" :: "v"(x)); cyc[n++] = tick() - t0; cyc[n++] = wall_clock64() - w0; }
  out[lane] = x + m0[0] + m1[1] + m2[2] + m3[3] + a[0] + a[7] + op + q[0] + acc_i;
}
}  // namespace
}  // namespace mavba

int main() {
  using namespace mavba;
  const int n = 32;
  std::vector<double> G(n * n), A(n * n, 0.0);
  unsigned s = 4321;
  for (auto& g : G) { s = s * 1664525u + 1013904223u; g = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int k = 0; k < n; ++k) a += G[i * n + k] * G[j * n + k]; A[i * n + j] = a + (i == j ? 2.0 : 0.0); }
  // host reference: L of the 32 x 32, X0 = L00^-1
  std::vector<double> L(A);
  for (int j = 0; j < n; ++j) { for (int k = 0; k < j; ++k) for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
    const double d = std::sqrt(L[j * n + j]); for (int i = j; i < n; ++i) L[i * n + j] /= d; }
  std::vector<double> X(16 * 16, 0.0);
  for (int c = 0; c < 16; ++c) for (int i = c; i < 16; ++i) { double v = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) v -= L[i * n + k] * X[k * 16 + c]; X[i * 16 + c] = v / L[i * n + i]; }
  std::vector<double> Dn(16 * 16);  // A11 - L10 L10^T
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = A[(16 + i) * n + 16 + j]; for (int k = 0; k < 16; ++k) v -= L[(16 + i) * n + k] * L[(16 + j) * n + k]; Dn[i * 16 + j] = v; }
  double *dA, *dO; long long* dc;
  (void)hipMalloc(&dA, n * n * 8); (void)hipMalloc(&dO, 768 * 8); (void)hipMalloc(&dc, 32 * 8);
  (void)hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  const char* what[] = {"product block (4 matrix instructions / step)", "no inverse (2 / step)", "no inverse + block below + next diagonal (5 / step)",
                        "scalar part only", "matrix instructions only", "product block, inverse's instructions last"};
  for (int v = 0; v < 6; ++v) {
    for (int pass = 0; pass < 2; ++pass) {
      switch (v) {
        case 0: hipLaunchKernelGGL(k_pivot<0>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
        case 1: hipLaunchKernelGGL(k_pivot<1>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
        case 2: hipLaunchKernelGGL(k_pivot<2>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
        case 3: hipLaunchKernelGGL(k_pivot<3>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
        case 4: hipLaunchKernelGGL(k_pivot<4>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
        case 5: hipLaunchKernelGGL(k_pivot<5>, dim3(1), dim3(64), 0, 0, dA, dO, dc, reps); break;
      }
      (void)hipDeviceSynchronize();
    }
    std::vector<double> O(768); long long c;
    (void)hipMemcpy(O.data(), dO, 768 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    double e_inv = 0, e_lt = 0, e_dn = 0, e_l = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      if (j <= i) e_inv = std::fmax(e_inv, std::fabs(O[i * 16 + j] - X[i * 16 + j]));
      if (j >= i) e_l = std::fmax(e_l, std::fabs(O[i * 16 + j] - L[j * n + i]));             // xfin = L^T rows (variants 1, 2)
      e_lt = std::fmax(e_lt, std::fabs(O[256 + i * 16 + j] - L[(16 + j) * n + i]));            // ltfin[m][n] = L10[n][m]
      e_dn = std::fmax(e_dn, std::fabs(O[512 + i * 16 + j] - Dn[i * 16 + j]));
    }
    printf("variant %d  %-58s %6lld ticks / 16 pivots (%5.0f / step)", v, what[v], c, c / 4.0);
    if (v == 0 || v == 5) printf("   inverse err %.1e", e_inv);
    if (v == 1 || v == 2) printf("   L^T err %.1e", e_l);
    if (v == 2) printf("  L10^T err %.1e  next-diagonal err %.1e", e_lt, e_dn);
    printf("\n");
  }
  {
    double* o; (void)hipMalloc(&o, 64 * 8);
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(64), 0, 0, o, dc, 1.5); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(64), 0, 0, o, dc, 1.5); (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, dc, 16 * 8, hipMemcpyDeviceToHost);
    printf("one wave alone on its SIMD (ticks of s_memtime):\n");
    printf("  dependent v_fma_f64                      %6.1f / instruction\n", h[0] / 64.0);
    printf("  8 independent v_fma_f64 chains           %6.2f / instruction\n", h[1] / 512.0);
    printf("  independent v_readlane_b32               %6.2f / instruction\n", h[2] / 64.0);
    printf("  dependent matrix instruction (C chain)   %6.1f / instruction\n", h[3] / 64.0);
    printf("  4 independent matrix accumulators        %6.1f / instruction\n", h[4] / 64.0);
    printf("  matrix result -> next operand            %6.1f / instruction\n", h[5] / 64.0);
    printf("  1 matrix + 16 independent v_fma_f64      %6.1f / group  (sum would be %.1f)\n", h[6] / 16.0, h[4] / 64.0 + 16 * h[1] / 512.0);
    printf("  1 matrix + 16 independent int VALU       %6.1f / group  (16 int VALU alone: %.1f)\n", h[7] / 16.0, h[8] / 16.0);
    printf("  1 matrix + 16 v_readlane_b32             %6.1f / group\n", h[9] / 16.0);
    printf("  dependent v_rsq_f64                      %6.1f / instruction\n", h[10] / 64.0);
    printf("  calibration: %lld ticks in %lld x 10 ns = %.0f ticks / us; dependent v_fma_f64 %.2f ticks\n", h[11], h[12], h[11] / (h[12] * 0.01), h[11] / 40000.0);
  }
  return 0;
}
