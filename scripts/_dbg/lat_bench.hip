// Dependent-issue latencies (cycles) of the instructions on the tile factorisation's critical path.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 64
__device__ __forceinline__ long long tick() {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__global__ void k_lat(double* out, long long* cyc, double seed) {
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0000001;
  long long t0, t1; int n = 0;
  // 1: dependent v_fma_f64
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(x, y, 0.5);
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 2: dependent v_mul_f64
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) x = x * y;
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 3: dependent v_rsq_f64
  x = fabs(x) + 1.0;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rsq(x);
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 4: dependent mfma f64 16x16x4 (C -> D chain)
  d4 acc = (d4){x, y, x, y};
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 5: mfma -> valu (read acc) -> mfma operand chain
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); x = acc[0] * y; }
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 6: readlane -> valu chain
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5); x = x * __hiloint2double(hi, lo); }
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 7: independent fma throughput (4 chains)
  double a = x, b = y + 1, c = x + 2, d = y + 3;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) { a = __builtin_fma(a, y, 0.5); b = __builtin_fma(b, y, 0.5); c = __builtin_fma(c, y, 0.5); d = __builtin_fma(d, y, 0.5); }
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 8: independent mfma throughput (2 accumulators)
  d4 acc2 = acc;
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, acc2, 0, 0, 0); }
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 9: LDS write -> read round trip (same wave)
  __shared__ double sh[128];
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) { sh[lane] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = sh[lane ^ 1] + 1.0; }
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 10: s_barrier cost with 4 waves
  t0 = tick();
#pragma unroll
  for (int i = 0; i < REP; ++i) __syncthreads();
  asm volatile("" :: "v"(x)); t1 = tick(); cyc[n++] = (t1 - t0);
  // 11: clock64 overhead
  t0 = tick(); t1 = tick(); cyc[n++] = (t1 - t0) * REP;
  // 12: calibration: s_memtime ticks per s_memrealtime tick (100 MHz)
  { long long w0 = wall_clock64(); t0 = tick();
#pragma unroll 1
    for (int i = 0; i < 20000; ++i) x = __builtin_fma(x, y, 0.5);
    asm volatile("" :: "v"(x)); t1 = tick(); long long w1 = wall_clock64();
    cyc[n++] = (t1 - t0); cyc[n++] = (w1 - w0); }
  out[lane] = x + acc[0] + acc2[1] + a + b + c + d;
}
int main() {
  double* o; long long* c; (void)hipMalloc(&o, 256 * 8); (void)hipMalloc(&c, 16 * 8);
  hipLaunchKernelGGL(k_lat, dim3(1), dim3(256), 0, 0, o, c, 1.5); (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k_lat, dim3(1), dim3(256), 0, 0, o, c, 1.5); (void)hipDeviceSynchronize();
  long long h[16]; (void)hipMemcpy(h, c, 16 * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"dep v_fma_f64", "dep v_mul_f64", "dep v_rsq_f64", "dep mfma_f64_16x16x4 (C chain)", "mfma -> valu -> mfma", "readlane x2 -> valu", "4 indep fma (per 4)", "2 indep mfma (per 2)", "LDS write->read (wave)", "s_barrier (4 waves)", "clock64 pair"};
  for (int i = 0; i < 11; ++i) printf("%-34s %.1f cycles\n", names[i], (double)h[i] / REP);
  printf("calibration: %lld memtime ticks in %lld x 10 ns -> %.1f ticks/us; dependent fma = %.2f ticks\n", h[11], h[12], (double)h[11] / (h[12] * 0.01), (double)h[11] / 20000);
  return 0;
}
