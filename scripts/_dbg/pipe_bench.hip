// Do FP64 matrix (v_mfma_f64_16x16x4_f64) and FP64 vector (v_fma_f64) instructions of DIFFERENT waves overlap on gfx950?
// One work-group of 512 lanes = two waves per SIMD. Every wave finds its SIMD (HW_ID) and its arrival slot on it and
// takes the role a configuration table gives that (SIMD, slot): idle, a matrix-only loop (NM independent-accumulator
// instructions) or a vector-only loop (NV fused multiply-adds on 16 independent chains). Each wave stamps its own loop
// with s_memtime. "sum" = both waves of a pair take about T_matrix + T_vector, "max" = each keeps its solo time.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_dbg/pipe_bench scripts/_dbg/pipe_bench.hip && scripts/_dbg/pipe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
enum Role { IDLE = 0, MAT = 1, VEC = 2, MIX = 3 };
constexpr int NM = 512;        // matrix instructions of a MAT wave
constexpr int NV = 8192;       // vector instructions of a VEC wave  (NM * 64 cycles == NV * 4 cycles at the documented rates)
struct Config { int role[4][2]; };

__device__ __forceinline__ long long tick() {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

__global__ void __launch_bounds__(512, 1) k_pipe(Config cfg, long long* __restrict__ out, double* __restrict__ sink, double seed) {
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID, all 32 bits
  const int simd = (hwid >> 4) & 3;
  int slot = 0;
  if (lane == 0) slot = atomicAdd(&s_cnt[simd], 1);
  slot = __builtin_amdgcn_readfirstlane(slot);
  const int role = slot < 2 ? cfg.role[simd][slot] : IDLE;
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = seed + 1e-3 * lane + i;
  const double y = 0.999999 + seed * 1e-9, z = 1e-7;
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, 0.0, 1.0, seed};
  __syncthreads();
  const long long t0 = tick();
  if (role == MAT) {
#pragma unroll 1
    for (int it = 0; it < NM / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y, acc[u & 3], 0, 0, 0);
    }
  } else if (role == VEC) {
#pragma unroll 1
    for (int it = 0; it < NV / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) x[u & 15] = __builtin_fma(x[u & 15], y, z);
    }
  } else if (role == MIX) {  // one wave: 1 matrix instruction, then 16 independent vector instructions (= 64 cycles each at the documented rates)
#pragma unroll 1
    for (int it = 0; it < NM / 4; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc[u], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) x[v] = __builtin_fma(x[v], y, z);
        __builtin_amdgcn_sched_barrier(0);  // (keep the 1 : 16 interleave)
      }
    }
  }
  asm volatile("" ::"v"(x[0]), "v"(acc[0]));
  const long long t1 = tick();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(size_t)blockIdx.x * 512 + tid] = s;
  if (lane == 0) {
    long long* o = out + ((size_t)blockIdx.x * 8 + wv) * 4;
    o[0] = role; o[1] = simd * 2 + slot; o[2] = t1 - t0; o[3] = hwid;
  }
}

int main() {
  long long* d_out; double* d_sink;
  const int max_wg = 256;
  (void)hipMalloc(&d_out, (size_t)max_wg * 8 * 4 * 8);
  (void)hipMalloc(&d_sink, (size_t)max_wg * 512 * 8);
  struct Named { const char* name; Config c; };
  auto all = [](int a, int b) { Config c; for (int s = 0; s < 4; ++s) { c.role[s][0] = a; c.role[s][1] = b; } return c; };
  auto one = [](int a, int b, int a1 = IDLE) { Config c; memset(&c, 0, sizeof c); c.role[0][0] = a; c.role[0][1] = b; c.role[1][0] = a1; return c; };
  std::vector<Named> cfgs = {
      {"matrix alone (one wave on the CU)", one(MAT, IDLE)},
      {"vector alone (one wave on the CU)", one(VEC, IDLE)},
      {"matrix + vector, SAME SIMD", one(MAT, VEC)},
      {"matrix + vector, DIFFERENT SIMDs", one(MAT, IDLE, VEC)},
      {"matrix + matrix, same SIMD", one(MAT, MAT)},
      {"vector + vector, same SIMD", one(VEC, VEC)},
      {"one wave interleaving 1 matrix : 16 vector", one(MIX, IDLE)},
      {"every SIMD: matrix alone", all(MAT, IDLE)},
      {"every SIMD: vector alone", all(VEC, IDLE)},
      {"every SIMD: matrix + vector", all(MAT, VEC)},
      {"every SIMD: matrix + matrix", all(MAT, MAT)},
      {"every SIMD: vector + vector", all(VEC, VEC)},
      {"every SIMD: interleaving wave alone", all(MIX, IDLE)},
      {"every SIMD: two interleaving waves", all(MIX, MIX)},
  };
  printf("NM = %d matrix instructions (v_mfma_f64_16x16x4_f64) per MAT wave, NV = %d v_fma_f64 per VEC wave; an interleaving wave issues NM + 16 NM\n", NM, NV);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {1, 256}) {
    printf("\n== grid of %d work-group(s) of 512 lanes (one per CU) ==\n", grid);
    for (auto& nc : cfgs) {
      for (int rep = 0; rep < 2; ++rep) {  // second launch is reported
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_pipe, dim3(grid), dim3(512), 0, 0, nc.c, d_out, d_sink, 1.5);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
      }
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h((size_t)grid * 8 * 4);
      (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      // averages over the work-groups, per (SIMD, slot) of work-group 0's layout; roles are placement-determined, so average by role
      double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}; int cnt[4] = {0, 0, 0, 0};
      for (int g = 0; g < grid; ++g)
        for (int w = 0; w < 8; ++w) {
          const long long* o = &h[((size_t)g * 8 + w) * 4];
          sum[o[0]] += (double)o[2]; cnt[o[0]]++; if ((double)o[2] > mx[o[0]]) mx[o[0]] = (double)o[2];
        }
      printf("%-46s", nc.name);
      if (cnt[MAT]) printf(" | MAT wave %8.0f cycles (%.1f / instr)", sum[MAT] / cnt[MAT], sum[MAT] / cnt[MAT] / NM);
      if (cnt[VEC]) printf(" | VEC wave %8.0f cycles (%.2f / instr)", sum[VEC] / cnt[VEC], sum[VEC] / cnt[VEC] / NV);
      if (cnt[MIX]) printf(" | MIX wave %8.0f cycles (%.1f per 1 + 16)", sum[MIX] / cnt[MIX], sum[MIX] / cnt[MIX] / NM);
      printf(" | kernel %.1f us\n", ms * 1e3);
      if (grid == 1 && &nc == &cfgs[2]) {
        printf("   placement of work-group 0 (wave: role simd.slot hw_id):");
        for (int w = 0; w < 8; ++w) printf("  w%d: %lld %lld.%lld %08llx", w, h[w * 4], h[w * 4 + 1] / 2, h[w * 4 + 1] % 2, (unsigned long long)h[w * 4 + 3]);
        printf("\n");
      }
    }
  }
  return 0;
}
