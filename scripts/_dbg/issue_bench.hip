// What does one vector instruction of k_schur_rows' non-FP64 kinds cost on gfx950, alone and next to a wave that issues
// v_mfma_f64_16x16x4_f64 on the same SIMD? (round 6: the kernel's time is the SUM of its instructions' issue slots; this
// prices v_swap_b32 under an exec mask against the v_cndmask_b32 pairs it replaces, the DPP moves, integer adds, and the
// 4x4x4 matrix instruction.)  One work-group of 512 lanes = two waves per SIMD, roles by (SIMD, arrival slot) as in
// pipe_bench.hip; every wave stamps its own loop with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_dbg/issue_bench scripts/_dbg/issue_bench.hip && scripts/_dbg/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
enum Role { IDLE = 0, MAT = 1, INTV = 2, SWAP = 3, CND = 4, DPP = 5, F64 = 6, MAT4 = 7, LDSW = 8, SWZT = 9, SWZL = 10, MIXS = 11, MATP = 12, MATQ = 13, NROLE = 14 };
static const char* kRoleName[NROLE] = {"idle", "mfma16x16x4", "v_add_u32", "v_swap_b32", "v_cndmask_b32", "v_mov_dpp", "v_fma_f64", "mfma4x4x4", "ds_write_b64", "ds_swizzle(tp)", "ds_swizzle(lat)", "swz+fma mix", "mfma+s_nop13", "mfma+s_nop15"};
constexpr int NI = 4096;  // instructions of a loop (matrix roles: NI / 16)
struct Config { int role[4][2]; };

__device__ __forceinline__ long long tick() {
  long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

__global__ void __launch_bounds__(512, 1) k_issue(Config cfg, long long* __restrict__ out, double* __restrict__ sink, double seed, int iseed) {
  __shared__ int s_cnt[4];
  __shared__ double s_buf[512 * 2];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const int simd = (hwid >> 4) & 3;
  int slot = 0;
  if (lane == 0) slot = atomicAdd(&s_cnt[simd], 1);
  slot = __builtin_amdgcn_readfirstlane(slot);
  const int role = slot < 2 ? cfg.role[simd][slot] : IDLE;
  double x[16];
  int q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = seed + 1e-3 * lane + i; q[i] = iseed + lane * 17 + i; }
  const double y = 0.999999 + seed * 1e-9, z = 1e-7;
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, 0.0, 1.0, seed};
  double a1[4] = {seed, 1.0, 2.0, 3.0};
  __syncthreads();
  const long long t0 = tick();
  if (role == MAT) {
#pragma unroll 1
    for (int it = 0; it < NI / 16 / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y, acc[u & 3], 0, 0, 0);
    }
  } else if (role == MATP) {  // the next matrix instruction is not presented before the pipe is nearly free again
#pragma unroll 1
    for (int it = 0; it < NI / 16 / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y, acc[u & 3], 0, 0, 0); asm volatile("s_nop 13" ::: "memory"); }
    }
  } else if (role == MATQ) {
#pragma unroll 1
    for (int it = 0; it < NI / 16 / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y, acc[u & 3], 0, 0, 0); asm volatile("s_nop 15" ::: "memory"); }
    }
  } else if (role == MAT4) {
#pragma unroll 1
    for (int it = 0; it < NI / 16 / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a1[u & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[u], y, a1[u & 3], 0, 0, 0);
    }
  } else if (role == INTV) {
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) q[u & 15] = q[u & 15] + q[(u + 5) & 15];
    }
  } else if (role == F64) {
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) x[u & 15] = __builtin_fma(x[u & 15], y, z);
    }
  } else if (role == SWAP) {
    if (lane & 4) {  // (an exec mask like the reduce-scatter's: half the lanes)
#pragma unroll 1
      for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) asm volatile("v_swap_b32 %0, %1" : "+v"(q[u & 15]), "+v"(q[(u + 5) & 15]));
      }
    }
  } else if (role == CND) {
    const bool up = lane & 4;
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(q[u & 15]) : "v"(q[(u + 3) & 15]), "v"(q[(u + 5) & 15]), "s"(__builtin_amdgcn_ballot_w64(up)));
    }
  } else if (role == DPP) {
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) q[u & 15] = __builtin_amdgcn_update_dpp(0, q[(u + 5) & 15], 0x141, 0xf, 0xf, true);
    }
  } else if (role == SWZT) {  // 16 independent swizzles in flight
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) q[u & 15] = __builtin_amdgcn_ds_swizzle(q[u & 15], 0x3C1F);
    }
  } else if (role == SWZL) {  // one dependent chain
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) q[0] = __builtin_amdgcn_ds_swizzle(q[0], 0x3C1F);
    }
  } else if (role == MIXS) {  // the all-reduce's shape: 18 swizzles, then 9 FP64 adds that need them, 8 independent multiply-adds between
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int lv = 0; lv < 2; ++lv) {
        int lo[9], hi[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) { lo[e] = __builtin_amdgcn_ds_swizzle(__double2loint(x[e]), 0x3C1F); hi[e] = __builtin_amdgcn_ds_swizzle(__double2hiint(x[e]), 0x3C1F); }
#pragma unroll
        for (int e = 9; e < 16; ++e) x[e] = __builtin_fma(x[e], y, z);
#pragma unroll
        for (int e = 0; e < 9; ++e) x[e] += __hiloint2double(hi[e], lo[e]);
      }
    }
  } else if (role == LDSW) {
#pragma unroll 1
    for (int it = 0; it < NI / 64; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) s_buf[tid + 512 * (u & 1)] = x[u & 15];
    }
  }
  asm volatile("" ::"v"(x[0]), "v"(acc[0]), "v"(q[0]), "v"(a1[0]));
  const long long t1 = tick();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + q[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + a1[i];
  sink[(size_t)blockIdx.x * 512 + tid] = s + s_buf[tid];
  if (lane == 0) {
    long long* o = out + ((size_t)blockIdx.x * 8 + wv) * 4;
    o[0] = role; o[1] = simd * 2 + slot; o[2] = t1 - t0; o[3] = hwid;
  }
}

int main() {
  long long* d_out; double* d_sink;
  hipMalloc(&d_out, 8 * 4 * 8); hipMalloc(&d_sink, 512 * 8);
  auto run = [&](const char* name, int r0, int r1) {
    Config c; std::memset(&c, 0, sizeof(c));
    c.role[0][0] = r0; c.role[0][1] = r1;  // SIMD 0 only: the two roles share it
    long long best[2] = {1LL << 60, 1LL << 60};
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(d_out, 0, 8 * 4 * 8);
      hipLaunchKernelGGL(k_issue, dim3(1), dim3(512), 0, 0, c, d_out, d_sink, 1.0 + rep, rep);
      hipDeviceSynchronize();
      long long h[32]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w)
        for (int s = 0; s < 2; ++s)
          if (h[4 * w + 1] == s && h[4 * w] == (s == 0 ? r0 : r1) && h[4 * w] != IDLE) best[s] = std::min(best[s], h[4 * w + 2]);
    }
    auto n = [](int r) { return (r == MAT || r == MAT4 || r == MATP || r == MATQ) ? NI / 16 : NI; };
    std::printf("%-34s", name);
    if (r0 != IDLE) std::printf("  %-14s %8lld ticks = %6.2f / instruction", kRoleName[r0], best[0], (double)best[0] / n(r0));
    if (r1 != IDLE) std::printf("  |  %-14s %8lld ticks = %6.2f / instruction", kRoleName[r1], best[1], (double)best[1] / n(r1));
    std::printf("\n");
  };
  for (int r = 1; r < NROLE; ++r) run("alone", r, IDLE);
  for (int r = 2; r < NROLE; ++r) { if (r == MAT4) continue; run("next to mfma16x16x4, same SIMD", MAT, r); }
  for (int r = 2; r < NROLE; ++r) { if (r == F64 || r == MAT4) continue; run("next to v_fma_f64, same SIMD", F64, r); }
  {  // is the swizzle rate a SIMD's or the CU's? one swizzle wave on each of the four SIMDs
    auto run4 = [&](const char* name, int r) {
      Config c; std::memset(&c, 0, sizeof(c));
      for (int s = 0; s < 4; ++s) c.role[s][0] = r;
      hipMemset(d_out, 0, 8 * 4 * 8);
      hipLaunchKernelGGL(k_issue, dim3(1), dim3(512), 0, 0, c, d_out, d_sink, 1.0, 1);
      hipDeviceSynchronize();
      long long h[32]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
      std::printf("%-34s", name);
      for (int w = 0; w < 8; ++w) if (h[4 * w] == r) std::printf("  %lld", h[4 * w + 2]);
      std::printf(" ticks per wave\n");
    };
    run4("4 SIMDs x 1 ds_swizzle(tp) wave", SWZT);
    run4("4 SIMDs x 1 v_mov_dpp wave", DPP);
    run4("4 SIMDs x 1 mix wave", MIXS);
  }
  for (int m : {MATP, MATQ})
    for (int r : {INTV, CND, DPP, F64, SWZT}) run("next to padded mfma, same SIMD", m, r);
  run("two dpp waves", DPP, DPP);
  run("two swizzle waves", SWZT, SWZT);
  run("two mix waves", MIXS, MIXS);
  run("two fma waves", F64, F64);
  run("two integer waves", INTV, INTV);
  run("two cndmask waves", CND, CND);
  return 0;
}
