// dev_reduce.h - wave-level reductions shared by the kernel files (gfx950, wave64).
#ifndef MAVBA_DEV_REDUCE_H_
#define MAVBA_DEV_REDUCE_H_
#include <hip/hip_runtime.h>

namespace mavba {

// Sum over the 64 lanes, result in EVERY lane: DPP row rotations (8, 4, 2, 1) give each 16-lane row
// its total, four v_readlane pairs combine the rows in a fixed order. No LDS traffic.
__device__ __forceinline__ double wave_sum(double v) {
#define MAVBA_ROR_ADD(N)                                                                                   \
  {                                                                                                        \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 | (N), 0xf, 0xf, false);        \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 | (N), 0xf, 0xf, false);        \
    v += __hiloint2double(hi, lo);                                                                         \
  }
  MAVBA_ROR_ADD(8) MAVBA_ROR_ADD(4) MAVBA_ROR_ADD(2) MAVBA_ROR_ADD(1)
#undef MAVBA_ROR_ADD
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// Work-group barrier that orders LDS traffic only. __syncthreads() is a work-group fence + barrier, and the fence also
// waits for every GLOBAL store the wave has in flight (s_waitcnt vmcnt(0)): thousands of cycles when the phase before the
// barrier streamed records to HBM (k_point_front: 6.6 k of 34 k cycles per tile). Use where the barrier only hands LDS
// data (or nothing) from one phase to the next and no lane reads global memory another lane of the group wrote.
// 1 / sqrt(d) to full FP64 precision without a divide or a square root: v_rsq_f64 + one third-order correction
// (max relative error 1.4e-16 over 1e6 random inputs, scripts/_dbg/tile_bench.hip).
__device__ __forceinline__ double dev_rsqrt(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double h = __builtin_fma(-(d * y), y, 1.0);
  return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
}
// chol3_inv (ba_math.h) for the owner lanes of the front-end kernels: the same factor and inverse from three reciprocal
// square roots - the library version's 3 sqrt + 6 divisions are ~300 dependent FP64 instructions, 2.8 k cycles during
// which 31 of 32 waves' lanes wait. Same storage order; false if C is not positive definite.
__device__ __forceinline__ bool chol3_inv_fast(const double* C, double* Gi) {
  const double r0 = dev_rsqrt(C[0]);
  const double l10 = C[1] * r0, l20 = C[2] * r0;
  const double d1 = __builtin_fma(-l10, l10, C[3]);
  const double r1 = dev_rsqrt(d1);
  const double l21 = __builtin_fma(-l20, l10, C[4]) * r1;
  const double d2 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, C[5]));
  const double r2 = dev_rsqrt(d2);
  Gi[0] = r0;
  Gi[1] = -(l10 * r0) * r1;
  Gi[2] = r1;
  Gi[4] = -(l21 * r1) * r2;
  Gi[3] = -__builtin_fma(l21, Gi[1], l20 * r0) * r2;
  Gi[5] = r2;
  return (C[0] > 0.0) && (d1 > 0.0) && (d2 > 0.0);
}

// Block-wide sum for 256-thread blocks; `scratch` = 4 doubles of LDS. Result in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wv] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}
__device__ __forceinline__ double block_max_256(double v, double* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wv] = v;
  __syncthreads();
  return fmax(fmax(scratch[0], scratch[1]), fmax(scratch[2], scratch[3]));
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace mavba
#endif
