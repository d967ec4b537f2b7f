// dense_chol.hip — structured SPD solve of the reduced camera system on gfx950.
//
// Replaces what the reference gets from Ceres' SPARSE_SCHUR back end (CHOLMOD
// factorisation of the reduced camera matrix, reference
// src/base3d/bundle_adjustment.cc:555). Right-looking blocked Cholesky, NB = 64, every product on
// v_mfma_f64_16x16x4_f64:
//
//   diag    a 64x64 diagonal tile is factorised AND inverted (tile_potrf_inv_la: 16x16 blocks whose 16
//           pivot steps are one rank-1 matrix instruction each, 4x4 blocks scheduled with look-ahead);
//   trsm    A_ik <- A_ik L_kk^-T is then a 64x64x64 product against the explicit inverse;
//   update  trailing tiles C_ij -= A_ik A_jk^T; the work-group that owns tile (k+1, k+1) goes on to
//           factorise + invert it, so a panel step is ONE launch (two for large trailing matrices);
//   fronts  with a nested-dissection order the uncoupled leading parts are factorised concurrently
//           (blockIdx.z = front); their contributions to the separator go to shadow blocks that are merged
//           before the separator's own (dense) chain;
//   solve   the right-hand side rides along as an extra row block (forward substitution is free); the
//           backward substitution is one launch of flag-synchronised work-groups, one per tile row.
//
// At BA sizes the cost is the chain of dependent panel steps, not the flops (DESIGN.md sections 4 and 6).
#include "internal.h"
#include <algorithm>
#include <cstdlib>

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

// Same value with one third-order (Halley) correction of the hardware seed instead of two Newton
// steps: y (1 + h (1/2 + 3/8 h)), h = 1 - d y^2. Seed error e -> ~e^3, two dependent operations shorter.
__device__ __forceinline__ double rsqrt_finish(double y, double h) {
  return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
}
__device__ __forceinline__ double rsqrt_halley(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  return rsqrt_finish(y, __builtin_fma(-(d * y), y, 1.0));
}

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// 16x16 SPD block: Cholesky factor AND the inverse of the factor, 16 pivot steps, the whole block
// living in ONE wave's MFMA accumulator layout (reg r of lane l = element [(l >> 4) + 4 r][l & 15]).
//
// Right-looking on the full symmetric block: at step J the scaled row J (u = a_J / sqrt(a_JJ),
// = column J of L by symmetry) sits in the 16 lanes of group t = J & 3, register J >> 2 — exactly
// where v_mfma_f64_16x16x4_f64 expects the k = t slice of both of its operands. So the rank-1
// update  A -= u u^T  is ONE matrix instruction whose operands are each lane's own (masked) value:
// no shuffles, no LDS, no per-column FMAs. The same instruction with the scaled row J of the
// running inverse as second operand applies the row operations to the identity, so X = L^-1 is
// finished together with the factor:  X -= u x_J^T.
// In: acc = the block (both triangles). Out: xacc = L^-1 (lower). Returns false on a bad pivot.
template <int J>
__device__ __forceinline__ void potrf_inv16_step(d4& acc, d4& xacc, d4& xfin, int lane, bool& ok) {
  constexpr int T4 = J & 3, RR = J >> 2;
  const double piv = readlane_d(acc[RR], 16 * T4 + J);
  ok = ok && (piv > 0.0);  // recorded, not repaired: a bad pivot just propagates NaNs and the solve is rejected
  const double rs = rsqrt_halley(piv);
  const bool in_row = (lane >> 4) == T4;
  const bool live = in_row && (lane & 15) >= J;
  const double u = live ? acc[RR] * rs : 0.0;      // row J of L^T (zero left of the diagonal)
  const double xj = in_row ? xacc[RR] * rs : 0.0;  // row J of the inverse, scaled = final
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
  xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, xj, xacc, 0, 0, 0);
  // Row J of acc / xacc is garbage now, but never read again: later steps only touch rows > J.
  // The finished inverse row is collected on the side (off the matrix-core dependency chain).
  xfin[RR] = in_row ? xj : xfin[RR];
}
// (Tried and measured slower on gfx950, scripts/_dbg/tile_bench.hip: running the scalar pivot recurrence
// piv_{J+1} = a11 - (a10 rs_J)^2 one step ahead of the matrix instructions so that the rsqrt overlaps the
// MFMA latency - 4600 instead of 4200 cycles per block, with or without a pinned instruction order.)
__device__ __forceinline__ bool potrf_inv16(d4& acc, d4& xacc, int lane) {
  bool ok = true;
  const int li = lane & 15, lk = lane >> 4;
  d4 xfin = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) xacc[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
  potrf_inv16_step<0>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<1>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<2>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<3>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<4>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<5>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<6>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<7>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<8>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<9>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<10>(acc, xacc, xfin, lane, ok); potrf_inv16_step<11>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<12>(acc, xacc, xfin, lane, ok); potrf_inv16_step<13>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<14>(acc, xacc, xfin, lane, ok); potrf_inv16_step<15>(acc, xacc, xfin, lane, ok);
  xacc = xfin;
  return ok;
}

// acc += A(16x16, row-major lda) * B^T  (NT)   or   A * B (NN), K = 16, operands in LDS.
// v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][k = l >> 4], B[k = l >> 4][l & 15];
// reg r of lane l holds D[(l >> 4) + 4 r][l & 15].
__device__ __forceinline__ d4 gemm16(const double* A, int lda, const double* B, int ldb, bool transB,
                                     double sign, d4 acc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4) {
    const double av = sign * A[li * lda + kk + lk];
    const double bv = transB ? B[li * ldb + kk + lk] : B[(kk + lk) * ldb + li];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ d4 load_d16(const double* C, int ldc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = C[(lk + 4 * r) * ldc + li];
  return v;
}
__device__ __forceinline__ void store_d16(double* C, int ldc, d4 v, int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(lk + 4 * r) * ldc + li] = v[r];
}

constexpr int kFuseBelow = 24;  // fuse the panel solve into the update when <= this many row blocks remain ...
constexpr int kFuseTasks = 256;  // ... and the step (all fronts) has at most this many tile updates
constexpr int kMaxBacksolveGroups = 2048;  // single-launch backward substitution up to this many tile rows

// Order LDS traffic inside ONE wave: a block written by some lanes is read back by other lanes of
// the same wave. The hardware executes a wave's LDS instructions in order; this only stops the
// compiler from moving the loads above the stores.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Look-ahead variant of tile_potrf_inv: same result in Ti (L^-1, lower; upper zero), two LDS tiles,
// no extra scratch, 6 work-group barriers instead of 18.
//
// The critical chain of a blocked tile factorisation is  potrf(D_c) -> L_{c+1,c} -> D_{c+1} -> potrf.
// Wave 0 walks exactly that chain; everything else (the rest of panel c, the trailing update by
// panel c, the off-diagonal blocks of the inverse) is done by waves 1..3 WHILE wave 0 is inside the
// next 16-pivot factorisation, each wave re-deriving the few panel blocks it needs instead of waiting
// for their owner (a 16^3 product costs ~4 matrix instructions, a barrier costs the whole potrf).
//
// LDS use: T lower blocks = the matrix (column c stays raw, blocks right of it are updated in place
// by their owner wave); T upper blocks = per-wave scratch; Ti diagonal + lower = the inverse;
// Ti upper block (c, j) = L_jc while the factorisation runs (zeroed at the end).
__device__ __forceinline__ bool tile_potrf_inv_la(double* T, double* Ti, int tid) {
  const int wv = tid >> 6, lane = tid & 63;
  const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
  auto Tb = [&](int i, int j) { return T + (16 * i) * GLD + 16 * j; };
  auto Xb = [&](int i, int j) { return Ti + (16 * i) * GLD + 16 * j; };
  auto Lb = [&](int i, int j) { return Ti + (16 * j) * GLD + 16 * i; };  // L_ij (i > j) parked at Ti block (j, i)
  // P = B * Dinv^T -> dst
  auto panel = [&](const double* B, const double* Dinv, double* dst) {
    d4 a = gemm16(B, GLD, Dinv, GLD, true, 1.0, zero, lane);
    store_d16(dst, GLD, a, lane);
  };
  // C -= Pa * Pb^T (in place)
  auto downdate = [&](double* C, const double* Pa, const double* Pb) {
    d4 a = load_d16(C, GLD, lane);
    a = gemm16(Pa, GLD, Pb, GLD, true, -1.0, a, lane);
    store_d16(C, GLD, a, lane);
  };
  // factor D_c = C - P P^T in registers, park its inverse in Ti(c, c)
  auto chain = [&](double* C, const double* P, int c) -> bool {
    d4 a = load_d16(C, GLD, lane), x;
    if (P) a = gemm16(P, GLD, P, GLD, true, -1.0, a, lane);
    const bool ok = potrf_inv16(a, x, lane);
    store_d16(Xb(c, c), GLD, x, lane);
    return ok;
  };
  bool ok = true;
  // ---- phase 0: D_0
  if (wv == 0) ok = chain(Tb(0, 0), nullptr, 0);
  __syncthreads();
  // ---- phase 1: panel 0; wave 0 goes on to D_1
  if (wv == 0) {
    panel(Tb(1, 0), Xb(0, 0), Lb(1, 0));
    wave_lds_sync();
    ok = chain(Tb(1, 1), Lb(1, 0), 1) && ok;
  } else if (wv == 1) {
    panel(Tb(2, 0), Xb(0, 0), Lb(2, 0));
    panel(Tb(1, 0), Xb(0, 0), Tb(0, 1));
    wave_lds_sync();
    downdate(Tb(2, 1), Lb(2, 0), Tb(0, 1));
    downdate(Tb(2, 2), Lb(2, 0), Lb(2, 0));
  } else if (wv == 2) {
    panel(Tb(3, 0), Xb(0, 0), Lb(3, 0));
    panel(Tb(1, 0), Xb(0, 0), Tb(0, 2));
    wave_lds_sync();
    downdate(Tb(3, 1), Lb(3, 0), Tb(0, 2));
    downdate(Tb(3, 3), Lb(3, 0), Lb(3, 0));
  } else {
    panel(Tb(3, 0), Xb(0, 0), Tb(1, 2));
    panel(Tb(2, 0), Xb(0, 0), Tb(1, 3));
    wave_lds_sync();
    downdate(Tb(3, 2), Tb(1, 2), Tb(1, 3));
  }
  __syncthreads();
  // ---- phase 2: panel 1; wave 0 goes on to D_2; wave 3 starts on the inverse
  if (wv == 0) {
    panel(Tb(2, 1), Xb(1, 1), Lb(2, 1));
    wave_lds_sync();
    ok = chain(Tb(2, 2), Lb(2, 1), 2) && ok;
  } else if (wv == 1) {
    panel(Tb(3, 1), Xb(1, 1), Lb(3, 1));
    panel(Tb(2, 1), Xb(1, 1), Tb(0, 1));
    wave_lds_sync();
    downdate(Tb(3, 2), Lb(3, 1), Tb(0, 1));
  } else if (wv == 2) {
    panel(Tb(3, 1), Xb(1, 1), Tb(0, 2));
    wave_lds_sync();
    downdate(Tb(3, 3), Tb(0, 2), Tb(0, 2));
  } else {
    // X_10 = -Dinv_1 (L_10 Dinv_0)
    d4 m = gemm16(Lb(1, 0), GLD, Xb(0, 0), GLD, false, 1.0, zero, lane);
    store_d16(Tb(1, 2), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(1, 1), GLD, Tb(1, 2), GLD, false, -1.0, zero, lane);
    store_d16(Xb(1, 0), GLD, m, lane);
  }
  __syncthreads();
  // ---- phase 3: panel 2; wave 0 goes on to D_3
  if (wv == 0) {
    panel(Tb(3, 2), Xb(2, 2), Lb(3, 2));
    wave_lds_sync();
    ok = chain(Tb(3, 3), Lb(3, 2), 3) && ok;
  } else if (wv == 1) {
    // X_21 = -Dinv_2 (L_21 Dinv_1)
    d4 m = gemm16(Lb(2, 1), GLD, Xb(1, 1), GLD, false, 1.0, zero, lane);
    store_d16(Tb(0, 1), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(2, 2), GLD, Tb(0, 1), GLD, false, -1.0, zero, lane);
    store_d16(Xb(2, 1), GLD, m, lane);
  } else if (wv == 2) {
    // X_20 = -Dinv_2 (L_20 X_00 + L_21 X_10)
    d4 m = gemm16(Lb(2, 0), GLD, Xb(0, 0), GLD, false, 1.0, zero, lane);
    m = gemm16(Lb(2, 1), GLD, Xb(1, 0), GLD, false, 1.0, m, lane);
    store_d16(Tb(0, 2), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(2, 2), GLD, Tb(0, 2), GLD, false, -1.0, zero, lane);
    store_d16(Xb(2, 0), GLD, m, lane);
  }
  __syncthreads();
  // ---- phase 4: last block row of the inverse, X_3j = -Dinv_3 sum_{k=j}^{2} L_3k X_kj
  if (wv < 3) {
    const int j = 2 - wv;  // wave 0: X_32, wave 1: X_31, wave 2: X_30
    double* scr = wv == 0 ? Tb(1, 2) : (wv == 1 ? Tb(0, 1) : Tb(0, 2));
    d4 m = zero;
    for (int k = j; k < 3; ++k) m = gemm16(Lb(3, k), GLD, Xb(k, j), GLD, false, 1.0, m, lane);
    store_d16(scr, GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(3, 3), GLD, scr, GLD, false, -1.0, zero, lane);
    store_d16(Xb(3, j), GLD, m, lane);
  }
  __syncthreads();
  // ---- the parked factor blocks leave the upper triangle of Ti
  for (int e = tid; e < 6 * 256; e += 256) {
    const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
    const int bi = b < 3 ? 0 : (b < 5 ? 1 : 2), bj = b < 3 ? b + 1 : (b < 5 ? b - 1 : 3);
    Ti[(16 * bi + r) * GLD + 16 * bj + c] = 0.0;
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

// Factor + invert the diagonal tiles listed in `tiles` (one work-group each): the first tile of every
// front. Block 0 also clears the backward substitution's flags for this solve.
__global__ void __launch_bounds__(256) k_chol_diag0(const double* __restrict__ M, int ld, const int* __restrict__ tiles,
                                                    double* __restrict__ inv, double* __restrict__ fail,
                                                    unsigned* __restrict__ flags, int nflags) {
  __shared__ __attribute__((aligned(16))) double T[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ti[NB * GLD];
  const int tid = threadIdx.x;
  const int t = tiles[blockIdx.x];
  if (blockIdx.x == 0)
    for (int f = tid; f < nflags; f += 256) flags[f] = 0u;
  load_tile(M + (size_t)t * NB * ld + (size_t)t * NB, ld, T, tid);
  __syncthreads();
  const bool ok = tile_potrf_inv_la(T, Ti, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(inv + (size_t)t * NB * NB, NB, Ti, tid);
}

// Panel solve, front blockIdx.z: row block blockIdx.x of the front's active list (the last one is the
// right-hand-side block):   A_ik <- A_ik L_kk^-T = A_ik (L_kk^-1)^T
__global__ void __launch_bounds__(256) k_chol_trsm(const double* __restrict__ M, double* __restrict__ Lout, int ld,
                                                   const CholFront* __restrict__ fronts,
                                                   const double* __restrict__ inv, const int* __restrict__ rows,
                                                   int aug) {
  const CholFront F = fronts[blockIdx.z];
  if ((int)blockIdx.x > F.na) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  const int tid = threadIdx.x;
  const int k = F.k;
  const int i = (int)blockIdx.x < F.na ? rows[F.act_off + blockIdx.x] : aug;  // active row block, or the right-hand side
  const double* Ain = M + (size_t)i * NB * ld + (size_t)k * NB;
  double* A = Lout + (size_t)i * NB * ld + (size_t)k * NB;
  load_tile(Ain, ld, As, tid);
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

// store the wave's 2x2 MFMA tiles (D layout) of a 64x64 product into an LDS tile (pitch GLD)
__device__ __forceinline__ void quadrant_to_lds(double* S, int wr, int wc, int lane, const d4 acc[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(wr + 16 * m + lk + 4 * r) * GLD + wc + 16 * n + li] = acc[m][n][r];
}

// Panel k, fused triangular solve + trailing update. Work-group (i, j), k < j <= i (i may be the
// right-hand-side block):
//   P_i = A_ik L_kk^-T,  P_j = A_jk L_kk^-T   (re-derived here from the raw panel tiles: two
//                                               extra 64^3 MFMA products instead of a launch)
//   tile (i, j) -= P_i P_j^T
// Column j == k+1 work-groups store P_i (the final L_ik, needed by the backward substitution);
// the owner of tile (k+1, k+1) then factorises + inverts it for the next panel.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_chol_update(double* __restrict__ M, double* __restrict__ Lout, int ld,
                                                     const CholFront* __restrict__ fronts,
                                                     double* __restrict__ inv, double* __restrict__ fail,
                                                     const int* __restrict__ rows, int aug,
                                                     double* __restrict__ shadow) {
  const CholFront F = fronts[blockIdx.z];
  const int na = F.na, k = F.k;
  if ((int)blockIdx.x >= na || (int)blockIdx.y > na) return;
  // only the row blocks whose envelope reaches panel k take part (act[0] is k + 1 when the front goes on)
  const int* act = rows + F.act_off;
  const int j = act[blockIdx.x], i = (int)blockIdx.y < na ? act[blockIdx.y] : aug;
  if (j > i) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Cs[NB * GLD];
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  // the tile being updated (ancestor columns of a front with a shadow: that front's shadow block);
  // its loads are issued first so that their latency hides behind the products
  double* C = M + (size_t)i * NB * ld + (size_t)j * NB;
  size_t ldc = (size_t)ld;
  if (F.sh_off >= 0 && j >= F.sh_begin) {
    ldc = (size_t)(aug - F.sh_begin) * NB;
    C = shadow + F.sh_off + (size_t)(i - F.sh_begin) * NB * ldc + (size_t)(j - F.sh_begin) * NB;
  }
  d4 cin[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) cin[m][n][r] = C[(size_t)(wr + 16 * m + lk + 4 * r) * ldc + wc + 16 * n + li];
  d4 acc[2][2];
  if constexpr (FUSED) {
    load_tile(M + (size_t)i * NB * ld + (size_t)k * NB, ld, As, tid);
    load_tile(inv + (size_t)k * NB * NB, NB, Bs, tid);
    if (i != j) load_tile(M + (size_t)j * NB * ld + (size_t)k * NB, ld, Cs, tid);
    __syncthreads();
    mfma_quadrant_nt(As, Bs, wr, wc, lane, acc);      // P_i
    d4 accj[2][2];
    if (i != j) mfma_quadrant_nt(Cs, Bs, wr, wc, lane, accj);  // P_j
    __syncthreads();
    quadrant_to_lds(As, wr, wc, lane, acc);
    if (i != j) quadrant_to_lds(Cs, wr, wc, lane, accj);
    __syncthreads();
    // L_ik is final; it goes to the second matrix (the raw tile is still being read by the
    // other work-groups of this launch)
    if (blockIdx.x == 0) store_tile(Lout + (size_t)i * NB * ld + (size_t)k * NB, ld, As, tid);
  } else {
    // panel tiles were already solved by k_chol_trsm into the second matrix
    load_tile(Lout + (size_t)i * NB * ld + (size_t)k * NB, ld, As, tid);
    if (i != j) load_tile(Lout + (size_t)j * NB * ld + (size_t)k * NB, ld, Cs, tid);
    __syncthreads();
  }
  const double* Pj = (i != j) ? Cs : As;
  mfma_quadrant_nt(As, Pj, wr, wc, lane, acc);
  const bool next_diag = (blockIdx.x == 0 && blockIdx.y == 0 && F.factor_next);
  if (!next_diag) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(wr + 16 * m + lk + 4 * r) * ldc + wc + 16 * n + li] = cin[m][n][r] - acc[m][n][r];
    return;
  }
  // tile (k+1, k+1): keep the updated tile in LDS, factorise + invert it for the next panel
  __syncthreads();  // everyone is done reading As / Cs
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr + 16 * m + lk + 4 * r) * GLD + wc + 16 * n + li] = cin[m][n][r] - acc[m][n][r];
  __syncthreads();
  const bool ok = tile_potrf_inv_la(Cs, As, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(inv + (size_t)(k + 1) * NB * NB, NB, As, tid);
}

// End of a tree level: M[i][j] += sum of the level's shadow blocks that cover tile (i, j) (i may be the
// right-hand-side row). A shadow with origin o covers tiles i, j >= o. grid = (nb - begin, nb - begin + 1).
__global__ void __launch_bounds__(256) k_chol_merge(double* __restrict__ M, int ld, const double* __restrict__ shadow,
                                                    const CholMerge* __restrict__ merges, int num_merges, int begin,
                                                    int aug) {
  const int j = begin + blockIdx.x, i = begin + blockIdx.y;
  if (j > i) return;
  double* C = M + (size_t)i * NB * ld + (size_t)j * NB;
  for (int e = threadIdx.x; e < NB * NB / 2; e += 256) {
    const int row = e >> 5, c2 = (e & 31) * 2;
    double2 v = *reinterpret_cast<const double2*>(C + (size_t)row * ld + c2);
    for (int q = 0; q < num_merges; ++q) {
      const int o = merges[q].sh_begin;
      if (j < o) continue;  // (i >= j >= o)
      const size_t lds = (size_t)(aug - o) * NB;
      const double* Sh = shadow + merges[q].sh_off + (size_t)(i - o) * NB * lds + (size_t)(j - o) * NB;
      const double2 w = *reinterpret_cast<const double2*>(Sh + (size_t)row * lds + c2);
      v.x += w.x; v.y += w.y;
    }
    *reinterpret_cast<double2*>(C + (size_t)row * ld + c2) = v;
  }
}

// Backward substitution, tile k (fallback for very many tile rows, single segment only):
// y_k = L_kk^-T z_k; then z_j -= L_kj^T y_k for j < k.
// grid = k + 1: block k stores y_k, block j < k updates z_j (disjoint segments).
__global__ void __launch_bounds__(64) k_chol_backsolve(const double* __restrict__ M, int ld, int k, int first,
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
  const int j = first + blockIdx.x;  // L_kj is structurally zero left of `first`
  if (j == k) { y[k * NB + lane] = mine; return; }
  yk[lane] = mine;
  __syncthreads();
  const double* L = M + (size_t)k * NB * ld + (size_t)j * NB;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; m += 2) {
    a0 += L[(size_t)m * ld + lane] * yk[m];
    a1 += L[(size_t)(m + 1) * ld + lane] * yk[m + 1];
  }
  z[j * NB + lane] -= a0 + a1;
}
__global__ void k_scatter_y(int n, const int* __restrict__ scatter, const double* __restrict__ y, double* __restrict__ y_nat) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && scatter[t] >= 0) y_nat[scatter[t]] = y[t];
}

// Backward substitution L^T x = z in ONE launch: work-group b owns tile row k = nb - 1 - b,
//   x_k = L_kk^-T (z_k - sum_{i > k} L_ik^T x_i),
// and consumes the x_i in decreasing i as their owners publish them (flag per tile, release/acquire at
// agent scope). Owners of later rows have smaller block indices, so they are dispatched first and the
// wait can never dead-lock. Each of the 4 waves takes 16 of the 64 rows of every tile; the tile values
// are fetched BEFORE the wait, so a step of the chain is {flag + 64 values of x, 16 FMAs, two LDS
// reductions}, not a kernel launch. Tile (i, k) takes part iff k is inside row i's envelope for k's
// segment - uncoupled parts of a nested-dissection ordering therefore never wait for each other.
__global__ void __launch_bounds__(256) k_chol_backsolve_all(const double* __restrict__ L, int ld, int nb, int nseg,
                                                            const int* __restrict__ seg_of_tile,
                                                            const int* __restrict__ seg_first,
                                                            const double* __restrict__ inv,
                                                            const double* __restrict__ z, double* y,
                                                            unsigned* flags, const int* __restrict__ scatter,
                                                            double* __restrict__ y_nat) {
  const int k = nb - 1 - (int)blockIdx.x;
  const int sk = seg_of_tile[k];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  double acc = (wv == 0) ? z[(size_t)k * NB + lane] : 0.0;
  // this wave's 16 rows of L_kk^-1 (lower, compact 64 x 64)
  double li[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) li[q] = inv[(size_t)k * NB * NB + (size_t)(16 * wv + q) * NB + lane];
  int i = nb - 1;
  while (i > k && seg_first[i * nseg + sk] > k) --i;  // tiles outside the envelope are structurally zero
  double l[16];
  if (i > k) {
    const double* Lt = L + (size_t)i * NB * ld + (size_t)k * NB + (size_t)(16 * wv) * ld + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) l[q] = Lt[(size_t)q * ld];
  }
  while (i > k) {
    int nx = i - 1;
    while (nx > k && seg_first[nx * nseg + sk] > k) --nx;
    double ln[16];
    if (nx > k) {
      const double* Lt = L + (size_t)nx * NB * ld + (size_t)k * NB + (size_t)(16 * wv) * ld + lane;
#pragma unroll
      for (int q = 0; q < 16; ++q) ln[q] = Lt[(size_t)q * ld];
    }
    while (__hip_atomic_load(&flags[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
    const double* xi = y + (size_t)i * NB + 16 * wv;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      a0 = __builtin_fma(l[q], xi[q], a0);
      a1 = __builtin_fma(l[q + 1], xi[q + 1], a1);
    }
    acc -= a0 + a1;
#pragma unroll
    for (int q = 0; q < 16; ++q) l[q] = ln[q];
    i = nx;
  }
  part[wv][lane] = acc;
  __syncthreads();
  const double r = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];  // (z_k - sum)_lane
  __syncthreads();
  if (wv == 0) part[0][lane] = r;
  __syncthreads();
  // x_k[c] = sum_m Linv[m][c] r[m]; this wave's m range
  double b0 = 0.0, b1 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q += 2) {
    b0 = __builtin_fma(li[q], part[0][16 * wv + q], b0);
    b1 = __builtin_fma(li[q + 1], part[0][16 * wv + q + 1], b1);
  }
  __syncthreads();
  part[wv][lane] = b0 + b1;
  __syncthreads();
  if (wv == 0) {
    const double x = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    y[(size_t)k * NB + lane] = x;
    if (scatter) {
      const int t = scatter[k * NB + lane];
      if (t >= 0) y_nat[t] = x;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&flags[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

void CholStructure::release() {
  if (d_ints) device_free(d_ints);
  if (d_fronts) device_free(d_fronts);
  if (d_shadow) device_free(d_shadow);
  if (d_merges) device_free(d_merges);
  d_ints = nullptr; d_fronts = nullptr; d_shadow = nullptr; d_merges = nullptr;
}
CholStructure::~CholStructure() { release(); }

hipError_t CholStructure::build_dense(int nb_) {
  std::vector<std::pair<int, int>> pairs;
  for (int i = 0; i < nb_; ++i) pairs.emplace_back(i, 0);  // every row starts at tile 0
  return build(nb_, pairs, {}, 0);
}

hipError_t CholStructure::build(int nb_, const std::vector<std::pair<int, int>>& tile_pairs,
                                const std::vector<CholNode>& tree_in, hipStream_t st) {
  release();
  nb = nb_;
  const int never = nb + 1;
  nodes = tree_in;
  // valid tree: contiguous ascending cover of [0, nb), parents after children, exactly one root (the last node)
  auto tree_ok = [&]() {
    if (nodes.empty() || nb > kMaxBacksolveGroups) return false;  // (the per-tile backward fallback is single-segment)
    int at = 0;
    for (size_t n = 0; n < nodes.size(); ++n) {
      if (nodes[n].begin != at || nodes[n].end <= nodes[n].begin) return false;
      at = nodes[n].end;
      const bool last = n + 1 == nodes.size();
      if (last ? nodes[n].parent != -1 : (nodes[n].parent <= (int)n || nodes[n].parent >= (int)nodes.size())) return false;
    }
    return at == nb;
  };
  std::vector<int> height;
  auto is_ancestor = [&](int a, int d) {  // a == d or a above d
    while (d != -1 && d != a) d = nodes[d].parent;
    return d == a;
  };
  auto assign_segments = [&]() {
    nseg = (int)nodes.size();
    seg_of_tile.assign(nb, 0);
    for (int n = 0; n < nseg; ++n)
      for (int t = nodes[n].begin; t < nodes[n].end; ++t) seg_of_tile[t] = n;
    seg_first.assign((size_t)nb * nseg, never);
    for (int i = 0; i < nb; ++i) seg_first[(size_t)i * nseg + seg_of_tile[i]] = i;
    for (const auto& pr : tile_pairs) {
      const int tr = pr.first, tc = pr.second;
      const int q = seg_of_tile[tc];
      if (!is_ancestor(seg_of_tile[tr], q)) return false;  // coupling across two branches: not a valid dissection
      int& f = seg_first[(size_t)tr * nseg + q];
      f = std::min(f, tc);
    }
    return true;
  };
  if (!tree_ok() || !assign_segments()) {
    nodes.assign(1, CholNode{0, nb, -1});
    assign_segments();
  }
  height.assign(nseg, 0);
  std::vector<char> is_leaf(nseg, 1);
  for (int n = 0; n < nseg; ++n)
    if (nodes[n].parent >= 0) { is_leaf[nodes[n].parent] = 0; height[nodes[n].parent] = std::max(height[nodes[n].parent], height[n] + 1); }
  // Separators fill in: dense inside, and an ancestor row that couples to any descendant of a separator couples
  // to the whole separator (children precede parents, so their entries are final when the parent is visited).
  if (nseg > 1)
    for (int n = 0; n < nseg; ++n) {
      if (is_leaf[n]) continue;
      for (int i = nodes[n].begin; i < nodes[n].end; ++i) seg_first[(size_t)i * nseg + n] = nodes[n].begin;
      for (int i = nodes[n].end; i < nb; ++i) {
        bool coupled = seg_first[(size_t)i * nseg + n] != never;
        for (int d = 0; d < n && !coupled; ++d)
          coupled = is_ancestor(n, d) && seg_first[(size_t)i * nseg + d] != never;
        if (coupled) seg_first[(size_t)i * nseg + n] = nodes[n].begin;
      }
    }

  // shadow blocks: every non-root node except the first of its level writes its ancestor updates to its own
  // block, origin = its parent's first tile, (nb - origin + 1) x (nb - origin) tiles
  const int H = *std::max_element(height.begin(), height.end());
  std::vector<long long> sh_off(nseg, -1);
  std::vector<int> sh_begin(nseg, 0);
  shadow_doubles = 0;
  for (int hgt = 0; hgt < H; ++hgt) {
    bool first = true;
    for (int n = 0; n < nseg; ++n) {
      if (height[n] != hgt || nodes[n].parent < 0) continue;
      sh_begin[n] = nodes[nodes[n].parent].begin;
      if (first) { first = false; continue; }
      const size_t ns = (size_t)(nb - sh_begin[n]);
      sh_off[n] = (long long)shadow_doubles;
      shadow_doubles += (ns + 1) * ns * 4096;
    }
  }

  std::vector<int> rows;
  fronts.clear(); steps.clear(); init_tiles.clear(); merges.clear();
  envelope_tiles = 0; factor_flops = 0.0; chain_steps = 0; num_fronts_max = 1;
  const double t3 = 64.0 * 64.0 * 64.0;
  auto add_front = [&](std::vector<CholFront>& cur, int k, int n) {
    CholFront F;
    F.k = k; F.act_off = (int)rows.size(); F.factor_next = (k + 1 < nodes[n].end) ? 1 : 0;
    F.sh_begin = sh_begin[n]; F.sh_off = sh_off[n];
    // Row k + 1 leads the list whenever the front goes on (its diagonal tile is factorised by the owner of
    // tile (k+1, k+1)); then the rows of the same node whose envelope reaches panel k, then the rows of its
    // ancestors that couple to it (rows of other branches never do).
    if (F.factor_next) rows.push_back(k + 1);
    for (int i = k + 2; i < nodes[n].end; ++i) if (seg_first[(size_t)i * nseg + n] <= k) rows.push_back(i);
    for (int i = nodes[n].end; i < nb; ++i) if (seg_first[(size_t)i * nseg + n] <= k) rows.push_back(i);
    F.na = (int)rows.size() - F.act_off;
    envelope_tiles += 1 + F.na;
    const double na = F.na;
    // tile factor + inverse (~2/3 t3), panel solves (na + rhs) * 2 t3, trailing update incl. rhs row
    factor_flops += (2.0 / 3.0) * t3 + (na + 1.0) * 2.0 * t3 + (na * (na + 1.0) / 2.0 + na) * 2.0 * t3;
    cur.push_back(F);
  };
  auto close_step = [&](std::vector<CholFront>& cur) {
    std::stable_sort(cur.begin(), cur.end(), [](const CholFront& a, const CholFront& b) { return (a.na > 0) > (b.na > 0); });
    CholStep S{};
    S.kind = 0; S.front_off = (int)fronts.size();
    for (const CholFront& F : cur) { if (F.na > 0) ++S.nf; else ++S.nf0; S.max_na = std::max(S.max_na, F.na); S.tasks += F.na * (F.na + 1) / 2 + F.na; }
    fronts.insert(fronts.end(), cur.begin(), cur.end());
    steps.push_back(S);
    cur.clear();
  };
  std::vector<CholFront> cur;
  for (int n = 0; n < nseg; ++n) if (height[n] == 0) init_tiles.push_back(nodes[n].begin);
  num_leaf_init = (int)init_tiles.size();
  for (int hgt = 0; hgt <= H; ++hgt) {
    int lead = 0, width = 0;
    for (int n = 0; n < nseg; ++n) if (height[n] == hgt) { lead = std::max(lead, nodes[n].end - nodes[n].begin); ++width; }
    num_fronts_max = std::max(num_fronts_max, width);
    for (int s = 0; s < lead; ++s) {
      for (int n = 0; n < nseg; ++n)
        if (height[n] == hgt && nodes[n].begin + s < nodes[n].end) add_front(cur, nodes[n].begin + s, n);
      close_step(cur);
    }
    chain_steps += lead;
    if (hgt == H) break;
    // end of the level: merge its shadows, then start the next level's nodes
    CholStep Mg{};
    Mg.kind = 1; Mg.front_off = (int)merges.size(); Mg.merge_begin = nb; Mg.init_off = (int)init_tiles.size();
    for (int n = 0; n < nseg; ++n)
      if (height[n] == hgt && sh_off[n] >= 0) { merges.push_back(CholMerge{sh_begin[n], sh_off[n]}); ++Mg.nf; Mg.merge_begin = std::min(Mg.merge_begin, sh_begin[n]); }
    for (int n = 0; n < nseg; ++n) if (height[n] == hgt + 1) { init_tiles.push_back(nodes[n].begin); ++Mg.nf0; }
    steps.push_back(Mg);
  }

  // device copies: rows | seg_of_tile | seg_first | init_tiles | flags
  std::vector<int> pack(rows);
  const size_t o_seg = pack.size();
  pack.insert(pack.end(), seg_of_tile.begin(), seg_of_tile.end());
  const size_t o_first = pack.size();
  pack.insert(pack.end(), seg_first.begin(), seg_first.end());
  const size_t o_init = pack.size();
  pack.insert(pack.end(), init_tiles.begin(), init_tiles.end());
  const size_t o_flags = pack.size();
  pack.resize(pack.size() + nb, 0);
  hipError_t e = device_alloc(reinterpret_cast<void**>(&d_ints), pack.size() * sizeof(int));
  if (e == hipSuccess) e = hipMemcpyAsync(d_ints, pack.data(), pack.size() * sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_fronts), std::max<size_t>(fronts.size(), 1) * sizeof(CholFront));
  if (e == hipSuccess && !fronts.empty())
    e = hipMemcpyAsync(d_fronts, fronts.data(), fronts.size() * sizeof(CholFront), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_merges), std::max<size_t>(merges.size(), 1) * sizeof(CholMerge));
  if (e == hipSuccess && !merges.empty())
    e = hipMemcpyAsync(d_merges, merges.data(), merges.size() * sizeof(CholMerge), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && shadow_doubles)
    e = device_alloc(reinterpret_cast<void**>(&d_shadow), shadow_doubles * sizeof(double));
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // the staging vectors go out of scope
  if (e != hipSuccess) { release(); return e; }
  d_rows = d_ints;
  d_seg_of_tile = d_ints + o_seg;
  d_seg_first = d_ints + o_first;
  d_init = d_ints + o_init;
  d_flags = reinterpret_cast<unsigned*>(d_ints + o_flags);
  return hipSuccess;
}

// diag_ws: n_pad * 64 doubles (the inverses of the factor's diagonal tiles); L: second
// (n_pad + 64) x n_pad matrix receiving the factor's off-diagonal tiles and the
// forward-substituted right-hand side. `cs` = tile structure + launch schedule of the matrix.
void dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* diag_ws, double* L, const CholStructure& cs,
                            const int* y_scatter, double* y_nat) {
  const int nb = n_pad / NB, ld = n_pad;
  double* inv = diag_ws;
  if (cs.shadow_doubles) (void)hipMemsetAsync(cs.d_shadow, 0, cs.shadow_doubles * sizeof(double), st);
  static const int fuse_below = [] { const char* e = std::getenv("MAVBA_CHOL_FUSE"); return e ? std::atoi(e) : kFuseBelow; }();  // tuning knobs
  static const int fuse_tasks = [] { const char* e = std::getenv("MAVBA_CHOL_FUSE_TASKS"); return e ? std::atoi(e) : kFuseTasks; }();
  hipLaunchKernelGGL(k_chol_diag0, dim3(cs.num_leaf_init), dim3(256), 0, st, M, ld, cs.d_init, inv, fail, cs.d_flags, nb);
  for (const CholStep& S : cs.steps) {
    if (S.kind == 1) {
      const int nt = nb - S.merge_begin;
      if (S.nf > 0)
        hipLaunchKernelGGL(k_chol_merge, dim3(nt, nt + 1), dim3(256), 0, st, M, ld, cs.d_shadow, cs.d_merges + S.front_off, S.nf,
                           S.merge_begin, nb);
      hipLaunchKernelGGL(k_chol_diag0, dim3(S.nf0), dim3(256), 0, st, M, ld, cs.d_init + S.init_off, inv, fail, cs.d_flags, 0);
      continue;
    }
    const CholFront* F = cs.d_fronts + S.front_off;
    if (S.max_na > fuse_below || S.tasks > fuse_tasks) {
      // much trailing work (more tile updates than one round of the CUs absorbs): one panel solve, then a lean
      // update (one product per work-group, 2 work-groups per CU)
      hipLaunchKernelGGL(k_chol_trsm, dim3(S.max_na + 1, 1, S.nf), dim3(256), 0, st, M, L, ld, F, inv, cs.d_rows, nb);
      hipLaunchKernelGGL((k_chol_update<false>), dim3(S.max_na, S.max_na + 1, S.nf), dim3(256), 0, st, M, L, ld, F, inv, fail,
                         cs.d_rows, nb, cs.d_shadow);
    } else if (S.max_na > 0) {
      // small trailing matrix: latency matters, fold the panel solve into the update launch
      hipLaunchKernelGGL((k_chol_update<true>), dim3(S.max_na, S.max_na + 1, S.nf), dim3(256), 0, st, M, L, ld, F, inv, fail,
                         cs.d_rows, nb, cs.d_shadow);
    }
    if (S.nf0)  // fronts with nothing below their tile: only the right-hand-side block is left
      hipLaunchKernelGGL(k_chol_trsm, dim3(1, 1, S.nf0), dim3(256), 0, st, M, L, ld, F + S.nf, inv, cs.d_rows, nb);
  }
  double* z = L + (size_t)n_pad * ld;
  if (nb <= kMaxBacksolveGroups) {
    hipLaunchKernelGGL(k_chol_backsolve_all, dim3(nb), dim3(256), 0, st, L, ld, nb, cs.nseg, cs.d_seg_of_tile, cs.d_seg_first,
                       inv, z, y, cs.d_flags, y_scatter, y_nat);
  } else {
    // more tile rows than work-groups that are certainly resident: one small launch per tile (single segment)
    for (int k = nb - 1; k >= 0; --k)
      hipLaunchKernelGGL(k_chol_backsolve, dim3(k - cs.seg_first[k] + 1), dim3(64), 0, st, L, ld, k, cs.seg_first[k], inv, z, y);
    if (y_scatter) hipLaunchKernelGGL(k_scatter_y, dim3((n_pad + 255) / 256), dim3(256), 0, st, n_pad, y_scatter, y, y_nat);
  }
}

}  // namespace mavba
